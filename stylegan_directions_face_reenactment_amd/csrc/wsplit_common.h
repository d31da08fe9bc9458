// Shared pieces of the split-Winograd conv kernels (wsplit.hip: 128 couts x 64 tiles per block; wswide.hip: 128 x 128):
// vector types, the split-pair conversion, the launch parameter block, counted-wait helpers.  See wsplit.hip for the algebra.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sgdfr {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef int ws_frag __attribute__((ext_vector_type(4)));

constexpr float WS_F16_XSCALE = 0.0625f, WS_F16_WSCALE = 64.f, WS_F16_OUT = 0.25f, WS_F16_MAX = 65504.f;     // as split.hip
constexpr int WS_CB = 16;

typedef int ws_i32x8 __attribute__((ext_vector_type(8)));

// SGDFR_SPLIT_FP16F8 (fp16 main term + fp8 cross terms): the hi chunks, the hand-over and every scale are SGDFR_SPLIT_FP16's
template <int ET>
struct ws_main_et { static constexpr int value = (ET == SGDFR_SPLIT_FP16F8) ? SGDFR_SPLIT_FP16 : ET; };

// e4m3 exponents of the fp8 chunks (e4m3: 448 at the top, full 3-bit mantissas down to 2^-6, a fixed step of 2^-9 below).
// Activations live in the fp16 domain, where the range plan puts the calibrated maximum of a layer's input near 2^10 (6 binades of
// headroom under 65504, functional.CALIBRATION_HEADROOM) and |lo| <= 2^-11 |hi|: hi * 2^-4 and lo * 2^7 put that maximum at 2^6 --
// elements down to 2^-12 of it keep their three bits, a batch up to 2^2.8 louder than calibrated still fits, beyond that the CROSS
// terms clamp at 448 (the main term does not: the error of such an element grows to the single-fp16 level, 2^-11 relative).
// Transformed weights: hi * 2^-EW, lo * 2^(11 - EW) with EW = floor(log2(max |w| * scale)) - 7 (the rows of G sum to <= 1 in
// magnitude, so max |U| <= max |w * scale| lands in [128, 256)), read from the pack's trailer.
constexpr int WS_F8_XHI = -4, WS_F8_XLO = 7;
static_assert(WS_F8_XLO == 11 + WS_F8_XHI, "both cross terms carry the same power of two");
__device__ __forceinline__ int ws_f8_wexp(float maxw) {
    const int e = (int)((__builtin_bit_cast(unsigned, maxw) >> 23) & 0xffu) - 127 - 7;
    return e < -40 ? -40 : e > 40 ? 40 : e;
}

// four values -> one dword of e4m3 (v_cvt_pk_fp8_f32: OCP e4m3 on gfx950 = format code 0 of v_mfma_scale_f32_32x32x64_f8f6f4)
__device__ __forceinline__ unsigned ws_f8x4(float v0, float v1, float v2, float v3, float mul) {
    // (beyond 448 the conversion would make a NaN)
    v0 = __builtin_amdgcn_fmed3f(v0 * mul, -448.f, 448.f); v1 = __builtin_amdgcn_fmed3f(v1 * mul, -448.f, 448.f);
    v2 = __builtin_amdgcn_fmed3f(v2 * mul, -448.f, 448.f); v3 = __builtin_amdgcn_fmed3f(v3 * mul, -448.f, 448.f);
    int a = 0;
    a = __builtin_amdgcn_cvt_pk_fp8_f32(v0, v1, a, false);
    a = __builtin_amdgcn_cvt_pk_fp8_f32(v2, v3, a, true);
    return (unsigned)a;
}

// The fp8 "lo" chunk of eight channels is two 8-byte halves (channels 0-3, 4-7), each (4 x first | 4 x second): first = lo, second =
// hi for activations, first = hi, second = lo for weights -- so byte k of a weight chunk meets byte k of an activation chunk in
// w_hi * x_lo or w_lo * x_hi.  One half from the packed fp16 pairs of its four channels:
__device__ __forceinline__ uint2 ws_f8_half(unsigned h01, unsigned h23, unsigned l01, unsigned l23, float mul_lo, float mul_hi, bool weights_order) {
    const ws_f32x2 a = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, h01), ws_f32x2), b = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, h23), ws_f32x2);
    const ws_f32x2 c = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, l01), ws_f32x2), d = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, l23), ws_f32x2);
    const unsigned hi8 = ws_f8x4(a[0], a[1], b[0], b[1], mul_hi), lo8 = ws_f8x4(c[0], c[1], d[0], d[1], mul_lo);
    return weights_order ? make_uint2(hi8, lo8) : make_uint2(lo8, hi8);
}

// hi chunk (8 x fp16) + lo chunk (8 x fp16) of eight channels -> the lo chunk rewritten as fp8
__device__ __forceinline__ void ws_f8_lo_chunk(const uint4& vh, uint4& vl, float mul_lo, float mul_hi, bool weights_order) {
    const uint2 h0 = ws_f8_half(vh.x, vh.y, vl.x, vl.y, mul_lo, mul_hi, weights_order);
    const uint2 h1 = ws_f8_half(vh.z, vh.w, vl.z, vl.w, mul_lo, mul_hi, weights_order);
    vl = make_uint4(h0.x, h0.y, h1.x, h1.y);
}

// both cross terms of two (kernel row, channel block) slices: K = 64 = 2 k-halves x (chunk 0 | chunk 1), a chunk = (8 x fp8 | 8 x fp8)
// ordered (w_hi | w_lo) on the weight side and (x_lo | x_hi) on the activation side; scale_a carries the constant exponent
__device__ __forceinline__ ws_f32x16 ws_mfma_f8(ws_frag a0, ws_frag a1, ws_frag b0, ws_frag b1, ws_f32x16 c, int scale_a) {
    const ws_i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const ws_i32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 127);
}

template <int ET>
__device__ __forceinline__ ws_f32x16 ws_mfma(ws_frag a, ws_frag b, ws_f32x16 c) {
    if (ET == SGDFR_SPLIT_FP16 || ET == SGDFR_SPLIT_FP16F8)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, a), __builtin_bit_cast(ws_f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ws_bf16x8, a), __builtin_bit_cast(ws_bf16x8, b), c, 0, 0, 0);
}

// two floats -> packed hi pair, packed lo pair (split.hip's split_pair: same rounding, same clamp-and-count rule)
template <int ET>
__device__ __forceinline__ void ws_pair(float a, float b, unsigned& hi, unsigned& lo, unsigned& sat) {
    if (ET == SGDFR_SPLIT_FP16) {
        sat += (!(fabsf(a) <= WS_F16_MAX) || !(fabsf(b) <= WS_F16_MAX)) ? 1u : 0u;
        a = __builtin_amdgcn_fmed3f(a, -WS_F16_MAX, WS_F16_MAX);
        b = __builtin_amdgcn_fmed3f(b, -WS_F16_MAX, WS_F16_MAX);
        const ws_f16x2 h = __builtin_convertvector((ws_f32x2){a, b}, ws_f16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const ws_f32x2 hf = __builtin_convertvector(h, ws_f32x2);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a - hf[0], b - hf[1]}, ws_f16x2));
    } else {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a, b}, ws_bf16x2));
        const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a - ha, b - hb}, ws_bf16x2));
    }
}

struct WsParams {
    const unsigned char* v;      // WS input
    const unsigned char* wsp;    // U pack
    const unsigned char* zeros;  // >= 16 zero bytes
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    float* y;
    const float* rgb_w;          // fused ToRGB, as split.hip: [3][Cout] weights, [B][Cout] styles, partial sums [B][T*3][H*W]
    const float* rgb_s;
    float* rgb_part;
    unsigned char* xs_out;       // the activation in the next (transposed) conv's split input form [B][Cout/8][hi,lo][H*W][8]
    const float* s_next;
    unsigned* sat;
    int B, Cin, Cout, H, W;
    int TW;                      // tiles per image row (W / 2)
    int TCT, TR, tct_shift;      // patch: TR rows x TCT tile columns (TR * TCT = 128)
    int tiles_x, tiles_y;
    int xs;                      // staged positions per (t, part, k-half) run: (TR + 2) * TCT
    int n_pix_tiles, n_cout_tiles;
    int total_blocks;            // tiles x cout tiles; the grid may be smaller (persistent blocks)
    int act;
    float slope, gain;
    int dbg;
    int desync;                  // first-round start spread: estimated block time in 4096-clock units (0 = off)
    FastDiv fd_xs, fd_tiles_x, fd_per_img, fd_npt;
};

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in 0..12 (the instruction takes an immediate; larger n wait for 12: stricter)
__device__ __forceinline__ void ws_wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: ws_wait_vmcnt<0>(); break;
        case 1: ws_wait_vmcnt<1>(); break;
        case 2: ws_wait_vmcnt<2>(); break;
        case 3: ws_wait_vmcnt<3>(); break;
        case 4: ws_wait_vmcnt<4>(); break;
        case 5: ws_wait_vmcnt<5>(); break;
        case 6: ws_wait_vmcnt<6>(); break;
        case 7: ws_wait_vmcnt<7>(); break;
        case 8: ws_wait_vmcnt<8>(); break;
        case 9: ws_wait_vmcnt<9>(); break;
        case 10: ws_wait_vmcnt<10>(); break;
        case 11: ws_wait_vmcnt<11>(); break;
        default: ws_wait_vmcnt<12>(); break;
    }
}

// wswide.hip: the F(4,3) conv on 128 couts x 128 tiles per block (half the transformed-weight bytes per MFMA).  Returns -1 when the
// shape / tile count does not suit it (the caller then launches wsplit_kernel), else 0 after the launch / an error code.
int wswide_try_launch(WsParams p, int arith, void* stream);
int wswide_by_tile_count(WsParams p);
unsigned int wswide_saturation_count(int reset);

}  // namespace sgdfr
