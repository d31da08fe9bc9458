// Shared pieces of the split-Winograd conv kernels (wsplit.hip: 128 couts x 64 tiles per block; wswide.hip: 128 x 128):
// vector types, the split-pair conversion, the launch parameter block, counted-wait helpers.  See wsplit.hip for the algebra.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sgdfr {

typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));

constexpr float WS_F16_XSCALE = 0.0625f, WS_F16_WSCALE = 64.f, WS_F16_OUT = 0.25f, WS_F16_MAX = 65504.f;     // as split.hip
constexpr int WS_CB = 16;

// (ws_main_et, WS_F8_*, ws_f8_wexp, ws_f8x4 / ws_f8_half / ws_f8_lo_chunk, ws_mfma_f8: the fp8 cross-term pieces live in common.h -- split.hip's
//  transposed conv and the blur use them too)

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
    // (last: the fields above keep the offsets the kernels' scalar loads were tuned with -- moving them cost wsplit_kernel 450 SGPR reloads)
    int xs_f8;                   // the hand-over's lo chunks as fp8 cross-term operands (SGDFR_SPLIT_HANDOVER_F8: the XSF8 instantiations of both F(4,3) kernels)
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
