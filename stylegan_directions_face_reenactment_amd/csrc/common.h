// Shared helpers for the gfx950 kernels behind include/sgdfr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sgdfr.h"

namespace sgdfr {

void set_error(const char* fmt, ...);

// Returns 0 when the preceding launch was accepted; records hipGetLastError() otherwise.
int check_launch(const char* what);

#define SGDFR_REQUIRE(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            sgdfr::set_error(__VA_ARGS__); \
            return 1;                   \
        }                               \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// y = act( sum_s partials[s] + noise_w*noise + bias ), fixed order (modconv.hip); `inner` = elements per (b, c) plane
int launch_splitk_reduce(const float* partials, int splits, int64_t n, const float* noise, int64_t noise_bstride,
                         const float* noise_w, const float* bias, float* y, int C, int inner, int act, float slope, float gain,
                         hipStream_t st);

constexpr int kWave = 64;  // gfx950 wavefront

// Sum / max over the 64 lanes of a wave, the same value in every lane.  Four DPP steps inside each row of 16 lanes (quad
// swaps, half-row mirror, row mirror: VALU speed) and three v_readlane for the four row totals -- instead of six dependent
// ds_bpermute_b32 round trips (the __shfl_xor butterfly), which made the skinny style / mapping GEMMs butterfly-bound.
// PRECONDITION: all 64 lanes of the wave are active at the call.  v_readlane ignores EXEC, so a row whose lanes have left a
// loop or taken another branch contributes stale registers.  Call it only after reconvergence (wave-uniform control flow), or
// guard with `__builtin_popcountll(__ballot(1)) == 64` and fall back to per-lane atomics / __shfl_xor (backward.hip's
// blur_adjoint_kernel does).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror: every lane of a row holds the row's sum
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)),
                       __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48))));
}

// n / d for 0 <= n < 2^31 without a hardware divide: q = (n * mul) >> sh with mul = ceil(2^sh / d), sh = 31 + ceil(log2 d)
// (exact: the error term n * (mul*d - 2^sh) < 2^31 * d <= 2^sh).  The per-tile index arithmetic of the split conv kernels does
// ~35 divisions by launch-constant divisors per tile (~25 VALU instructions each); the host precomputes their multipliers.
struct FastDiv {
    unsigned mul, sh;
};
static inline FastDiv make_fastdiv(int d) {
    FastDiv f{0u, 31u};
    if (d < 1) d = 1;
    unsigned s = 0;
    while ((1ll << s) < d) ++s;
    f.sh = 31 + s;
    f.mul = (unsigned)(((1ull << f.sh) + (unsigned long long)d - 1) / (unsigned long long)d);
    return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }

// Input transform B^T d of the 1-D Winograd split conv (wsplit.hip; shared with the blur's hand-over in upfirdn2d.hip, which must
// produce the same bits as sgdfr_to_wsplit_f32).  F(2,3): d_j = in[2*tile - 1 + j], j = 0..3; F(4,3): d_j = in[4*tile - 1 + j], j = 0..5 (interpolation points
// 0, +-1, +-2, inf: Lavin & Gray's matrices).
template <int POS>
__device__ __forceinline__ void ws_input_transform(const float (&d)[POS], float (&v)[POS]) {
    if (POS == 4) {
        v[0] = d[0] - d[2];
        v[1] = d[1] + d[2];
        v[2] = d[2] - d[1];
        v[POS - 1] = d[1] - d[POS - 1];
    } else {
        // B^T rows: (4 0 -5 0 1 0) (0 -4 -4 1 1 0) (0 4 -4 -1 1 0) (0 -2 -1 2 1 0) (0 2 -1 -2 1 0) (0 4 0 -5 0 1)
        const float d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[POS - 2], d5 = d[POS - 1];
        const float a = d4 - 4.f * d2, b = d3 - 4.f * d1;         // (-4 d2 + d4), (-4 d1 + d3)
        const float c = d4 - d2, e = 2.f * (d3 - d1);             // (-d2 + d4), (-2 d1 + 2 d3)
        v[0] = fmaf(4.f, d[0], fmaf(-5.f, d2, d4));
        v[1] = a + b;
        v[2] = a - b;
        v[3] = c + e;
        v[POS - 2] = c - e;
        v[POS - 1] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
    }
}

__device__ __forceinline__ float lrelu_gain(float v, float slope, float gain) {
    return (v > 0.f ? v : v * slope) * gain;
}

// ---- fp8 cross terms of the split product (include/sgdfr.h SGDFR_SPLIT_FP16F8): shared by wswide.hip, split.hip and the producers
typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef int ws_frag __attribute__((ext_vector_type(4)));
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
// `sat` counts the halves whose hi term had to be clamped (the lo term follows it: |lo| <= 2^-11 |hi|, and the two exponents differ by
// 11) -- into the caller's saturation word, like a clamped fp16 pair: a batch louder than 2^2.8 x the calibrated maximum is re-rendered
// under a wider plan instead of running its cross terms at single-fp16 accuracy.
__device__ __forceinline__ uint2 ws_f8_half(unsigned h01, unsigned h23, unsigned l01, unsigned l23, float mul_lo, float mul_hi, bool weights_order,
                                            unsigned& sat) {
    const ws_f32x2 a = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, h01), ws_f32x2), b = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, h23), ws_f32x2);
    const ws_f32x2 c = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, l01), ws_f32x2), d = __builtin_convertvector(__builtin_bit_cast(ws_f16x2, l23), ws_f32x2);
    sat += (fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(b[0]), fabsf(b[1]))) * mul_hi > 448.f) ? 1u : 0u;
    const unsigned hi8 = ws_f8x4(a[0], a[1], b[0], b[1], mul_hi), lo8 = ws_f8x4(c[0], c[1], d[0], d[1], mul_lo);
    return weights_order ? make_uint2(hi8, lo8) : make_uint2(lo8, hi8);
}

// hi chunk (8 x fp16) + lo chunk (8 x fp16) of eight channels -> the lo chunk rewritten as fp8
__device__ __forceinline__ void ws_f8_lo_chunk(const uint4& vh, uint4& vl, float mul_lo, float mul_hi, bool weights_order, unsigned& sat) {
    const uint2 h0 = ws_f8_half(vh.x, vh.y, vl.x, vl.y, mul_lo, mul_hi, weights_order, sat);
    const uint2 h1 = ws_f8_half(vh.z, vh.w, vl.z, vl.w, mul_lo, mul_hi, weights_order, sat);
    vl = make_uint4(h0.x, h0.y, h1.x, h1.y);
}

// both cross terms of two (kernel row, channel block) slices: K = 64 = 2 k-halves x (chunk 0 | chunk 1), a chunk = (8 x fp8 | 8 x fp8)
// ordered (w_hi | w_lo) on the weight side and (x_lo | x_hi) on the activation side; scale_a carries the constant exponent
__device__ __forceinline__ ws_f32x16 ws_mfma_f8(ws_frag a0, ws_frag a1, ws_frag b0, ws_frag b1, ws_f32x16 c, int scale_a) {
    const ws_i32x8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const ws_i32x8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 127);
}

}  // namespace sgdfr
