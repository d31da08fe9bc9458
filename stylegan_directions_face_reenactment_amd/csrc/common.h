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

}  // namespace sgdfr
