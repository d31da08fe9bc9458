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

__device__ __forceinline__ float lrelu_gain(float v, float slope, float gain) {
    return (v > 0.f ? v : v * slope) * gain;
}

}  // namespace sgdfr
