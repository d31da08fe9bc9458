// Measured MFMA ceiling of the device the library runs on (bench.py reports it beside the nominal peak).
//
// MI355X clocks to its power budget, and the power of the matrix cores depends on how many operand bits toggle: a bare loop
// of v_mfma_f32_32x32x16_f16 on ZERO operands sustains ~2.3 PFLOP/s, the same loop on RANDOM operands ~1.6 PFLOP/s, and
// with its fragments read from LDS at the split conv's ratio (8 ds_read_b128 per 12 MFMAs) ~1.46 PFLOP/s -- whatever the
// schedule, 1 or 2 waves per SIMD (scripts/mfma16_probe.hip has the longer sweep).  The nominal 2.5 PFLOP/s (2.4 GHz x
// every SIMD issuing back to back) is therefore not reachable by any kernel on real data; the conv roofline keeps it as
// its denominator (the rule of the bench contract) and prints this measured figure next to it.
#include "common.h"

namespace sgdfr {

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef int pfrag128 __attribute__((ext_vector_type(4)));

// counter hash -> two fp16 values of magnitude O(1) with all mantissa bits random
__device__ __forceinline__ unsigned probe_pair(unsigned i) {
    unsigned h = i * 2654435761u + 0x9e3779b9u;
    h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    // sign(1) exponent 01110..10000 (0.5 .. 4) mantissa random, per half
    const unsigned lo = (h & 0x83ffu) | ((14u + ((h >> 10) & 3u)) << 10);
    const unsigned hi = ((h >> 16) & 0x83ffu) | ((14u + ((h >> 26) & 3u)) << 10);
    return lo | (hi << 16);
}

template <int ET, int USE_LDS>
__global__ __launch_bounds__(512, 1) void mfma_ceiling_kernel(float* out, int iters, int random) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = random ? probe_pair(i) : 0u;
    __syncthreads();
    pf32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave & 3) * 64 + (lane & 31)) * 16;
    pfrag128 a[2][2], b[2][2];      // [part hi/lo][tile]
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[part][m] = *reinterpret_cast<const pfrag128*>(pa + part * 16384 + m * 512);
            b[part][m] = *reinterpret_cast<const pfrag128*>(pb + part * 16384 + m * 512);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            if (USE_LDS) {
#pragma unroll
                for (int part = 0; part < 2; ++part)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        a[part][m] = *reinterpret_cast<const pfrag128*>(pa + part * 16384 + tap * 4096 + m * 512 + (it & 1) * 2048);
                        b[part][m] = *reinterpret_cast<const pfrag128*>(pb + part * 16384 + (tap + m * 32 + (it & 7)) * 16);
                    }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)      // hi*hi, hi*lo, lo*hi: the split conv's three products
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (ET == SGDFR_SPLIT_FP16)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, a[t == 2][m]),
                                                                               __builtin_bit_cast(pf16x8, b[t == 1][n]), acc[m][n], 0, 0, 0);
                        else
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, a[t == 2][m]),
                                                                                __builtin_bit_cast(pbf16x8, b[t == 1][n]), acc[m][n], 0, 0, 0);
                    }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 512 + tid] = s;
}

}  // namespace sgdfr

using namespace sgdfr;

// Sustained rate of v_mfma_f32_32x32x16_{f16,bf16} on this device, in 16-bit TFLOP/s (divide by 3 for the split arithmetics'
// fp32-product rate).  lds_fragments: operands re-read from LDS at the split conv's ratio (else they stay in registers);
// random_operands: random mantissas (else zeros: the number a data-free microbenchmark would print).  One block of 8 waves
// per CU, `blocks` blocks (<= 0: 256).  Synchronises the stream; scratch = blocks*512 floats of device memory.
extern "C" int sgdfr_mfma_ceiling_probe(int arith, int lds_fragments, int random_operands, int iters, int blocks, float* scratch,
                                        double* tflops16, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "mfma_ceiling_probe: arith must be SGDFR_SPLIT_BF16/FP16");
    SGDFR_REQUIRE(scratch && tflops16 && iters > 0, "mfma_ceiling_probe: null pointer or iters <= 0");
    if (blocks <= 0) blocks = 256;
    hipStream_t st = as_stream(stream);
    void (*kern)(float*, int, int) =
        arith == SGDFR_SPLIT_FP16 ? (lds_fragments ? mfma_ceiling_kernel<SGDFR_SPLIT_FP16, 1> : mfma_ceiling_kernel<SGDFR_SPLIT_FP16, 0>)
                                  : (lds_fragments ? mfma_ceiling_kernel<SGDFR_SPLIT_BF16, 1> : mfma_ceiling_kernel<SGDFR_SPLIT_BF16, 0>);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        set_error("mfma_ceiling_probe: hipEventCreate failed");
        return 2;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, st, scratch, iters / 8 + 1, random_operands);      // warm-up (clocks, code)
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, st, scratch, iters, random_operands);
    hipEventRecord(e1, st);
    int rc = check_launch("mfma_ceiling_probe");
    float ms = 0.f;
    if (rc == 0 && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f)) {
        set_error("mfma_ceiling_probe: timing failed");
        rc = 2;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (rc) return rc;
    *tflops16 = (double)blocks * 8 * iters * 36 * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    return 0;
}
