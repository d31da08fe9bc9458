// Does it pay to run the HBM-bound launches of the path (blur, ToRGB) on a SUBSET of the CUs beside the MFMA-bound convs of another
// batch?  The blur needs the whole chip for 1.1 ms per forward but only as many CUs as saturate HBM; the convs leave HBM mostly idle.
// This probe prices the idea before any plumbing: a streaming copy (the blur's stand-in: 16-byte loads + stores, grid-stride) and a
// bare MFMA loop with LDS fragment reads (the convs' stand-in, one 512-thread block per CU) run alone on the full chip, alone on CU
// subsets (hipExtStreamCreateWithCUMask), and concurrently on complementary subsets.
//   hipcc --offload-arch=gfx950 -O3 scripts/cu_mask_probe.hip -o build/cu_mask_probe && build/cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int frag128 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = src[i];
        v.x ^= 1u;
        dst[i] = v;
    }
}

// work-list form: blocks pull tiles of `tile` uint4 from a counter, so the kernel finishes when the WORK is done whatever the number of
// CUs it was given (a masked stream serialises a big grid anyway; this keeps the arithmetic identical)
__global__ __launch_bounds__(512, 1) void mfma_kernel(float* out, const unsigned* seed, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = seed[i];
    __syncthreads();
    f32x16 acc[2];
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave >> 1) * 64 + (lane & 31)) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            frag128 a[2][2], b[2];
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                b[part] = *reinterpret_cast<const frag128*>(pb + part * 16384 + (step + (it & 7)) * 16);
#pragma unroll
                for (int m = 0; m < 2; ++m) a[part][m] = *reinterpret_cast<const frag128*>(pa + part * 16384 + (step % 3) * 4096 + m * 512);
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[term == 2][m]), __builtin_bit_cast(f16x8, b[term == 1]),
                                                                    acc[m], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

static hipStream_t masked_stream(int every, int phase, bool invert) {      // CUs i with (i % every == phase) (or all the others)
    std::vector<uint32_t> mask(8, 0u);
    for (int i = 0; i < 256; ++i) {
        const bool in = (i % every) == phase;
        if (in != invert) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); exit(1); }
    return s;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = (size_t)1 << 26;        // 64 M uint4 = 1 GiB read + 1 GiB written per copy
    uint4 *src, *dst;
    hipMalloc(&src, n * 16); hipMalloc(&dst, n * 16);
    hipMemset(src, 1, n * 16);
    unsigned* h = (unsigned*)malloc(65536);
    srand(1);
    for (int i = 0; i < 16384; ++i) h[i] = f2h((rand() / (float)RAND_MAX - 0.5f) * 8.f) | ((unsigned)f2h((rand() / (float)RAND_MAX - 0.5f) * 8.f) << 16);
    unsigned* seed; hipMalloc(&seed, 65536); hipMemcpy(seed, h, 65536, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, sizeof(float) * 1024 * 512);
    hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipStream_t full; hipStreamCreate(&full);
    hipEvent_t e0, e1, f0, f1;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    const int iters = 4000;                  // per block: 20000 x 36 MFMAs per wave
    const int mfma_blocks = 1024;            // 4 rounds on 256 CUs
    auto run_copy = [&](hipStream_t s, int blocks) { copy_kernel<<<blocks, 256, 0, s>>>(src, dst, n); };
    auto run_mfma = [&](hipStream_t s) { mfma_kernel<<<mfma_blocks, 512, 131072, s>>>(out, seed, iters); };
    auto ms_of = [&](hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; };
    run_copy(full, 256 * 8); run_mfma(full); hipDeviceSynchronize();
    // alone, full chip
    hipEventRecord(e0, full); run_copy(full, 256 * 8); hipEventRecord(e1, full); hipEventSynchronize(e1);
    const float copy_full = ms_of(e0, e1);
    hipEventRecord(e0, full); run_mfma(full); hipEventRecord(e1, full); hipEventSynchronize(e1);
    const float mfma_full = ms_of(e0, e1);
    printf("alone, 256 CUs: copy %.3f ms = %.2f TB/s | mfma %.3f ms\n", copy_full, 2.0 * n * 16 / copy_full / 1e9, mfma_full);
    // back to back on one stream = today's schedule
    hipEventRecord(e0, full); run_copy(full, 256 * 8); run_mfma(full); hipEventRecord(e1, full); hipEventSynchronize(e1);
    printf("sequential on one stream: %.3f ms\n", ms_of(e0, e1));
    // two unmasked streams
    {
        hipStream_t s2; hipStreamCreate(&s2);
        hipDeviceSynchronize();
        hipEventRecord(e0, full); hipEventRecord(f0, s2);
        run_mfma(full); run_copy(s2, 256 * 8);
        hipEventRecord(e1, full); hipEventRecord(f1, s2); hipDeviceSynchronize();
        printf("two unmasked streams: mfma %.3f ms, copy %.3f ms\n", ms_of(e0, e1), ms_of(f0, f1));
    }
    for (int every : {8, 4, 3, 2}) {
        hipStream_t sh = masked_stream(every, 0, false), sm = masked_stream(every, 0, true);
        const int ncu_h = (256 + every - 1) / every;
        run_copy(sh, ncu_h * 8); run_mfma(sm); hipDeviceSynchronize();
        hipEventRecord(e0, sh); run_copy(sh, ncu_h * 8); hipEventRecord(e1, sh); hipEventSynchronize(e1);
        const float c_alone = ms_of(e0, e1);
        hipEventRecord(e0, sm); run_mfma(sm); hipEventRecord(e1, sm); hipEventSynchronize(e1);
        const float m_alone = ms_of(e0, e1);
        hipDeviceSynchronize();
        // concurrently: copies repeated on the small subset while the MFMA kernel runs on the rest
        hipEventRecord(e0, sm); hipEventRecord(f0, sh);
        run_mfma(sm);
        const int reps = 8;
        for (int r = 0; r < reps; ++r) run_copy(sh, ncu_h * 8);
        hipEventRecord(e1, sm); hipEventRecord(f1, sh); hipDeviceSynchronize();
        const float m_con = ms_of(e0, e1), c_con = ms_of(f0, f1) / reps;
        printf("copy on %3d CUs (1 of %d), mfma on %3d: alone copy %.3f ms = %.2f TB/s (full chip x%.2f), mfma %.3f ms (x%.2f) | together: copy %.3f ms = %.2f TB/s, mfma %.3f ms (x%.2f)\n",
               ncu_h, every, 256 - ncu_h, c_alone, 2.0 * n * 16 / c_alone / 1e9, c_alone / copy_full, m_alone, m_alone / mfma_full, c_con,
               2.0 * n * 16 / c_con / 1e9, m_con, m_con / mfma_full);
        hipStreamDestroy(sh); hipStreamDestroy(sm);
    }
    return 0;
}
