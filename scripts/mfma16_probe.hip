// What does v_mfma_f32_32x32x16_f16 sustain on this part with RANDOM operands (the chip clocks to its power budget, and
// toggling sets the power)?  Arms: registers only / fragments from LDS at the split conv's ratio (8 ds_read_b128 per 12
// MFMAs) / the same with a barrier every 36 MFMAs; zero-filled vs random data; 1 or 2 waves per SIMD.  Prints TFLOP/s, the
// average shader clock (s_memtime ticks / wall time) and MFMA-pipe duty at that clock.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma16_probe.hip -o gpurun_out/mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int frag128 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void lds_void_p;
typedef const __attribute__((address_space(1))) void glb_void_p;
// DMA6: global_load_lds pieces per wave per SIX iterations (6 x 36 = 216 MFMAs = one channel block of the 128 x 512 tile: 14)
template <int WAVES, int USE_LDS, int BARRIER, int ORDER = 0, int DMA6 = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(float* out, const unsigned* seed, int iters, unsigned long long* clk, const unsigned char* src = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = seed[i];
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave & 3) * 64 + (lane & 31)) * 16;
    frag128 a[2][2], b[2][2];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[part][m] = *reinterpret_cast<const frag128*>(pa + part * 16384 + m * 512);
            b[part][m] = *reinterpret_cast<const frag128*>(pb + part * 16384 + m * 512);
        }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (DMA6) {
            const int ph = it % 6, n0 = ph * DMA6 / 6, n1 = (ph + 1) * DMA6 / 6;
            for (int v = n0; v < n1; ++v)
                __builtin_amdgcn_global_load_lds((glb_void_p*)(src + ((((size_t)blockIdx.x * 61 + it * 17 + wave * DMA6 + v) & 4095) << 10) + lane * 16),
                                                 (lds_void_p*)(lds + 65536 + ((wave * DMA6 + v) & 63) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            if (USE_LDS) {
#pragma unroll
                for (int part = 0; part < 2; ++part)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        a[part][m] = *reinterpret_cast<const frag128*>(pa + part * 16384 + tap * 4096 + m * 512 + (it & 1) * 2048);
                        b[part][m] = *reinterpret_cast<const frag128*>(pb + part * 16384 + (tap + m * 32 + (it & 7)) * 16);
                    }
            }
#define MF(PA, M, PB, N) acc[M][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA][M]), __builtin_bit_cast(f16x8, b[PB][N]), acc[M][N], 0, 0, 0)
            if (ORDER == 0) {          // the conv kernel's order: product term outermost
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) MF(t == 2, m, t == 1, n);
            } else if (ORDER == 1) {   // snake inside a term: one operand changes per step
#pragma unroll
                for (int t = 0; t < 3; ++t) { MF(t == 2, 0, t == 1, 0); MF(t == 2, 0, t == 1, 1); MF(t == 2, 1, t == 1, 1); MF(t == 2, 1, t == 1, 0); }
            } else if (ORDER == 2) {   // A-stationary: a_hi[m] meets b_hi0 b_hi1 b_lo0 b_lo1, then a_lo[m] meets b_hi0 b_hi1
#pragma unroll
                for (int m = 0; m < 2; ++m) { MF(0, m, 0, 0); MF(0, m, 0, 1); MF(0, m, 1, 1); MF(0, m, 1, 0); MF(1, m, 0, 0); MF(1, m, 0, 1); }
            } else if (ORDER == 3) {   // B-stationary
#pragma unroll
                for (int n = 0; n < 2; ++n) { MF(0, 0, 0, n); MF(0, 1, 0, n); MF(1, 1, 0, n); MF(1, 0, 0, n); MF(0, 0, 1, n); MF(0, 1, 1, n); }
            } else {                   // full Gray path over the 12 products
                MF(0, 0, 0, 0); MF(0, 0, 0, 1); MF(0, 0, 1, 1); MF(0, 0, 1, 0); MF(0, 1, 1, 0); MF(0, 1, 0, 0);
                MF(0, 1, 0, 1); MF(0, 1, 1, 1); MF(1, 1, 0, 1); MF(1, 1, 0, 0); MF(1, 0, 0, 0); MF(1, 0, 0, 1);
            }
#undef MF
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}


// 1-D Winograd F(2,3) pricing.  Per 16-channel block a plain wave tile (64 couts x 64 pixels) issues 9 taps x 12
// MFMAs fed by 9 x 8 ds_read_b128; the Winograd form of the SAME 4096 outputs (32 couts x 64 two-pixel tiles, four transform
// positions = 8 accumulators) issues 12 (ky, t) steps x 6 MFMAs fed by 12 x 6 reads -- 2/3 of the MFMAs, the same fragment reads.
// SHAPE 0: 1 A tile x 2 B tiles per step (6 reads / 6 MFMAs); SHAPE 1: one wave per SIMD with 16 accumulators (2 x 2 tiles x 4
// positions = 256 registers): 8 reads / 12 MFMAs per step.
// DMA: global_load_lds pieces (64 lanes x 16 B = 1 KB) per wave and iteration (one 16-channel block = 72 MFMAs per wave), read
// from a 4 MB L2-resident buffer into a scratch LDS region: 128 couts x 128 tiles per block stage 40 KB of V + 96 KB of U per
// channel block = 17 pieces per wave (the plain 128 x 512 tile: 110 KB per 216 MFMAs per wave).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
template <int WAVES, int SHAPE, int DMA>
__global__ __launch_bounds__(WAVES * 64, 1) void probe_wino(float* out, const unsigned* seed, int iters, unsigned long long* clk, const unsigned char* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = seed[i];
    __syncthreads();
    constexpr int MI = SHAPE ? 2 : 1, NI = 2;
    f32x16 acc[4][MI][NI];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave & 3) * 64 + (lane & 31)) * 16;
    frag128 a[2][MI], b[2][NI];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int step = 0; step < 12; ++step) {      // (ky, t)
            const int t = step & 3;
            if (DMA) {
                const int n0 = step * DMA / 12, n1 = (step + 1) * DMA / 12;
#pragma unroll
                for (int v = n0; v < n1; ++v)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(src + ((((size_t)blockIdx.x * 61 + it * 17 + wave * DMA + v) & 4095) << 10) + lane * 16),
                                                     (lds_void_t*)(lds + 65536 + ((wave * DMA + v) & 63) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int m = 0; m < MI; ++m)
                    a[part][m] = *reinterpret_cast<const frag128*>(pa + part * 16384 + (step % 3) * 4096 + m * 512 + (it & 1) * 2048);
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    b[part][n] = *reinterpret_cast<const frag128*>(pb + part * 16384 + (step + n * 32 + (it & 7)) * 16);
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int n = 0; n < NI; ++n)
                        acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[term == 2][m]),
                                                                               __builtin_bit_cast(f16x8, b[term == 1][n]), acc[t][m][n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[t][m][n][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

template <int WAVES, int SHAPE, int DMA = 0>
void run_wino(const char* name, const unsigned* seed_dev, const unsigned char* src = nullptr) {
    float* out; unsigned long long* clk;
    const int blocks = 256, iters = 3000;
    constexpr int MI = SHAPE ? 2 : 1;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe_wino<WAVES, SHAPE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    probe_wino<WAVES, SHAPE, DMA><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, 200, clk, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_wino<WAVES, SHAPE, DMA><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mfmas = (double)iters * 12 * 6 * MI;               // per wave
    const double flops = (double)blocks * WAVES * mfmas * 2.0 * 32 * 32 * 16;
    const double ghz = c / (ms * 1e6);
    const double duty = mfmas * 32 * (WAVES / 4) / (double)c;
    // outputs per step-set: 32*MI couts x 64 tiles x 2 pixels, K = 16 channels x 9 taps -> algorithmic fp32 flops
    const double alg = (double)blocks * WAVES * iters * (32.0 * MI) * 128 * 2.0 * 16 * 9;
    printf("%-34s waves %d: %8.3f ms %7.1f TFLOP/s (16-bit issued), %6.1f algorithmic fp32 TFLOP/s, clock %.2f GHz, MFMA duty %.2f\n",
           name, WAVES, ms, flops / ms / 1e9, alg / ms / 1e9, ghz, duty);
    hipFree(out); hipFree(clk);
}

// F(4,3) pricing: 4 outputs from 6 transformed inputs (18 MFMA columns per 16 channels and output QUAD instead of 36; V is 6
// values per 4 pixels = 1.5x the bytes, U 6/3 = 2x).  Six accumulators per wave (32 couts x 32 four-pixel tiles x 6 positions):
// per (ky, t) step 2 + 2 fragment reads for 3 MFMAs; 128 couts x 64 tiles per block stage 36 KB of V + 144 KB of U per channel
// block = 22.5 pieces per wave per 54 MFMAs.
template <int WAVES, int DMA>
__global__ __launch_bounds__(WAVES * 64, 1) void probe_f43(float* out, const unsigned* seed, int iters, unsigned long long* clk, const unsigned char* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = seed[i];
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave & 3) * 64 + (lane & 31)) * 16;
    frag128 a[2], b[2];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int step = 0; step < 18; ++step) {      // (ky, t)
            const int t = step % 6;
            if (DMA) {
                const int n0 = step * DMA / 18, n1 = (step + 1) * DMA / 18;
#pragma unroll
                for (int v = n0; v < n1; ++v)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(src + ((((size_t)blockIdx.x * 61 + it * 17 + wave * DMA + v) & 4095) << 10) + lane * 16),
                                                     (lds_void_t*)(lds + 65536 + ((wave * DMA + v) & 63) * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                a[part] = *reinterpret_cast<const frag128*>(pa + part * 16384 + (step % 3) * 4096 + (it & 1) * 2048);
                b[part] = *reinterpret_cast<const frag128*>(pb + part * 16384 + (step + (it & 7)) * 16);
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[term == 2]), __builtin_bit_cast(f16x8, b[term == 1]), acc[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

template <int WAVES, int DMA>
void run_f43(const char* name, const unsigned* seed_dev, const unsigned char* src) {
    float* out; unsigned long long* clk;
    const int blocks = 256, iters = 3000;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe_f43<WAVES, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    probe_f43<WAVES, DMA><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, 200, clk, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_f43<WAVES, DMA><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double mfmas = (double)iters * 54;
    const double flops = (double)blocks * WAVES * mfmas * 2.0 * 32 * 32 * 16;
    const double alg = (double)blocks * WAVES * iters * 32.0 * (32 * 4) * 2.0 * 16 * 9;     // 32 couts x 32 tiles x 4 pixels, K = 16 x 9
    printf("%-34s waves %d: %8.3f ms %7.1f TFLOP/s (16-bit issued), %6.1f algorithmic fp32 TFLOP/s, clock %.2f GHz, MFMA duty %.2f\n",
           name, WAVES, ms, flops / ms / 1e9, alg / ms / 1e9, c / (ms * 1e6), mfmas * 32 * (WAVES / 4) / (double)c);
    hipFree(out); hipFree(clk);
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }

template <int WAVES, int USE_LDS, int BARRIER, int ORDER = 0, int DMA6 = 0>
void run(const char* name, const unsigned* seed_dev, const unsigned char* src = nullptr) {
    float* out; unsigned long long* clk;
    const int blocks = 256, iters = 4000;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<WAVES, USE_LDS, BARRIER, ORDER, DMA6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    probe<WAVES, USE_LDS, BARRIER, ORDER, DMA6><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, 200, clk, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<WAVES, USE_LDS, BARRIER, ORDER, DMA6><<<blocks, WAVES * 64, 131072>>>(out, seed_dev, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * WAVES * iters * 36 * 2.0 * 32 * 32 * 16;
    const double ghz = c / (ms * 1e6);
    const double duty = (double)iters * 36 * 32 * (WAVES / 4) / (double)c;
    printf("%-34s waves %d: %8.3f ms %7.1f TFLOP/s (16-bit) = %6.1f fp32-equiv, clock %.2f GHz, MFMA duty %.2f\n", name, WAVES, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 3, ghz, duty);
    hipFree(out); hipFree(clk);
}

int main() {
    unsigned* h = (unsigned*)malloc(65536);
    unsigned* dz; unsigned* dr;
    hipMalloc(&dz, 65536); hipMalloc(&dr, 65536);
    hipMemset(dz, 0, 65536);
    srand(1);
    for (int i = 0; i < 16384; ++i) {
        float u0 = 0, u1 = 0;
        for (int k = 0; k < 12; ++k) { u0 += rand() / (float)RAND_MAX; u1 += rand() / (float)RAND_MAX; }
        h[i] = f2h((u0 - 6.f) * 4.f) | ((unsigned)f2h((u1 - 6.f) * 4.f) << 16);
    }
    hipMemcpy(dr, h, 65536, hipMemcpyHostToDevice);
    unsigned char* src;                       // 4 MB + slack of random halves for the DMA arms
    hipMalloc(&src, (4096 + 64) * 1024);
    for (int i = 0; i < (4096 + 64) / 64; ++i) hipMemcpy(src + (size_t)i * 65536, h, 65536, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 0, 0>("registers, zeros", dz);
        run<8, 0, 0>("registers, random", dr);
        run<4, 0, 0>("registers, random", dr);
        run<8, 1, 0>("LDS fragments, zeros", dz);
        run<8, 1, 0>("LDS fragments, random", dr);
        run<4, 1, 0>("LDS fragments, random", dr);
        run<8, 1, 1>("LDS + barrier/36, random", dr);
        run<8, 1, 0, 1>("LDS random, snake order", dr);
        run<8, 1, 0, 2>("LDS random, A-stationary", dr);
        run<8, 1, 0, 3>("LDS random, B-stationary", dr);
        run<8, 1, 0, 4>("LDS random, Gray path", dr);
        run<8, 0, 0, 2>("registers random, A-stationary", dr);
        run<8, 0, 0, 4>("registers random, Gray path", dr);
        run_wino<8, 0>("Winograd F(2,3) 1x2x4, random", dr);
        run_wino<4, 1>("Winograd F(2,3) 2x2x4, random", dr);
        run_wino<8, 0>("Winograd F(2,3) 1x2x4, zeros", dz);
        run<8, 1, 1, 0, 14>("LDS + barrier + DMA 14/216, random", dr, src);
        run<8, 1, 1, 0, 28>("LDS + barrier + DMA 28/216, random", dr, src);
        run_wino<8, 0, 17>("Winograd 1x2x4 + DMA 17/72", dr, src);
        run_wino<8, 0, 12>("Winograd 1x2x4 + DMA 12/72", dr, src);
        run_wino<8, 0, 9>("Winograd 1x2x4 + DMA 9/72", dr, src);
        run_wino<4, 1, 24>("Winograd 2x2x4 + DMA 24/144", dr, src);
        run_f43<8, 0>("Winograd F(4,3) 1x1x6, no DMA", dr, src);
        run_f43<8, 23>("Winograd F(4,3) 1x1x6 + DMA 23/54", dr, src);
        run_f43<8, 16>("Winograd F(4,3) 1x1x6 + DMA 16/54", dr, src);
    }
    return 0;
}
