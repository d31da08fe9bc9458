// Prices candidate FORMS of the F(4,3) split-Winograd K loop (csrc/wsplit.hip) before any of them is written as a kernel
// (round 5: VERDICT r4 items 1 and 4).  Same method as scripts/mfma16_probe.hip: a bare loop at each form's instruction ratios --
// v_mfma_f32_32x32x16_f16 on random operands, fragments read from LDS with ds_read_b128, operand DMA with global_load_lds
// (1 KB pieces from an L2-resident source), optionally VALU filler and global stores standing for a deferred epilogue --
// timed on all 256 CUs, with the shader clock (s_memtime) beside the wall time.
//
//   form                              waves  wave tile (couts x tiles)   reads / MFMAs per position   DMA pieces per wave and channel block
//   today (wsplit_kernel<.,6>)          8      32 x 32   (MI 1, NI 1)       4 / 3                        21.75  per 54 MFMAs
//   item 1: 4 waves, 2 acc sets         4      64 x 32   (MI 2, NI 1)       6 / 6                        43.5   per 108
//   128 x 128 block, 8 waves            8      64 x 32 or 32 x 64           6 / 6                        25     per 108
//   128 x 128 block, 4 waves            4      64 x 64   (MI 2, NI 2)       8 / 12                       50     per 216
//
//   hipcc --offload-arch=gfx950 -O3 scripts/wsplit_form_probe.hip -o gpurun_out/wsplit_form_probe && gpurun_out/wsplit_form_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int frag128 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// DMA: pieces per wave per channel block (18 steps); FILL10: VALU filler instructions per MFMA x 10; ST: 16-byte global stores per
// wave and channel block (the deferred epilogue's hand-over stores); BAR: s_barrier every 3 steps (a half-stage)
template <int WAVES, int MI, int NI, int DMA, int FILL10, int ST, int BAR>
__global__ __launch_bounds__(WAVES * 64, 1) void form_probe(float* out, const unsigned* seed, int iters, unsigned long long* clk, const unsigned char* src,
                                                           uint4* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = seed[i];
    __syncthreads();
    f32x16 acc[6][MI][NI];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][m][n][r] = 0.f;
    const unsigned char* pa = lds + ((lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave >> 1) * 64 + (lane & 31)) * 16;
    frag128 a[2][MI], b[2][NI];
    float fill[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) fill[i] = (float)(lane + i);
    uint4* my_sink = sink + ((size_t)blockIdx.x * WAVES + wave) * 64 * 64 + lane;
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
#pragma unroll
                for (int m = 0; m < MI; ++m)
                    a[part][m] = *reinterpret_cast<const frag128*>(pa + part * 16384 + (step % 3) * 4096 + m * 512 + (it & 1) * 2048);
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    b[part][n] = *reinterpret_cast<const frag128*>(pb + part * 16384 + (step + n * 32 + (it & 7)) * 16);
            }
            int k = 0;
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int n = 0; n < NI; ++n, ++k) {
                        acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[term == 2][m]),
                                                                             __builtin_bit_cast(f16x8, b[term == 1][n]), acc[t][m][n], 0, 0, 0);
                        if (FILL10) {       // independent fma chains: stand-ins for the deferred epilogue's scale / lrelu / split VALU
                            const int g0 = ((step * 3 * MI * NI + k) * FILL10) / 10, g1 = ((step * 3 * MI * NI + k + 1) * FILL10) / 10;
#pragma unroll
                            for (int g = g0; g < g1; ++g) fill[g & 7] = __builtin_fmaf(fill[g & 7], 1.0001f, fill[(g + 3) & 7]);
                        }
                    }
            if (ST) {
                const int n0 = step * ST / 18, n1 = (step + 1) * ST / 18;
#pragma unroll
                for (int v = n0; v < n1; ++v)
                    my_sink[((it * ST + v) & 63) * 64] = make_uint4(__builtin_bit_cast(unsigned, fill[0]), __builtin_bit_cast(unsigned, fill[1]),
                                                                    __builtin_bit_cast(unsigned, fill[2]), (unsigned)it);
            }
            if (BAR && step % 3 == 2) __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[t][m][n][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += fill[i] * 1e-30f;
    out[blockIdx.x * blockDim.x + tid] = s;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

template <int WAVES, int MI, int NI, int DMA, int FILL10 = 0, int ST = 0, int BAR = 1>
void run(const char* name, const unsigned* seed_dev, const unsigned char* src, uint4* sink, double alg_per_issued = 2.0) {
    float* out; unsigned long long* clk;
    const int blocks = 256, iters = 3000 / (MI * NI);
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64); hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = form_probe<WAVES, MI, NI, DMA, FILL10, ST, BAR>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    kern<<<blocks, WAVES * 64, 131072>>>(out, seed_dev, 100, clk, src, sink);
    hipDeviceSynchronize();
    float best = 1e30f; unsigned long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kern<<<blocks, WAVES * 64, 131072>>>(out, seed_dev, iters, clk, src, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); }
    }
    const double mfmas = (double)iters * 54 * MI * NI;          // per wave
    const double flops = (double)blocks * WAVES * mfmas * 2.0 * 32 * 32 * 16;
    const double alg = flops / 3 * alg_per_issued;               // F(4,3): half the MFMA work of the direct form (2.0), 3 products per fp32 product
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
    printf("%-44s waves %d tile %dx%d regs %3d: %8.3f ms %7.1f TF 16-bit issued = %6.1f algorithmic fp32 TF, clock %.2f GHz, MFMA duty %.2f, %5.1f clk/MFMA/SIMD\n",
           name, WAVES, 32 * MI, 32 * NI, fa.numRegs, best, flops / best / 1e9, alg / best / 1e9, c / (best * 1e6), mfmas * 32 * (WAVES / 4) / (double)c,
           (double)c / (mfmas * (WAVES / 4)));
    hipFree(out); hipFree(clk);
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }

int main() {
    unsigned* h = (unsigned*)malloc(65536);
    unsigned* dr;
    hipMalloc(&dr, 65536);
    srand(1);
    for (int i = 0; i < 16384; ++i) {
        float u0 = 0, u1 = 0;
        for (int k = 0; k < 12; ++k) { u0 += rand() / (float)RAND_MAX; u1 += rand() / (float)RAND_MAX; }
        h[i] = f2h((u0 - 6.f) * 4.f) | ((unsigned)f2h((u1 - 6.f) * 4.f) << 16);
    }
    hipMemcpy(dr, h, 65536, hipMemcpyHostToDevice);
    unsigned char* src;                       // 4 MB + slack of random halves for the DMA arms (L2-resident per XCD)
    hipMalloc(&src, (4096 + 64) * 1024);
    for (int i = 0; i < (4096 + 64) / 64; ++i) hipMemcpy(src + (size_t)i * 65536, h, 65536, hipMemcpyHostToDevice);
    uint4* sink;
    hipMalloc(&sink, sizeof(uint4) * 256 * 8 * 64 * 64);
    for (int rep = 0; rep < 2; ++rep) {
        printf("--- round %d\n", rep);
        // today's form
        run<8, 1, 1, 0>("today 8w 32x32, no DMA", dr, src, sink);
        run<8, 1, 1, 22>("today 8w 32x32 + DMA 22/54", dr, src, sink);
        // item 1: four waves, 64 x 32 per wave, same block tile (same DMA bytes per MFMA = 44 pieces per 108 MFMAs per wave)
        run<4, 2, 1, 0>("item1 4w 64x32, no DMA", dr, src, sink);
        run<4, 2, 1, 44>("item1 4w 64x32 + DMA 44/108", dr, src, sink);
        run<4, 2, 1, 44, 14, 4>("item1 4w 64x32 + DMA + filler 1.4 + 4 st", dr, src, sink);
        run<4, 2, 1, 44, 28, 8>("item1 4w 64x32 + DMA + filler 2.8 + 8 st", dr, src, sink);
        run<4, 1, 2, 44>("item1' 4w 32x64 + DMA 44/108", dr, src, sink);
        // the same wave tile at two waves per SIMD = a 128 x 128 block tile: half the U bytes per MFMA (25 pieces per 108 MFMAs)
        run<8, 2, 1, 0>("128x128 8w 64x32, no DMA", dr, src, sink);
        run<8, 2, 1, 25>("128x128 8w 64x32 + DMA 25/108", dr, src, sink);
        run<8, 1, 2, 25>("128x128 8w 32x64 + DMA 25/108", dr, src, sink);
        // ... and with four waves of 64 x 64
        run<4, 2, 2, 0>("128x128 4w 64x64, no DMA", dr, src, sink);
        run<4, 2, 2, 50>("128x128 4w 64x64 + DMA 50/216", dr, src, sink);
        // Round 6: the TRANSPOSED conv (split_kernel.h deep plan: block 64 couts x 256 super-pixels, 8 waves of
        // 32 x 64, all nine taps of a channel block per stage) as a 1-D Winograd form along x -- the even output phase is a 2-tap
        // correlation (kx in {0, 2}), the odd phase 1 tap -- priced at each form's operand bytes per MFMA.  Per block and 16-channel block:
        //   direct      9 taps: W 36 KB + X 16 KB per 432 MFMAs (120 B / MFMA)                              -> 13 pieces per 108 MFMAs of a wave
        //   F(2,2) on x 7.5 taps: U (3 + 1) / 3 x 36 KB + V (3 + 2) / 2 x 16 KB per 360 MFMAs (244 B / MFMA)   -> 26 per 108
        //   F(4,2) on x 6.75 taps: U (5 + 1) / 3 x 36 KB + V (5 + 4) / 4 x 16 KB per 324 MFMAs (333 B / MFMA)  -> 36 per 108
        // (a Winograd position has its own U_t AND its own V_t: the nine taps of the direct form share ONE staged x image)
        run<8, 1, 2, 0>("up direct 8w 32x64, no DMA", dr, src, sink, 1.0);
        run<8, 1, 2, 13>("up direct 8w 32x64 + DMA 13/108", dr, src, sink, 1.0);
        run<8, 1, 2, 26>("up F(2,2)x 8w 32x64 + DMA 26/108", dr, src, sink, 9.0 / 7.5);
        run<8, 1, 2, 36>("up F(4,2)x 8w 32x64 + DMA 36/108", dr, src, sink, 9.0 / 6.75);
        // ... the output transform A^T M of the even phase folded into the loop (wswide's position-outer arrangement): ~0.5 VALU per MFMA
        run<8, 1, 2, 36, 5>("up F(4,2)x + DMA 36/108 + filler 0.5", dr, src, sink, 9.0 / 6.75);
        // what the filler costs the 8-wave form (the deferred epilogue is a 4-wave idea; for reference)
        run<8, 1, 1, 22, 14, 2>("today 8w + DMA + filler 1.4 + 2 st", dr, src, sink);
    }
    return 0;
}
