// Probe for the split-bf16 (bf16x3) conv design: what does v_mfma_f32_32x32x16_bf16 sustain when every operand
// comes from LDS (ds_read_b128, conflict-free [k-half][row][8] layout) with an MI x NI register tile and 3 products
// (hi*hi, hi*lo, lo*hi) per tile, optionally with fp32 VALU filler (the staging split) in the same wave?
//   hipcc --offload-arch=gfx950 -O3 scripts/bf16_probe.hip -o gpurun_out/bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MI, int NI, int WAVES, int FILL>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 3);
    __syncthreads();
    f32x16 acc[MI][NI];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    // A: [part 2][khalf 2][64 rows][8] bf16 per tap (4 KB per part at 64 rows); B: [part][khalf][positions][8]
    const unsigned char* pa = lds + ((lane >> 5) * 64 + (lane & 31)) * 16;
    const unsigned char* pb = lds + 32768 + ((lane >> 5) * 512 + (wave & 3) * 64 + (lane & 31)) * 16;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = lane + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(pa + tap * 4096 + (m & 1) * 512 + (it & 1) * 2048);
                al[m] = *reinterpret_cast<const bf16x8*>(pa + 16384 + tap * 4096 + (m & 1) * 512 + (it & 1) * 2048);
            }
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                bh[n] = *reinterpret_cast<const bf16x8*>(pb + (tap + (n & 1) * 32) * 16);
                bl[n] = *reinterpret_cast<const bf16x8*>(pb + 16384 + (tap + (n & 1) * 32) * 16);
            }
#pragma unroll
            for (int j = 0; j < FILL; ++j) f[j & 7] = f[j & 7] * 1.0001f + 0.5f;
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MI, int NI, int WAVES, int FILL>
void run(const char* name) {
    float* out;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<MI, NI, WAVES, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    probe<MI, NI, WAVES, FILL><<<blocks, WAVES * 64, 65536>>>(out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MI, NI, WAVES, FILL><<<blocks, WAVES * 64, 65536>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 3 * MI * NI * 3 * 2.0 * 32 * 32 * 16;
    printf("%-40s MIxNI %dx%d waves %d fill %3d: %8.3f ms  %7.1f bf16 TFLOP/s = %6.1f fp32-equivalent\n", name, MI, NI, WAVES,
           FILL, ms, flops / ms / 1e9, flops / ms / 1e9 / 3);
    hipFree(out);
}

int main() {
    run<2, 2, 4, 0>("LDS operands");
    run<2, 4, 4, 0>("LDS operands");
    run<4, 4, 4, 0>("LDS operands");
    run<2, 2, 8, 0>("LDS operands");
    run<2, 4, 8, 0>("LDS operands");
    run<2, 2, 4, 16>("+ VALU filler");
    run<2, 2, 4, 48>("+ VALU filler");
    run<2, 4, 4, 48>("+ VALU filler");
    run<2, 2, 8, 48>("+ VALU filler");
    run<2, 4, 8, 96>("+ VALU filler");
    return 0;
}
