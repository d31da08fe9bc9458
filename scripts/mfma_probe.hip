// Probe: what does one wave per SIMD sustain on v_mfma_f32_32x32x2_f32 with 16 accumulators when LDS reads /
// VALU fillers / barriers are added?  hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT, int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(float* out, int iters, int P) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = lane * 0.01f + k; b[k] = lane * 0.02f - k; }
    const float* pu = lds + (lane & 31) * 16 + (tid >> 6) * 1024;
    const float* px = lds + 8192 + (lane & 31) * 2 + (lane >> 5) * 512;
    for (int it = 0; it < iters; ++it) {
        float un[16], dn[4][4];
        if (VARIANT >= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(pu + ((j ^ (lane & 3)) << 2) + (it & 3) * 512);
                un[4 * j] = q.x; un[4 * j + 1] = q.y; un[4 * j + 2] = q.z; un[4 * j + 3] = q.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 x0 = *reinterpret_cast<const float2*>(px + r * P + (it & 3) * 64);
                const float2 x1 = *reinterpret_cast<const float2*>(px + r * P + 2 + (it & 3) * 64);
                dn[r][0] = x0.x; dn[r][1] = x0.y; dn[r][2] = x1.x; dn[r][3] = x1.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[k % NACC], 0, 0, 0);
        if (VARIANT >= 1) __builtin_amdgcn_sched_barrier(0);
        float vn[16];
        if (VARIANT >= 2) {
            float tmp[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tmp[0][c] = dn[0][c] - dn[2][c]; tmp[1][c] = dn[1][c] + dn[2][c];
                tmp[2][c] = dn[2][c] - dn[1][c]; tmp[3][c] = dn[1][c] - dn[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vn[r * 4 + 0] = (tmp[r][0] - tmp[r][2]) * 1.01f; vn[r * 4 + 1] = (tmp[r][1] + tmp[r][2]) * 1.01f;
                vn[r * 4 + 2] = (tmp[r][2] - tmp[r][1]) * 1.01f; vn[r * 4 + 3] = (tmp[r][1] - tmp[r][3]) * 1.01f;
            }
        } else if (VARIANT >= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) vn[r * 4 + c] = dn[r][c];
        }
#pragma unroll
        for (int k = 4; k < 16; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[k % NACC], 0, 0, 0);
        if (VARIANT >= 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a[k] = un[k]; b[k] = vn[k]; }
        }
        if (VARIANT >= 3 && (it & 3) == 3) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

// Waves 0-3: MFMA only.  Waves 4-7 (same SIMDs): FILL = 1 fp32 VALU chain, 2 LDS reads, 3 global loads -- does a
// second wave's non-matrix work overlap with fp32 MFMAs of its SIMD partner?
template <int FILL>
__global__ __launch_bounds__(512, 1) void probe_split(float* out, const float* gsrc, int iters, unsigned long long* gticks) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        const float a = lane * 0.01f, b = lane * 0.02f;
        const unsigned long long t0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
        }
        const unsigned long long t1 = wall_clock64();
        if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(gticks), t1 - t0);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[k][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lane + j;
        // roughly iters*16*64 cycles of partner MFMA time; fill it with ~the same number of filler instructions
        for (int it = 0; it < iters * 16; ++it) {
            if (FILL == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * 1.0001f + 0.5f;
            } else if (FILL == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += lds[(lane * 4 + j * 256 + it) & 8191];
            } else if (FILL == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += gsrc[((size_t)blockIdx.x * 4096 + lane + j * 64 + (it & 15) * 256) & 1048575];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
        out[blockIdx.x * 512 + tid] = s;
    }
}

template <int FILL>
void run_split(const char* name) {
    float *out, *src;
    const int blocks = 256, iters = 4000;
    hipMalloc(&out, sizeof(float) * blocks * 512);
    hipMalloc(&src, sizeof(float) * 1048576);
    hipMemset(src, 0, sizeof(float) * 1048576);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* ticks;
    hipMalloc(&ticks, 8);
    probe_split<FILL><<<blocks, 512>>>(out, src, 100, ticks);
    hipDeviceSynchronize();
    hipMemset(ticks, 0, 8);
    hipEventRecord(e0);
    probe_split<FILL><<<blocks, 512>>>(out, src, iters, ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
    unsigned long long h = 0;
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double mfma_ms = (double)h * 1e-5;   // wall_clock64 ticks at 100 MHz
    printf("%-60s kernel %7.3f ms, slowest MFMA wave %7.3f ms -> %6.1f TFLOP/s while the partner runs\n", name, ms, mfma_ms,
           flops / mfma_ms / 1e9);
    hipFree(out); hipFree(src);
}

template <int VARIANT, int NACC, int WAVES>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    hipMalloc(&out, sizeof(float) * blocks * WAVES * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VARIANT, NACC, WAVES><<<blocks, WAVES * 64>>>(out, 100, 66);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VARIANT, NACC, WAVES><<<blocks, WAVES * 64>>>(out, iters, 66);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 16 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks/CU %d waves/blk %d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, WAVES, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<0, 16, 4>("V0 mfma only, 16 acc", 1);
    run<0, 4, 4>("V0 mfma only, 4 acc", 1);
    run<0, 4, 4>("V0 mfma only, 4 acc", 2);
    run<1, 16, 4>("V1 + LDS fragment reads (b128/b64)", 1);
    run<2, 16, 4>("V2 + input-transform VALU", 1);
    run<3, 16, 4>("V3 + barrier / 64 MFMAs", 1);
    run<2, 4, 4>("V2 4 acc, 2 blocks/CU", 2);
    run<3, 4, 4>("V3 4 acc, 2 blocks/CU", 2);
    run<3, 4, 4>("V3 4 acc, 3 blocks/CU", 3);
    run_split<0>("split: 4 MFMA waves + 4 idle waves");
    run_split<1>("split: 4 MFMA waves + 4 fp32-VALU waves (8 fma / 64 cyc)");
    run_split<2>("split: 4 MFMA waves + 4 LDS-read waves");
    run_split<3>("split: 4 MFMA waves + 4 global-load waves");
    return 0;
}
