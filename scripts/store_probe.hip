// Probe: epilogue store bandwidth, 8 waves per CU each writing a 64 x 64 fp32 tile of an NCHW tensor:
//   V=0  64 x global_store_dword   per wave (lane = pixel, one cout row per instruction: 2 x 128 B segments)  [what the conv epilogues do]
//   V=1  16 x global_store_dwordx4 per wave (lane = 4 pixels of one cout: 8 x 128 B segments per instruction)
//   hipcc --offload-arch=gfx950 -O3 scripts/store_probe.hip -o scripts/store_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int V>
__global__ __launch_bounds__(512) void probe(float* __restrict__ y, int HW, int tiles_per_img) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int img = blockIdx.x / tiles_per_img, tile = blockIdx.x % tiles_per_img;
    // block tile: 64 couts x 512 pixels (8 waves x 64 pixels), Cout = 64
    float* base = y + (size_t)img * 64 * HW + (size_t)tile * 512 + wave * 64;
    const float v = (float)(blockIdx.x + lane);
    if (V == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    base[(size_t)co * HW + n * 32 + l31] = v + r;
                }
    } else {
        // lane -> (cout sub-row j = lane & 3 (+4*hi), pixel quad k = (lane >> 2) & 7)
        const int j = lane & 3, k = (lane >> 2) & 7;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = m * 32 + j + 8 * g + 4 * hi;
                    *reinterpret_cast<float4*>(base + (size_t)co * HW + n * 32 + k * 4) = make_float4(v, v + 1, v + 2, v + g);
                }
    }
}

template <int V>
void run(const char* name, int misalign = 0) {      // misalign: floats added to the base, so every 128-byte run straddles two lines
    const int B = 64, HW = 65536;
    float* y0;
    hipMalloc(&y0, (size_t)B * 64 * HW * 4 + 256);
    float* y = y0 + misalign;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = HW / 512;
    probe<V><<<B * tiles, 512>>>(y, HW, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) probe<V><<<B * tiles, 512>>>(y, HW, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us per 1.07 GB  = %6.2f TB/s\n", name, ms / 5 * 1e3, (double)B * 64 * HW * 4 / (ms / 5 * 1e-3) / 1e12);
    hipFree(y0);
}

int main() {
    run<0>("global_store_dword x64");
    run<1>("global_store_dwordx4 x16");
    run<0>("global_store_dword x64");
    run<0>("dword x64, runs misaligned", 13);
    run<1>("dwordx4 x16, misaligned", 13);
    run<0>("global_store_dword x64");
    return 0;
}
