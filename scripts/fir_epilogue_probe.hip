// Prices the EPILOGUE of a fused up-layer (transposed conv + 4x4 FIR blur + noise / bias / leaky-ReLU + split hand-over in one kernel,
// DESIGN 4.10 "what comes next") before anyone writes its K loop again: experiments/csrc/upfir.hip measured 0.91 ms for this part alone on
// the 128 -> 256 level (B = 64) against 0.52 ms for the whole stand-alone blur kernel.  Here the same work in the form a rewrite would
// use, per 16 x 16 patch of super-pixels (256 MFMA columns, 14 x 14 of them interior) and 64 couts:
//   * the four parity phases of a super-pixel go to LDS as one 16-byte quad, sixteen couts at a time ([cout 16][position 256][4]: 64 KB),
//   * one thread = (interior super-pixel, 8-channel group): 3 x 3 quads per channel -> the separable [1 3 3 1] (x) [1 3 3 1] / 16 FIR
//     (vertical pass, then horizontal: 8 FMAs per output instead of 16) -> bias + leaky-ReLU -> x next style -> hi / lo fp16 ->
//     eight 16-byte stores (2 x 2 output pixels x hi, lo) into the next conv's split input form.
// Reports us per launch for the 128 -> 256 level's tile count and the implied clocks per tile, with and without the stores.
//   hipcc --offload-arch=gfx950 -O3 scripts/fir_epilogue_probe.hip -o build/fir_epilogue_probe && build/fir_epilogue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a - hf[0], b - hf[1]}, f16x2));
}

// STORES: 0 = results only summed (keeps the arithmetic alive), 1 = hand-over stores
template <int STORES>
__global__ __launch_bounds__(512, 1) void fir_epilogue(const float* __restrict__ bias, const float* __restrict__ snext, unsigned char* __restrict__ xs,
                                                      int tiles, int img_w, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* const T = reinterpret_cast<float4*>(smem);                       // [cout 16][position 256] quads (px0py0, px0py1, px1py0, px1py1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc_sum = 0.f;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        for (int slice = 0; slice < 4; ++slice) {                          // sixteen couts at a time
            // "accumulators" of this wave's MFMA tiles: 32 couts x 64 positions x 4 phases per wave in the real kernel; here every thread
            // writes the 8 quads that would be its share of the slice (16 couts x 256 positions / 512 threads), values from registers
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = k * 512 + tid;                              // (cout, position)
                const float v = (float)((idx * 2654435761u >> 20) & 1023) * (1.f / 512.f) - 1.f + (float)(tile & 7) * 0.01f;
                T[idx] = make_float4(v, v * 0.5f, -v, v + 0.25f);
            }
            __syncthreads();
            // 14 x 14 interior super-pixels x 2 channel groups of 8 = 392 items for 512 threads
            if (tid < 392) {
                const int grp = tid / 196, sp = tid - grp * 196;
                const int a = 1 + sp / 14, b = 1 + sp % 14;                 // position inside the 16 x 16 patch
                float out[2][2][8];                                         // [row of the quad][pixel of the quad][channel]
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4* tc = T + (grp * 8 + c) * 256;
                    // T rows 2a-1 .. 2a+3, columns 2b-1 .. 2b+3 of this channel: a 5 x 5 window out of the 3 x 3 quads
                    float w[5][5];
#pragma unroll
                    for (int da = -1; da <= 1; ++da)
#pragma unroll
                        for (int db = -1; db <= 1; ++db) {
                            const float4 q = tc[(a + da) * 16 + (b + db)];
                            // quad (px, py): rows 2(a+da)+py, columns 2(b+db)+px
#pragma unroll
                            for (int py = 0; py < 2; ++py)
#pragma unroll
                                for (int px = 0; px < 2; ++px) {
                                    const int r = 2 * da + py + 1, cc = 2 * db + px + 1;      // window coordinates 0..4 wanted
                                    if (r >= 0 && cc >= 0) w[r][cc] = px == 0 ? (py == 0 ? q.x : q.y) : (py == 0 ? q.z : q.w);
                                }
                        }
                    // output (2a + r, 2b + s), r, s in 0..1 = taps [1 3 3 1] / 4 over window rows r .. r+3, then over columns s .. s+3
                    // (the up-layer's gain of 4 folded in)
                    float vcol[2][5];
#pragma unroll
                    for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
                            vcol[r][cc] = fmaf(0.25f, w[r][cc], fmaf(0.75f, w[r + 1][cc], fmaf(0.75f, w[r + 2][cc], 0.25f * w[r + 3][cc])));
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2)
                            out[r][s2][c] = fmaf(0.25f, vcol[r][s2], fmaf(0.75f, vcol[r][s2 + 1], fmaf(0.75f, vcol[r][s2 + 2], 0.25f * vcol[r][s2 + 3])));
                }
                const int cg = slice * 2 + grp;                              // 8-channel group of the 64 couts
                const float4 b0 = *reinterpret_cast<const float4*>(bias + cg * 8), b1 = *reinterpret_cast<const float4*>(bias + cg * 8 + 4);
                const float4 s0 = *reinterpret_cast<const float4*>(snext + cg * 8), s1 = *reinterpret_cast<const float4*>(snext + cg * 8 + 4);
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const int tpr = img_w / 28;                                  // patches per image row (14 super-pixels = 28 output pixels)
                const int ty = (tile / tpr), tx = tile - ty * tpr;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        unsigned h[4], l[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v0 = out[r][s][2 * c] + bv[2 * c], v1 = out[r][s][2 * c + 1] + bv[2 * c + 1];
                            v0 = (v0 > 0.f ? v0 : 0.2f * v0) * 1.41421356f * sv[2 * c];
                            v1 = (v1 > 0.f ? v1 : 0.2f * v1) * 1.41421356f * sv[2 * c + 1];
                            split2(v0, v1, h[c], l[c]);
                        }
                        if (STORES) {
                            const size_t oy = (size_t)(ty * 28 + 2 * (a - 1) + r), ox = (size_t)(tx * 28 + 2 * (b - 1) + s);
                            const size_t hw = (size_t)img_w * img_w * 64;      // (64 images' worth of rows: the tile index runs over all of them)
                            unsigned char* dst = xs + ((size_t)cg * 2 * hw + (oy % ((size_t)img_w * 64)) * img_w + ox) * 16;
                            *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
                            *reinterpret_cast<uint4*>(dst + hw * 16) = make_uint4(l[0], l[1], l[2], l[3]);
                        } else {
                            acc_sum += __builtin_bit_cast(float, h[0] ^ h[1] ^ h[2] ^ h[3] ^ l[0] ^ l[1] ^ l[2] ^ l[3]) * 1e-30f;
                        }
                    }
            }
        }
    }
    if (!STORES) sink[blockIdx.x * 512 + tid] = acc_sum;
}

template <int STORES>
static void run(const char* name, const float* bias, const float* sn, unsigned char* xs, int tiles, int img_w, float* sink) {
    hipFuncSetAttribute((const void*)fir_epilogue<STORES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fir_epilogue<STORES><<<256, 512, 65536>>>(bias, sn, xs, tiles, img_w, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        fir_epilogue<STORES><<<256, 512, 65536>>>(bias, sn, xs, tiles, img_w, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-26s %d tiles (64 couts x 14 x 14 super-pixels): %8.1f us per launch, %6.2f us per tile and CU\n", name, tiles, best * 1e3,
           best * 1e3 / ((tiles + 255) / 256));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int img_w = 252;                        // 9 patches of 28 output pixels per row (the real level: 256)
    const int tiles = 64 * 9 * 9;                 // B = 64 images of the 128 -> 256 level, one cout tile (Cout = 64)
    float *bias, *sn, *sink; unsigned char* xs;
    hipMalloc(&bias, 64 * 4); hipMalloc(&sn, 64 * 4); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(bias, 0, 256); hipMemset(sn, 0x3c, 256);
    hipMalloc(&xs, (size_t)8 * 2 * img_w * img_w * 64 * 16 + (1 << 20));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("LDS + FIR + split, no stores", bias, sn, xs, tiles, img_w, sink);
        run<1>("... + hand-over stores", bias, sn, xs, tiles, img_w, sink);
    }
    printf("for scale: the stand-alone blur kernel of this level takes ~517 us, the transposed conv ~470 us (B = 64, profiles/r05_d_*)\n");
    return 0;
}
