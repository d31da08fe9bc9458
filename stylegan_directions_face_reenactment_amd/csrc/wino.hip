// Winograd F(2x2, 3x3) form of the shared-weight modulated 3x3 convolution on fp32 MFMA.
//
// The plain 3x3 layers are MFMA-bound at the fp32 matrix rate (157 TFLOP/s), so the only way past that roof is
// fewer multiplies: Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input patch with 16
// multiplies per (cin, cout) instead of 36 -- 2.25x less MFMA work, in exact-coefficient fp32 arithmetic
// (transform matrices hold only 0, +-1, +-1/2).
//
//     U = G g G^T          [Cin][16][Cout]   precomputed per weight version (sgdfr_modconv_prepack_wino_f32)
//     V = B^T (x*s) B      16 values per (cin, tile), computed IN REGISTERS from the staged input patch
//     M_xi[o][t] += U_xi[o][i] * V_xi[i][t]    sixteen independent GEMMs -> 16 MFMA accumulators per wave
//     Y = A^T M A          2x2 outputs per tile, per lane, straight from the accumulators; then the usual
//                          d * Y + noise + bias -> leaky-ReLU epilogue, float2 stores.
//
// Staging is the same padded-flat q-space trick as csrc/modconv.hip: the 4x4 patch of tile (img, ty, tx) starts at
// q = (img*(H+1) + 2ty)*(W+1) + 2tx and its rows are P = W+1 apart, so a block's 64 consecutive tiles need one
// contiguous q-range per channel.  One barrier per K stage (8 input channels = 64 MFMAs per wave), LDS stages
// double-buffered, next stage prefetched into registers under the MFMAs.
#include "common.h"

namespace sgdfr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoParams {
    const float* x;
    int64_t x_bstride;
    const float* u;      // [Cin][16][Cout]
    const float* s;
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    const float* zeros;  // >= 16 B of zeros (source of the padding positions)
    float* y;
    int B, Cin, Cout, H, W, P, R;
    int TW, TH;          // tiles per row / per column (W/2, H/2)
    int n_t_tiles, n_o_tiles;
    int total_tiles;
    int xs, xlen;
    int act;
    float slope, gain;
};

constexpr int WCK = 8;    // input channels per LDS stage
constexpr int WNT = 64;   // couts per block
constexpr int WTT = 64;   // tiles per block
constexpr int WEX = 4;    // staged q elements per thread per channel (xlen <= 1024)

__device__ __forceinline__ int wino_qbase(const WinoParams& p, int t) {
    const int per_img = p.TW * p.TH;
    const int img = t / per_img;
    const int rem = t - img * per_img;
    const int ty = rem / p.TW, tx = rem - ty * p.TW;
    return (img * p.R + 2 * ty) * p.P + 2 * tx;
}

// V = sc * B^T d B for one 4x4 patch (32 add/sub + 16 mul per patch)
__device__ __forceinline__ void wino_input_transform(const float (&dd)[4][4], float sc, float (&vv)[16]) {
    float tmp[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // B^T d  (rows)
        tmp[0][c] = dd[0][c] - dd[2][c];
        tmp[1][c] = dd[1][c] + dd[2][c];
        tmp[2][c] = dd[2][c] - dd[1][c];
        tmp[3][c] = dd[1][c] - dd[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (.) B  (columns), then the style scale
        vv[a * 4 + 0] = (tmp[a][0] - tmp[a][2]) * sc;
        vv[a * 4 + 1] = (tmp[a][1] + tmp[a][2]) * sc;
        vv[a * 4 + 2] = (tmp[a][2] - tmp[a][1]) * sc;
        vv[a * 4 + 3] = (tmp[a][1] - tmp[a][3]) * sc;
    }
}

__global__ __launch_bounds__(256, 1) void wino_mfma_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stage_floats = WCK * p.xs + WCK * 16 * WNT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave >> 1, wt = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W;

    int lid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ct = lid / p.n_t_tiles, tt = lid - ct * p.n_t_tiles;
    const int n0 = ct * WNT;
    const int t0 = tt * WTT;
    const int q0 = wino_qbase(p, t0);

    // this lane's tile (column of the B operand) and its patch origin inside the staged range
    int tile = t0 + wt * 32 + l31;
    const bool tile_ok = tile < p.total_tiles;
    if (!tile_ok) tile = p.total_tiles - 1;
    const int boff = wino_qbase(p, tile) - q0;

    // staging descriptors: padding positions of the q-range read a zero word instead of x
    const float* xsrc[WEX];
    const int nex = (p.xlen + 255) >> 8;
#pragma unroll
    for (int e = 0; e < WEX; ++e) {
        const int j = tid + e * 256;
        const int q = q0 + j;
        const int pir = q / p.P;
        const int pc = q - pir * p.P;
        const int img = pir / p.R;
        const int pr = pir - img * p.R;
        const bool ok = (j < p.xlen) && pc >= 1 && pr >= 1 && img < p.B;
        xsrc[e] = ok ? p.x + (int64_t)img * p.x_bstride + (pr - 1) * p.W + (pc - 1) : nullptr;
    }
    // style of this lane's tile image, for channel (.. + hi) of each k-pair
    const int simg = (wino_qbase(p, tile) / p.P) / p.R;
    const float* sp = p.s + (int64_t)simg * p.Cin + hi;

    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    // Global -> LDS by DMA (no staging registers, no arithmetic: the style scale is applied after the input
    // transform, which is linear).  dwordx4 for the weight slab, dword for the q-range (zero word for padding).
    auto issue_stage = [&](int c0, float* lx, float* lu) {
#pragma unroll
        for (int e = 0; e < WEX; ++e) {
            if (e < nex) {
                const int j0 = (tid & ~63) + e * 256;   // wave-uniform LDS base; the DMA adds lane*4
#pragma unroll
                for (int c = 0; c < WCK; ++c) {
                    const float* src = xsrc[e] ? xsrc[e] + (int64_t)(c0 + c) * HW : p.zeros;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lx + c * p.xs + j0), 4, 0, 0);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int f = tid + v * 256;
            const int row = f >> 4, col = (f & 15) * 4;   // WNT/4 = 16 float4 per row
            const float* src = p.u + ((int64_t)c0 * 16 + row) * p.Cout + n0 + col;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lu + (size_t)((tid & ~63) + v * 256) * 4), 16, 0, 0);
        }
    };

    const int nstage = p.Cin / WCK;
    const int P = p.P;
    issue_stage(0, smem, smem + WCK * p.xs);
    float sv[WCK / 2];
#pragma unroll
    for (int cp = 0; cp < WCK / 2; ++cp) sv[cp] = sp[cp * 2];
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        float* lx = smem + (st & 1) * stage_floats;
        float* lu = lx + WCK * p.xs;
        float sn[WCK / 2];
        if (st + 1 < nstage) {
            float* nx = smem + ((st + 1) & 1) * stage_floats;
            issue_stage((st + 1) * WCK, nx, nx + WCK * p.xs);
#pragma unroll
            for (int cp = 0; cp < WCK / 2; ++cp) sn[cp] = sp[(st + 1) * WCK + cp * 2];
        }
        // Software-pipelined over the 4 k-pairs of the stage: the LDS reads (16 weight fragments + the 4x4 patch)
        // of k-pair n+1 are issued BEFORE the 16 MFMAs of k-pair n, and its input transform
        //     V = s * B^T x B     (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
        // is free to interleave with those MFMAs, so the matrix pipe does not wait on LDS latency.
        float uc[16], vc[16], un[16], dn[4][4];
        {
            const float* px = lx + hi * p.xs + boff;
            const float* pu = lu + (hi * 16) * WNT + wo * 32 + l31;
#pragma unroll
            for (int k = 0; k < 16; ++k) uc[k] = pu[k * WNT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) dn[r][c] = px[r * P + c];
            wino_input_transform(dn, sv[0], vc);
        }
#pragma unroll
        for (int cp = 0; cp < WCK / 2; ++cp) {
            if (cp + 1 < WCK / 2) {
                const float* px = lx + ((cp + 1) * 2 + hi) * p.xs + boff;
                const float* pu = lu + (((cp + 1) * 2 + hi) * 16) * WNT + wo * 32 + l31;
#pragma unroll
                for (int k = 0; k < 16; ++k) un[k] = pu[k * WNT];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) dn[r][c] = px[r * P + c];
                __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[k], vc[k], acc[k], 0, 0, 0);
            if (cp + 1 < WCK / 2) {
                wino_input_transform(dn, sv[cp + 1], vc);
#pragma unroll
                for (int k = 0; k < 16; ++k) uc[k] = un[k];
            }
        }
        if (st + 1 < nstage) {
#pragma unroll
            for (int cp = 0; cp < WCK / 2; ++cp) sv[cp] = sn[cp];
        }
        __syncthreads();   // stage st consumed by every wave; stage st+1's DMA has landed (vmcnt(0) before the barrier)
    }

    // ---- output transform Y = A^T M A per lane, then the StyledConv epilogue
    if (!tile_ok) return;
    const int per_img = p.TW * p.TH;
    const int64_t img = tile / per_img;
    const int rem = tile - (int)img * per_img;
    const int ty = rem / p.TW, tx = rem - ty * p.TW;
    const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
    float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.noise) {
        const float* np = p.noise + img * p.noise_bstride + (2 * ty) * p.W + 2 * tx;
        nz[0][0] = nw * np[0]; nz[0][1] = nw * np[1];
        nz[1][0] = nw * np[p.W]; nz[1][1] = nw * np[p.W + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = n0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= p.Cout) continue;
        float m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = acc[k][r];
        float ra[2][4];   // A^T M  (rows): [0] = m0+m1+m2, [1] = m1-m2-m3, per column
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ra[0][c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
            ra[1][c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
        }
        float yv[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            yv[a][0] = ra[a][0] + ra[a][1] + ra[a][2];
            yv[a][1] = ra[a][1] - ra[a][2] - ra[a][3];
        }
        const float dv = p.d ? p.d[img * p.Cout + co] : 1.f;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float* dst = p.y + (img * p.Cout + co) * HW + (2 * ty) * p.W + 2 * tx;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v0 = yv[a][0] * dv + nz[a][0] + bv;
            float v1 = yv[a][1] * dv + nz[a][1] + bv;
            if (p.act) {
                v0 = lrelu_gain(v0, p.slope, p.gain);
                v1 = lrelu_gain(v1, p.slope, p.gain);
            }
            *reinterpret_cast<float2*>(dst + a * p.W) = make_float2(v0, v1);
        }
    }
}

// U[i][xi][o] = (G g G^T)[xi] * scale,  g = weight[o][i][3][3]  (optionally rotated 180 degrees and read
// transposed, which turns the pack into the one of the adjoint conv: weight is then indexed [i][o])
__global__ __launch_bounds__(256) void prepack_wino_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout,
                                                          int Cin, int transpose_flip, float scale) {
    // output dims: rows = "in" channels of this conv, cols = "out" channels
    const int n_in = transpose_flip ? Cout : Cin, n_out = transpose_flip ? Cin : Cout;
    const int64_t n = (int64_t)n_in * n_out;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % n_out);
        const int i = (int)(idx / n_out);
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                g[ky][kx] = (transpose_flip ? w[((int64_t)i * Cin + o) * 9 + (2 - ky) * 3 + (2 - kx)]
                                            : w[((int64_t)o * Cin + i) * 9 + ky * 3 + kx]) * scale;
        float t[4][3];   // G g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t[0][c] = g[0][c];
            t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
            t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
            t[3][c] = g[2][c];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0];
            const float u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
            const float u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
            const float u3 = t[a][2];
            float* dst = u + ((int64_t)i * 16 + a * 4) * n_out + o;
            dst[0] = u0; dst[n_out] = u1; dst[2 * (int64_t)n_out] = u2; dst[3 * (int64_t)n_out] = u3;
        }
    }
}

}  // namespace sgdfr

using namespace sgdfr;

// Largest q-distance between the first and last patch origin of 64 consecutive tiles starting at a multiple of 64.
static int wino_span(int H, int W) {
    const int TW = W / 2, TH = H / 2, P = W + 1;
    const int per_img = TW * TH;
    if (TW % WTT == 0) return 2 * (WTT - 1);                                            // inside one tile row
    if (WTT % TW == 0 && per_img % WTT == 0) return (WTT / TW - 1) * 2 * P + 2 * (TW - 1);  // whole tile rows of one image
    const int rows_crossed = (WTT - 1) / TW + 1, imgs_crossed = (WTT - 1) / per_img + 1;
    return rows_crossed * 2 * P + imgs_crossed * P + 2 * (TW - 1);   // a row step is 2P, an image step 3P
}

extern "C" int sgdfr_modconv_prepack_wino_f32(const float* weight, float* u, int Cout, int Cin, int transpose_flip,
                                              void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0, "prepack_wino: bad shape %d %d", Cout, Cin);
    SGDFR_REQUIRE(weight && u, "prepack_wino: null pointer");
    const int64_t n = (int64_t)Cout * Cin;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prepack_wino_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, u, Cout, Cin,
                       transpose_flip, 1.0f / sqrtf((float)Cin * 9));
    return check_launch("modconv_prepack_wino");
}

extern "C" int sgdfr_modconv2d_wino_supported(int B, int Cin, int Cout, int H, int W) {
    if (Cin % WCK != 0 || Cout % WNT != 0 || (H & 1) || (W & 1) || H < 2 || W < 2) return 0;
    // staged q-range of 64 consecutive tiles must fit WEX*256 elements
    return (wino_span(H, W) + 3 * (W + 1) + 4) <= WEX * 256 ? 1 : 0;
}

extern "C" int sgdfr_modconv2d_wino_f32(const float* x, int64_t x_bstride, const float* u, const float* s, const float* d,
                                        const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                                        const float* zeros, float* y, int B, int Cin, int Cout, int H, int W, int act,
                                        float slope, float gain, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv_wino: bad shape B=%d Cin=%d Cout=%d H=%d W=%d",
                  B, Cin, Cout, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(sgdfr_modconv2d_wino_supported(B, Cin, Cout, H, W), "modconv_wino: shape not supported "
                  "(needs Cin %% 8 == 0, Cout %% 64 == 0, even H and W); use sgdfr_modconv2d_fwd_f32");
    SGDFR_REQUIRE(x && u && s && y && zeros, "modconv_wino: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "modconv_wino: noise without noise_w");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  "modconv_wino: s, u and y must be 16-byte aligned");
    WinoParams p{};
    p.x = x; p.x_bstride = x_bstride; p.u = u; p.s = s; p.d = d; p.noise = noise; p.noise_bstride = noise_bstride;
    p.noise_w = noise_w; p.bias = bias; p.zeros = zeros; p.y = y;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    p.TW = W / 2; p.TH = H / 2;
    p.total_tiles = B * p.TW * p.TH;
    SGDFR_REQUIRE((int64_t)B * p.R * p.P + 4ll * p.P + 8 < (1ll << 31), "modconv_wino: batch too large for 32-bit indices");
    p.n_t_tiles = (p.total_tiles + WTT - 1) / WTT;
    p.n_o_tiles = Cout / WNT;
    p.xlen = wino_span(H, W) + 3 * p.P + 4;
    p.xs = (p.xlen + 255) & ~255;   // whole 256-element DMA rows: stray lanes of the last row stay inside the channel
    p.act = act; p.slope = slope; p.gain = gain;
    const size_t lds = 2 * (size_t)(WCK * p.xs + WCK * 16 * WNT) * sizeof(float);
    SGDFR_REQUIRE(lds <= 160 * 1024, "modconv_wino: LDS request %zu too large", lds);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return check_launch("modconv_wino(lds attribute)");
    hipLaunchKernelGGL(wino_mfma_kernel, dim3(p.n_t_tiles * p.n_o_tiles), dim3(256), lds, as_stream(stream), p);
    return check_launch("modconv2d_wino");
}
