// Small dense layers of the path: mapping-network EqualLinear (+ fused leaky-ReLU), the per-layer
// style modulation, the demodulation coefficients and DirectionMatrix.  M = batch (1..512),
// N, K <= 4096: a few hundred MFLOP per forward against ~2 TFLOP of convolution, so these are
// LDS-tiled fp32 VALU kernels (exact fmaf chains) sized for launch latency, not MFMA.
#include <stdlib.h>

#include "common.h"

namespace sgdfr {

constexpr int LBK = 32;

// One BM x BN output tile of  y = epi(x[M,K] @ w[N,K]^T)  by a 256-thread block (16 x 16 threads, each
// (BM/16) x (BN/16) outputs).  Tiles of x and w are staged k-major in LDS (lanes along k for the global
// reads: 128-byte rows; padded rows make the transposed LDS writes conflict-free).
// EPI 0: y = act(acc*wscale + bias*bscale);  EPI 1: y = rsqrt(acc + wscale)  (wscale carries eps)
// EPI 2: y = add[m,n] + mul[m,n] * (acc*wscale)          (gradient of the demodulation w.r.t. the style)
// XOP 0: x ; 1: x^2 (demodulation: sum_i s^2 q) ; 2: -x * aux^3 (dL/dd -> dL/d(sum s^2 q) up to the factor 2s)
struct LinearExtra {
    const float* xaux;   // [M, K] (ld = ldx) for XOP 2
    const float* add;    // [M, N] (ld = ldy) for EPI 2
    const float* mul;    // [M, N] (ld = ldy) for EPI 2
};

template <int XOP, int EPI, int BM, int BN>
__device__ __forceinline__ void linear_tile(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ y, int64_t ldy, int M,
                                            int N, int K, int m0, int n0, float wscale, float bscale, int act,
                                            float slope, float gain, float (*xs)[BM + 1], float (*ws)[BN + 1],
                                            LinearExtra ex = LinearExtra{nullptr, nullptr, nullptr}) {
    constexpr int TM = BM / 16, TN = BN / 16;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[TM][TN] = {};
    for (int k0 = 0; k0 < K; k0 += LBK) {
        for (int e = tid; e < BM * LBK; e += 256) {
            const int kk = e & (LBK - 1), mm = e >> 5;
            float v = 0.f;
            if (m0 + mm < M && k0 + kk < K) {
                v = x[(int64_t)(m0 + mm) * ldx + k0 + kk];
                if (XOP == 1) v = v * v;
                if (XOP == 2) {
                    const float a = ex.xaux[(int64_t)(m0 + mm) * ldx + k0 + kk];
                    v = -v * a * a * a;
                }
            }
            xs[kk][mm] = v;
        }
        for (int e = tid; e < BN * LBK; e += 256) {
            const int kk = e & (LBK - 1), nn = e >> 5;
            float v = 0.f;
            if (n0 + nn < N && k0 + kk < K) v = w[(int64_t)(n0 + nn) * K + k0 + kk];
            ws[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < LBK; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = xs[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = ws[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + tx + 16 * j;
            if (n >= N) continue;
            float v;
            if (EPI == 0) {
                v = acc[i][j] * wscale + (bias ? bias[n] * bscale : 0.f);
                if (act == SGDFR_ACT_LRELU) v = lrelu_gain(v, slope, gain);
            } else if (EPI == 1) {
                v = rsqrtf(acc[i][j] + wscale);
            } else {
                v = ex.add[(int64_t)m * ldy + n] + ex.mul[(int64_t)m * ldy + n] * (acc[i][j] * wscale);
            }
            y[(int64_t)m * ldy + n] = v;
        }
    }
}

constexpr int LBM = 32, LBN = 32;

template <int XOP, int EPI>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int64_t ldx,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, int64_t ldy, int M, int N, int K,
                                                    float wscale, float bscale, int act, float slope, float gain,
                                                    LinearExtra ex) {
    __shared__ float xs[LBK][LBM + 1];
    __shared__ float ws[LBK][LBN + 1];
    linear_tile<XOP, EPI, LBM, LBN>(x, ldx, w, bias, y, ldy, M, N, K, blockIdx.y * LBM, blockIdx.x * LBN, wscale,
                                    bscale, act, slope, gain, xs, ws, ex);
}

template <int XOP, int EPI>
static int launch_linear(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int M,
                         int N, int K, float wscale, float bscale, int act, float slope, float gain, void* stream,
                         LinearExtra ex = LinearExtra{nullptr, nullptr, nullptr}) {
    dim3 grid((N + LBN - 1) / LBN, (M + LBM - 1) / LBM);
    hipLaunchKernelGGL((linear_kernel<XOP, EPI>), grid, dim3(256), 0, as_stream(stream), x, ldx, w, bias, y, ldy, M, N,
                       K, wscale, bscale, act, slope, gain, ex);
    return check_launch("linear");
}

// Skinny form of EPI 0 / XOP 0 for M = a batch of rows (<= 128) and K <= 512: one WAVE per output column keeps the weight
// row in registers (8 floats per lane) and walks the rows, 4 at a time for independent loads and butterflies.  The tiled
// kernel above launches (N/32) x (M/32) blocks that each walk K serially: 54 us for the mapping network's 64 x 512 x 512
// layers (67 such launches per trainer step); this form takes a few us.
// XOP / EPI as in linear_tile (the demodulation-gradient GEMM of the trainer's backward, 13 launches of 40 us per step in the
// tiled form, takes the same route).
template <int XOP, int EPI>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                           int M, int N, int K, float wscale, float bscale, int act,
                                                           float slope, float gain, LinearExtra ex) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    constexpr int KPL = 8;
    float wr[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int k = lane + 64 * j;
        wr[j] = k < K ? w[(int64_t)n * K + k] : 0.f;
    }
    const float bv = bias ? bias[n] * bscale : 0.f;
    constexpr int RB = 4;
    for (int m0 = 0; m0 < M; m0 += RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m = m0 + r < M ? m0 + r : M - 1;
            const float* xr = x + (int64_t)m * ldx;
            acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int k = lane + 64 * j;
                float xv = k < K ? xr[k] : 0.f;
                if (XOP == 1) xv = xv * xv;
                if (XOP == 2) {
                    const float a = k < K ? ex.xaux[(int64_t)m * ldx + k] : 0.f;
                    xv = -xv * a * a * a;
                }
                acc[r] = fmaf(xv, wr[j], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = wave_sum(acc[r]);
        if (lane < RB && m0 + lane < M) {
            float a = acc[0];
#pragma unroll
            for (int r = 1; r < RB; ++r) a = lane == r ? acc[r] : a;
            float v;
            if (EPI == 0) {
                v = a * wscale + bv;
                if (act == SGDFR_ACT_LRELU) v = lrelu_gain(v, slope, gain);
            } else if (EPI == 1) {
                v = rsqrtf(a + wscale);
            } else {
                v = ex.add[(int64_t)(m0 + lane) * ldy + n] + ex.mul[(int64_t)(m0 + lane) * ldy + n] * (a * wscale);
            }
            y[(int64_t)(m0 + lane) * ldy + n] = v;
        }
    }
}

// All per-layer style modulations (STAGE 0) or all demodulation coefficients (STAGE 1) of one generator
// forward in ONE launch: 20 (resp. 13) small GEMMs are independent, so their tiles simply share a grid.
struct StyleBatch {
    sgdfr_style_layer layer[SGDFR_MAX_STYLE_LAYERS];
    int tile_start[SGDFR_MAX_STYLE_LAYERS + 1];   // prefix sums of tiles per layer
    int n_layers;
    const float* latent;
    int B, L, D;
    float wscale;   // 1/sqrt(D), computed on the host like the single-layer entry point
};

// One WAVE per output column n of one layer: the wave keeps the weight row (K <= 512: 8 floats per lane) in registers,
// then walks the batch: dot(x[b,:], w[n,:]) by 8 FMAs per lane + a 6-step butterfly, lane 0 stores.  K-serial tiles
// (the generic linear kernel) need ~70 us for these skinny GEMMs whatever the batch; this form needs ~10 us.
// e of the range plan (include/sgdfr.h, sgdfr_style_layer): m = max |s| of the row, absmax = bit pattern of max |x| or 0
__device__ __forceinline__ int range_exponent(float m, unsigned absmax_bits, int use_absmax, int x_log2, int headroom) {
    const unsigned mb = __float_as_uint(m);
    if (mb == 0u || mb >= 0x7f800000u) return 0;                    // all-zero or non-finite styles: leave the row alone
    const int fl = max((int)(mb >> 23), 1) - 127;                   // floor(log2 m) (subnormals count as 2^-126)
    int L = x_log2;
    if (use_absmax) {
        if (absmax_bits == 0u || absmax_bits >= 0x7f800000u) return 0;     // zero or non-finite input: nothing to plan
        L = max((int)(absmax_bits >> 23), 1) - 127 + 1;
    }
    return min(max(18 - headroom - L - fl, -120), 120);
}

// A wave owns STYLE_NC consecutive output columns and STYLE_BU images per iteration: one column per wave re-read the whole
// [B, K] input matrix for every column (5000 columns x 64 rows x 2 KB = 650 MB through L2 per forward: 93 + 56 us for ~10 us
// of arithmetic in rocprofv3); with 4 columns the rows are read a quarter as often and 4 images' rows are in flight together.
constexpr int STYLE_NC = 4, STYLE_BU = 2;

template <int STAGE>
__global__ __launch_bounds__(256) void styles_batched_kernel(StyleBatch sb) {
    constexpr int NC = STYLE_NC, BU = STYLE_BU;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;                      // global column GROUP over all layers (tile_start counts groups)
    int li = 0;
    while (li + 1 < sb.n_layers && grp >= sb.tile_start[li + 1]) ++li;
    if (grp >= sb.tile_start[sb.n_layers]) return;
    const sgdfr_style_layer& ly = sb.layer[li];
    const int n0 = (grp - sb.tile_start[li]) * NC;
    const int K = STAGE == 0 ? sb.D : ly.cin;
    const int ncols = STAGE == 0 ? ly.cin : ly.cout;
    const float* wbase = STAGE == 0 ? ly.mod_w : ly.q;
    const float* xb = STAGE == 0 ? sb.latent + (int64_t)ly.latent_index * sb.D : ly.s;
    const int64_t ldx = STAGE == 0 ? (int64_t)sb.L * sb.D : ly.cin;
    float* out = STAGE == 0 ? ly.s : ly.d;
    const int ldo = ncols;
    constexpr int KPL = 8;                                       // K <= 512 -> at most 8 elements per lane
    float w[NC][KPL], bias[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = min(n0 + c, ncols - 1);
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int k = lane + 64 * j;
            w[c][j] = k < K ? wbase[(int64_t)n * K + k] : 0.f;
        }
        bias[c] = STAGE == 0 ? ly.mod_b[n] : 0.f;
    }
    const bool plan = STAGE == 1 && ly.s_n != nullptr;          // range plan: this wave also sees the whole style row
    const unsigned absmax = (plan && ly.x_absmax) ? *ly.x_absmax : 0u;
    // blockIdx.y = image chunk: the serial walk over the images (one exposed load latency per step, a handful of waves per
    // CU) is cut into gridDim.y independent pieces
    const int bchunk = (sb.B + gridDim.y - 1) / gridDim.y;
    const int b_lo = blockIdx.y * bchunk, b_hi = min(sb.B, b_lo + bchunk);
    for (int b0 = b_lo; b0 < b_hi; b0 += BU) {
        float xv[BU][KPL];
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const float* xr = xb + (int64_t)min(b0 + u, b_hi - 1) * ldx;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int k = lane + 64 * j;
                xv[u][j] = k < K ? xr[k] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int b = b0 + u;
            if (b >= b_hi) break;
            float m = 0.f;
            float acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                float v = xv[u][j];
                if (STAGE == 1) {
                    m = fmaxf(m, fabsf(v));
                    v *= v;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fmaf(v, w[c][j], acc[c]);     // per column: the same order as one column per wave
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = wave_sum(acc[c]);
            int e = 0;
            if (plan) {
                m = wave_max(m);
                e = range_exponent(m, absmax, ly.x_absmax != nullptr, ly.x_log2, ly.headroom);
            }
            if (lane < NC && n0 + lane < ncols) {                // lane c stores column c
                float a = acc[0];
#pragma unroll
                for (int c = 1; c < NC; ++c) a = lane == c ? acc[c] : a;
                float bv = bias[0];
#pragma unroll
                for (int c = 1; c < NC; ++c) bv = lane == c ? bias[c] : bv;
                const float r = STAGE == 0 ? a * sb.wscale + bv : rsqrtf(a + 1e-8f);
                out[(int64_t)b * ldo + n0 + lane] = r;
                if (plan) ly.d_n[(int64_t)b * ldo + n0 + lane] = ldexpf(r, -e);
            }
            if (plan && n0 == 0) {                               // the first group's wave also writes the scaled style row
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int k = lane + 64 * j;
                    if (k < K) ly.s_n[(int64_t)b * K + k] = ldexpf(xv[u][j], e);
                }
            }
        }
    }
}

// the range plan for one layer with existing s / d: one block per image
__global__ __launch_bounds__(256) void split_range_kernel(const float* __restrict__ s, const float* __restrict__ d,
                                                         float* __restrict__ s_n, float* __restrict__ d_n,
                                                         const unsigned* __restrict__ x_absmax, int x_absmax_bstride, int x_log2,
                                                         int headroom, int Cin, int Cout) {
    __shared__ float red[4];
    __shared__ unsigned redu[4];
    const int b = blockIdx.x;
    float m = 0.f;
    for (int i = threadIdx.x; i < Cin; i += 256) m = fmaxf(m, fabsf(s[(int64_t)b * Cin + i]));
    unsigned am = 0u;            // x_absmax_bstride words per image (0: one word for the batch): their maximum
    if (x_absmax)
        for (int i = threadIdx.x; i < max(x_absmax_bstride, 1); i += 256) am = max(am, x_absmax[(int64_t)b * x_absmax_bstride + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o, 64));
        am = max(am, (unsigned)__shfl_xor((int)am, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; redu[threadIdx.x >> 6] = am; }
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned absmax = max(max(redu[0], redu[1]), max(redu[2], redu[3]));
    const int e = range_exponent(m, absmax, x_absmax != nullptr, x_log2, headroom);
    for (int i = threadIdx.x; i < Cin; i += 256) s_n[(int64_t)b * Cin + i] = ldexpf(s[(int64_t)b * Cin + i], e);
    for (int i = threadIdx.x; i < Cout; i += 256) d_n[(int64_t)b * Cout + i] = ldexpf(d[(int64_t)b * Cout + i], -e);
}

// max |x| per image (or over the batch) as an atomicMax on bit patterns.  One atomic per BLOCK (the first version issued one per
// wave from 1024 blocks per image: 4096 same-address atomics per image serialised in L2 -- 156 us per call in the autograd
// forward of the direction trainer, 2.2 ms of a 14.6 ms step at B=16); four 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t x_bstride, int64_t n, unsigned* __restrict__ out,
                                                    int per_image) {
    __shared__ unsigned wmax[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * x_bstride;
    unsigned m = 0u;
    auto take4 = [&](const float4& v) {
        m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
    };
    if ((n & 3) == 0 && (x_bstride & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
        const float4* x4 = reinterpret_cast<const float4*>(xb);
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
            take4(v0); take4(v1); take4(v2); take4(v3);
        }
        for (; i < n4; i += stride) take4(x4[i]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(xb[i])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (m != 0u) atomicMax(out + (per_image ? b : 0), m);
    }
}

}  // namespace sgdfr

using namespace sgdfr;

extern "C" int sgdfr_linear_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                                int M, int N, int K, float wscale, float bscale, int act, float slope, float gain,
                                void* stream) {
    SGDFR_REQUIRE(M >= 0 && N > 0 && K > 0, "linear: bad shape M=%d N=%d K=%d", M, N, K);
    if (M == 0) return 0;
    SGDFR_REQUIRE(x && w && y, "linear: null pointer");
    SGDFR_REQUIRE(ldx >= K && ldy >= N, "linear: leading dims too small");
    SGDFR_REQUIRE(act == SGDFR_ACT_NONE || act == SGDFR_ACT_LRELU, "linear: unknown act %d", act);
    if (M <= 128 && K <= 512 && N >= 64) {      // a batch of rows through a small layer: wave-per-column form
        hipLaunchKernelGGL((linear_skinny_kernel<0, 0>), dim3((N + 3) / 4), dim3(256), 0, as_stream(stream), x, ldx, w, bias, y, ldy, M, N,
                           K, wscale, bscale, act, slope, gain, LinearExtra{nullptr, nullptr, nullptr});
        return check_launch("linear(skinny)");
    }
    return launch_linear<0, 0>(x, ldx, w, bias, y, ldy, M, N, K, wscale, bscale, act, slope, gain, stream);
}

extern "C" int sgdfr_style_demod_f32(const float* style, int64_t ld_style, const float* mod_w, const float* mod_b,
                                     const float* q, float* s, float* d, int B, int D, int Cin, int Cout,
                                     void* stream) {
    SGDFR_REQUIRE(B >= 0 && D > 0 && Cin > 0, "style_demod: bad shape B=%d D=%d Cin=%d", B, D, Cin);
    if (B == 0) return 0;
    SGDFR_REQUIRE(style && mod_w && mod_b && s, "style_demod: null pointer");
    SGDFR_REQUIRE(ld_style >= D, "style_demod: ld_style < D");
    int rc;
    static const bool skinny_ok = !(getenv("SGDFR_STYLE_SKINNY") && atoi(getenv("SGDFR_STYLE_SKINNY")) == 0);
    if (skinny_ok && B <= 128 && D <= 512 && Cin >= 64) {      // a few rows through a small layer: the wave-per-column form (see sgdfr_linear_f32)
        hipLaunchKernelGGL((linear_skinny_kernel<0, 0>), dim3((Cin + 3) / 4), dim3(256), 0, as_stream(stream), style, ld_style, mod_w, mod_b, s,
                           (int64_t)Cin, B, Cin, D, 1.0f / sqrtf((float)D), 1.0f, SGDFR_ACT_NONE, 0.f, 1.f,
                           LinearExtra{nullptr, nullptr, nullptr});
        rc = check_launch("style_demod(skinny)");
    } else {
        rc = launch_linear<0, 0>(style, ld_style, mod_w, mod_b, s, Cin, B, Cin, D, 1.0f / sqrtf((float)D), 1.0f,
                                 SGDFR_ACT_NONE, 0.f, 1.f, stream);
    }
    if (rc || !d) return rc;
    SGDFR_REQUIRE(q && Cout > 0, "style_demod: d requested without q/Cout");
    return launch_linear<1, 1>(s, Cin, q, nullptr, d, Cout, B, Cout, Cin, 1e-8f, 0.f, 0, 0.f, 1.f, stream);
}

extern "C" int sgdfr_styles_batched_f32(const float* latent, int B, int L, int D, const sgdfr_style_layer* layers,
                                        int n_layers, void* stream) {
    SGDFR_REQUIRE(B >= 0 && L > 0 && D > 0, "styles_batched: bad shape B=%d L=%d D=%d", B, L, D);
    SGDFR_REQUIRE(n_layers > 0 && n_layers <= SGDFR_MAX_STYLE_LAYERS, "styles_batched: n_layers %d not in 1..%d",
                  n_layers, SGDFR_MAX_STYLE_LAYERS);
    if (B == 0) return 0;
    SGDFR_REQUIRE(latent && layers, "styles_batched: null pointer");
    StyleBatch sb{};
    sb.n_layers = n_layers; sb.latent = latent; sb.B = B; sb.L = L; sb.D = D;
    sb.wscale = 1.0f / sqrtf((float)D);
    SGDFR_REQUIRE(D <= 512, "styles_batched: style_dim %d > 512", D);
    int tiles = 0, dl = 0;     // tile_start[] holds prefix sums of output column GROUPS (STYLE_NC columns, one wave each)
    for (int i = 0; i < n_layers; ++i) {
        const sgdfr_style_layer& ly = layers[i];
        SGDFR_REQUIRE(ly.mod_w && ly.mod_b && ly.s && ly.cin > 0, "styles_batched: layer %d incomplete", i);
        SGDFR_REQUIRE(ly.latent_index >= 0 && ly.latent_index < L, "styles_batched: layer %d latent index %d out of range",
                      i, ly.latent_index);
        SGDFR_REQUIRE(!ly.d || (ly.q && ly.cout > 0), "styles_batched: layer %d wants d without q", i);
        SGDFR_REQUIRE((ly.s_n == nullptr) == (ly.d_n == nullptr) && (!ly.s_n || ly.d), "styles_batched: layer %d: the range plan needs s_n, d_n and d", i);
        SGDFR_REQUIRE(!ly.s_n || (ly.headroom >= 0 && ly.headroom <= 12 && abs(ly.x_log2) <= 100), "styles_batched: layer %d: bad range plan (x_log2 %d, headroom %d)", i, ly.x_log2, ly.headroom);
        sb.layer[i] = ly;
        SGDFR_REQUIRE(ly.cin <= 512 || !ly.d, "styles_batched: layer %d has %d input channels (> 512)", i, ly.cin);
        sb.tile_start[i] = tiles;
        tiles += (ly.cin + STYLE_NC - 1) / STYLE_NC;      // column groups
    }
    sb.tile_start[n_layers] = tiles;
    const int chunks = B >= 64 ? 8 : B >= 16 ? 4 : B >= 4 ? 2 : 1;       // image chunks (grid.y)
    hipLaunchKernelGGL(styles_batched_kernel<0>, dim3((tiles + 3) / 4, chunks), dim3(256), 0, as_stream(stream), sb);
    if (int rc = check_launch("styles_batched(modulation)")) return rc;
    // demodulation coefficients: only layers that asked for d
    StyleBatch sd{};
    sd.latent = latent; sd.B = B; sd.L = L; sd.D = D;
    tiles = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (!layers[i].d) continue;
        sd.layer[dl] = layers[i];
        sd.tile_start[dl] = tiles;
        tiles += (layers[i].cout + STYLE_NC - 1) / STYLE_NC;
        ++dl;
    }
    if (dl == 0) return 0;
    sd.tile_start[dl] = tiles;
    sd.n_layers = dl;
    hipLaunchKernelGGL(styles_batched_kernel<1>, dim3((tiles + 3) / 4, chunks), dim3(256), 0, as_stream(stream), sd);
    return check_launch("styles_batched(demod)");
}

extern "C" int sgdfr_split_range_f32(const float* s, const float* d, float* s_n, float* d_n, const unsigned* x_absmax,
                                     int x_absmax_bstride, int x_log2, int headroom, int B, int Cin, int Cout, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0, "split_range: bad shape %d %d %d", B, Cin, Cout);
    if (B == 0) return 0;
    SGDFR_REQUIRE(s && d && s_n && d_n, "split_range: null pointer");
    SGDFR_REQUIRE(headroom >= 0 && headroom <= 12 && abs(x_log2) <= 100, "split_range: bad plan (x_log2 %d, headroom %d)", x_log2, headroom);
    SGDFR_REQUIRE(x_absmax_bstride >= 0, "split_range: x_absmax_bstride is 0 (one word), 1 (one per image) or the words per image");
    hipLaunchKernelGGL(split_range_kernel, dim3(B), dim3(256), 0, as_stream(stream), s, d, s_n, d_n, x_absmax, x_absmax_bstride, x_log2,
                       headroom, Cin, Cout);
    return check_launch("split_range");
}

extern "C" int sgdfr_absmax_f32(const float* x, int64_t x_bstride, int64_t n_per_image, int B, unsigned* out, int per_image,
                                void* stream) {
    SGDFR_REQUIRE(B >= 0 && n_per_image >= 0 && x_bstride >= 0, "absmax: bad shape B=%d n=%lld", B, (long long)n_per_image);
    SGDFR_REQUIRE(out, "absmax: null output");
    if (hipMemsetAsync(out, 0, sizeof(unsigned) * (per_image ? (B > 0 ? B : 1) : 1), as_stream(stream)) != hipSuccess) {
        set_error("absmax: hipMemsetAsync failed");
        return 2;
    }
    if (B == 0 || n_per_image == 0) return 0;
    SGDFR_REQUIRE(x, "absmax: null input");
    const int nb = x_bstride == 0 ? 1 : B;                       // a broadcast image is read once
    // 16 K floats per block (16 loads of 16 bytes per thread), at most ~2048 blocks over the batch: enough loads in flight for HBM
    // speed, <= 128 atomics per image
    int gx = (int)((n_per_image + 16383) / 16384);
    const int cap = nb >= 16 ? 128 : 2048 / nb;
    gx = gx < 1 ? 1 : (gx > cap ? cap : gx);
    hipLaunchKernelGGL(absmax_kernel, dim3(gx, nb), dim3(256), 0, as_stream(stream), x, x_bstride, n_per_image, out, x_bstride == 0 ? 0 : per_image);
    return check_launch("absmax");
}

// ---- backward of sgdfr_styles_batched_f32 for the frozen generator (autograd.StylesBatchedFn): every layer's
//   ds_l[b,i] = gs_l[b,i] + s_l[b,i] * sum_o (-(A_l/d_l) * d_l^3)[b,o] * qt_l[i,o]           (demodulated 3x3 convs; sgdfr_demod_grad_f32)
//   ds_l[b,i] = (sum_j rgb_r[b,j,i] * rgb_w[j,i]) / sqrt(cin)                                (ToRGB: no demodulation)
// in ONE launch (stage 0), and the latent gradient glat[b,l,:] = sum_{layers of latent row l} ds_l[b,:] @ mod_w_l / sqrt(D) in a
// second (stage 1) -- instead of (A/d, demod_grad, a transposed copy of mod_w, a linear, an indexed add) x 20 layers.
struct StyleGradBatch {
    sgdfr_style_grad_layer layer[SGDFR_MAX_STYLE_LAYERS];
    int tile_start[SGDFR_MAX_STYLE_LAYERS + 1];   // stage 0: prefix sums of 4-column groups per layer
    int n_layers;
    float* glat;
    int B, L, D;
    float wscale;
};

__global__ __launch_bounds__(256) void styles_batched_bwd_ds_kernel(StyleGradBatch sb) {
    constexpr int NC = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    int li = 0;
    while (li + 1 < sb.n_layers && grp >= sb.tile_start[li + 1]) ++li;
    if (grp >= sb.tile_start[sb.n_layers]) return;
    const sgdfr_style_grad_layer& ly = sb.layer[li];
    const int n0 = (grp - sb.tile_start[li]) * NC;
    const int cin = ly.cin, K = ly.cout;
    if (ly.rgb_r) {          // ToRGB: lane = column within the group's 4 (x 16 images per pass)
        const float sc = 1.0f / sqrtf((float)cin);
        for (int e = lane; e < NC * sb.B; e += 64) {
            const int b = e / NC, i = n0 + e % NC;
            if (i >= cin) continue;
            const float* r = ly.rgb_r + (int64_t)b * 3 * cin;
            ly.ds[(int64_t)b * cin + i] = (r[i] * ly.rgb_w[i] + r[cin + i] * ly.rgb_w[cin + i] + r[2 * cin + i] * ly.rgb_w[2 * cin + i]) * sc;
        }
        return;
    }
    if (!ly.a) {             // a 3x3 conv without demodulation: ds = gs
        for (int e = lane; e < NC * sb.B; e += 64) {
            const int b = e / NC, i = n0 + e % NC;
            if (i < cin) ly.ds[(int64_t)b * cin + i] = ly.gs[(int64_t)b * cin + i];
        }
        return;
    }
    constexpr int KPL = 8;   // cout <= 512
    float w[NC][KPL];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = min(n0 + c, cin - 1);
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int k = lane + 64 * j;
            w[c][j] = k < K ? ly.qt[(int64_t)n * K + k] : 0.f;
        }
    }
    constexpr int RB = 2;          // images per iteration: their loads are in flight together
    for (int bb = 0; bb < sb.B; bb += RB) {
        float dv[RB][KPL], av[RB][KPL];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int b = min(bb + r, sb.B - 1);
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const int k = lane + 64 * j;
                dv[r][j] = k < K ? ly.d[(int64_t)b * K + k] : 1.f;
                av[r][j] = k < K ? ly.a[((int64_t)b * K + k) * ly.a_stride] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int b = bb + r;
            if (b >= sb.B) break;
            float acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const float gd = av[r][j] / dv[r][j];
                const float v = -gd * dv[r][j] * dv[r][j] * dv[r][j];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fmaf(v, w[c][j], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = wave_sum(acc[c]);
            if (lane < NC && n0 + lane < cin) {
                float a = acc[0];
#pragma unroll
                for (int c = 1; c < NC; ++c) a = lane == c ? acc[c] : a;
                const int64_t o = (int64_t)b * cin + n0 + lane;
                ly.ds[o] = ly.gs[o] + ly.s[o] * a;
            }
        }
    }
}

// glat[b, l, n] for one latent row l = blockIdx.y, 64 columns n per BLOCK, SG_BU images per block (blockIdx.z): lanes run along
// the D columns of mod_w [cin, D] (coalesced rows); the block's ds rows sit in LDS (broadcast reads); the four waves split the
// cin range and their partial sums meet in LDS, added in wave order.  Layers of one latent row are summed in layer order: no
// atomics, rows nobody reads come out as zeros.  (First version: one wave walked all of cin with ds from global memory -- 56
// blocks of a 512-step dependent loop, 468 us per launch at B=16; this form takes a few us.)
constexpr int SG_BU = 8;
constexpr int SG_MAXC = 512;
__global__ __launch_bounds__(256) void styles_batched_bwd_lat_kernel(StyleGradBatch sb) {
    __shared__ float dsl[SG_BU][SG_MAXC];
    __shared__ float part[4][SG_BU][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l = blockIdx.y;
    const int n = blockIdx.x * 64 + lane;
    const int b0 = blockIdx.z * SG_BU;
    float acc[SG_BU];
#pragma unroll
    for (int u = 0; u < SG_BU; ++u) acc[u] = 0.f;
    for (int li = 0; li < sb.n_layers; ++li) {
        const sgdfr_style_grad_layer& ly = sb.layer[li];
        if (ly.latent_index != l) continue;
        const int cin = ly.cin;
        for (int c0 = 0; c0 < cin; c0 += SG_MAXC) {
            const int cn = min(SG_MAXC, cin - c0);
            __syncthreads();
            for (int e = threadIdx.x; e < SG_BU * cn; e += 256) {
                const int u = e / cn, i = e - u * cn;
                dsl[u][i] = ly.ds[(int64_t)min(b0 + u, sb.B - 1) * cin + c0 + i];
            }
            __syncthreads();
            const int per = (cn + 3) / 4, i_lo = wave * per, i_hi = min(cn, i_lo + per);
            float p[SG_BU];
#pragma unroll
            for (int u = 0; u < SG_BU; ++u) p[u] = 0.f;
            int i = i_lo;
            for (; i + 4 <= i_hi; i += 4) {
                float wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[q] = n < sb.D ? ly.mod_w[(int64_t)(c0 + i + q) * sb.D + n] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < SG_BU; ++u) p[u] = fmaf(dsl[u][i + q], wv[q], p[u]);
            }
            for (; i < i_hi; ++i) {
                const float wv = n < sb.D ? ly.mod_w[(int64_t)(c0 + i) * sb.D + n] : 0.f;
#pragma unroll
                for (int u = 0; u < SG_BU; ++u) p[u] = fmaf(dsl[u][i], wv, p[u]);
            }
#pragma unroll
            for (int u = 0; u < SG_BU; ++u) acc[u] += p[u] * sb.wscale;
        }
    }
#pragma unroll
    for (int u = 0; u < SG_BU; ++u) part[wave][u][lane] = acc[u];
    __syncthreads();
    if (wave == 0 && n < sb.D) {
#pragma unroll
        for (int u = 0; u < SG_BU; ++u)
            if (b0 + u < sb.B)
                sb.glat[((int64_t)(b0 + u) * sb.L + l) * sb.D + n] = ((part[0][u][lane] + part[1][u][lane]) + part[2][u][lane]) + part[3][u][lane];
    }
}

// Stage 2 (only when the modulation weights are trained: PTI, libs/optimization.py:31-40): per layer
//   gmod_w[i, n] = sum_b ds[b,i] * latent[b, latent_index, n] / sqrt(D)      gmod_b[i] = sum_b ds[b,i]
// lanes along n (coalesced latent rows and gmod_w rows), one wave per 4 rows i.
__global__ __launch_bounds__(256) void styles_batched_bwd_w_kernel(StyleGradBatch sb, const float* __restrict__ latent) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = blockIdx.x * 4 + wave;
    int li = 0;
    while (li + 1 < sb.n_layers && grp >= sb.tile_start[li + 1]) ++li;
    if (grp >= sb.tile_start[sb.n_layers]) return;
    const sgdfr_style_grad_layer& ly = sb.layer[li];
    if (!ly.gmod_w && !ly.gmod_b) return;
    const int i0 = (grp - sb.tile_start[li]) * 4;
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        if (i >= ly.cin) break;
        if (ly.gmod_b && lane == 0) {
            float t = 0.f;
            for (int b = 0; b < sb.B; ++b) t += ly.ds[(int64_t)b * ly.cin + i];
            ly.gmod_b[i] = t;
        }
        if (ly.gmod_w) {
            for (int n = lane; n < sb.D; n += 64) {
                float t = 0.f;
                for (int b = 0; b < sb.B; ++b)
                    t = fmaf(ly.ds[(int64_t)b * ly.cin + i], latent[((int64_t)b * sb.L + ly.latent_index) * sb.D + n], t);
                ly.gmod_w[(int64_t)i * sb.D + n] = t * sb.wscale;
            }
        }
    }
}

// dq[o,i] = sum_b (-0.5 * (a/d) * d^3)[b,o] * s[b,i]^2 : dL/dQ of the demodulation d = rsqrt(sum_i s^2 Q + eps), one launch instead of
// (A/d, d^3, two products, two transposed copies, s*s, a linear)
__global__ __launch_bounds__(256) void demod_dq_kernel(const float* __restrict__ a, int64_t a_stride, const float* __restrict__ d,
                                                      const float* __restrict__ s, float* __restrict__ dq, int B, int Cin, int Cout) {
    const int64_t total = (int64_t)Cout * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % Cin), o = (int)(idx / Cin);
        float t = 0.f;
        for (int b = 0; b < B; ++b) {
            const float dv = d[(int64_t)b * Cout + o];
            const float coeff = (a[((int64_t)b * Cout + o) * a_stride] / dv) * (dv * dv * dv) * -0.5f;
            const float sv = s[(int64_t)b * Cin + i];
            t = fmaf(coeff, sv * sv, t);
        }
        dq[idx] = t;
    }
}

// Small parameter gradients of one backward, every layer in one launch (sgdfr_param_grads_f32): blockIdx -> (entry, block of it)
struct ParamGradBatch {
    sgdfr_param_grad e[SGDFR_MAX_PARAM_GRADS];
    int block_start[SGDFR_MAX_PARAM_GRADS + 1];
    int n, B;
};

__global__ __launch_bounds__(256) void param_grads_kernel(ParamGradBatch pb) {
    __shared__ float red[4];
    int ei = 0;
    while (ei + 1 < pb.n && (int)blockIdx.x >= pb.block_start[ei + 1]) ++ei;
    const sgdfr_param_grad& e = pb.e[ei];
    const int blk = blockIdx.x - pb.block_start[ei], nblk = pb.block_start[ei + 1] - pb.block_start[ei];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = pb.B, C = e.C;
    if (e.kind == SGDFR_PGRAD_BIAS) {               // out[c] = sum_b in[(b*C+c)*3]
        const int c = blk * 256 + tid;
        if (c >= C) return;
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += e.in[((int64_t)b * C + c) * 3];
        e.out[c] = t;
    } else if (e.kind == SGDFR_PGRAD_RGB_W) {       // out[j*C+i] = scale * sum_b in[(b*3+j)*C+i] * aux[b*C+i]
        const int idx = blk * 256 + tid;
        if (idx >= 3 * C) return;
        const int j = idx / C, i = idx - j * C;
        float t = 0.f;
        for (int b = 0; b < B; ++b) t = fmaf(e.in[((int64_t)b * 3 + j) * C + i], e.aux[(int64_t)b * C + i], t);
        e.out[idx] = t * e.scale;
    } else if (e.kind == SGDFR_PGRAD_NOISE) {       // out[0] = sum_{b,c} in[(b*C+c)*3+1]      (one block)
        float t = 0.f;
        for (int64_t k = tid; k < (int64_t)B * C; k += 256) t += e.in[k * 3 + 1];
        t = wave_sum(t);
        if (lane == 0) red[wave] = t;
        __syncthreads();
        if (tid == 0) e.out[0] = (red[0] + red[1]) + (red[2] + red[3]);
    } else {                                        // RGB_B: out[j] += sum_{b,p} in[(b*3+j)*HW+p]   (nblk/3 blocks per j, atomics into zeroed out)
        const int per_j = nblk / 3, j = blk / per_j, part = blk - j * per_j;
        float t = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* src = e.in + ((int64_t)b * 3 + j) * e.HW;
            for (int p = part * 256 + tid; p < e.HW; p += per_j * 256) t += src[p];
        }
        t = wave_sum(t);
        if (lane == 0) red[wave] = t;
        __syncthreads();
        if (tid == 0) atomicAdd(&e.out[j], (red[0] + red[1]) + (red[2] + red[3]));
    }
}

extern "C" int sgdfr_param_grads_f32(const sgdfr_param_grad* entries, int n, int B, void* stream) {
    SGDFR_REQUIRE(n > 0 && n <= SGDFR_MAX_PARAM_GRADS && B > 0, "param_grads: bad n=%d B=%d", n, B);
    SGDFR_REQUIRE(entries, "param_grads: null pointer");
    ParamGradBatch pb;
    pb.n = n; pb.B = B;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const sgdfr_param_grad& e = entries[i];
        SGDFR_REQUIRE(e.in && e.out && e.C > 0 && e.kind >= 0 && e.kind <= 3, "param_grads: entry %d: bad kind / C / null pointer", i);
        SGDFR_REQUIRE(e.kind != SGDFR_PGRAD_RGB_W || e.aux, "param_grads: entry %d: the ToRGB weight gradient needs aux = s", i);
        SGDFR_REQUIRE(e.kind != SGDFR_PGRAD_RGB_B || e.HW > 0, "param_grads: entry %d: the ToRGB bias gradient needs HW", i);
        pb.e[i] = e;
        pb.block_start[i] = blocks;
        if (e.kind == SGDFR_PGRAD_BIAS) blocks += (e.C + 255) / 256;
        else if (e.kind == SGDFR_PGRAD_RGB_W) blocks += (3 * e.C + 255) / 256;
        else if (e.kind == SGDFR_PGRAD_NOISE) blocks += 1;
        else { int per = (int)(((int64_t)B * e.HW + 16383) / 16384); per = per < 1 ? 1 : (per > 64 ? 64 : per); blocks += 3 * per; }
    }
    pb.block_start[n] = blocks;
    hipLaunchKernelGGL(param_grads_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), pb);
    return check_launch("param_grads");
}

extern "C" int sgdfr_demod_dq_f32(const float* a, int64_t a_stride, const float* d, const float* s, float* dq, int B, int Cin,
                                  int Cout, void* stream) {
    SGDFR_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && a_stride >= 1, "demod_dq: bad shape %d %d %d", B, Cin, Cout);
    SGDFR_REQUIRE(a && d && s && dq, "demod_dq: null pointer");
    const int64_t total = (int64_t)Cout * Cin;
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(demod_dq_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), a, a_stride, d, s, dq, B, Cin, Cout);
    return check_launch("demod_dq");
}

extern "C" int sgdfr_styles_batched_bwd_f32(const sgdfr_style_grad_layer* layers, int n_layers, const float* latent, float* glat,
                                            int B, int L, int D, void* stream) {
    SGDFR_REQUIRE(B >= 0 && L > 0 && D > 0 && n_layers > 0 && n_layers <= SGDFR_MAX_STYLE_LAYERS, "styles_batched_bwd: bad shape B=%d L=%d D=%d layers=%d",
                  B, L, D, n_layers);
    if (B == 0) return 0;
    SGDFR_REQUIRE(layers && (glat || latent), "styles_batched_bwd: null pointer");
    StyleGradBatch sb;
    sb.n_layers = n_layers; sb.glat = glat; sb.B = B; sb.L = L; sb.D = D; sb.wscale = 1.0f / sqrtf((float)D);
    int groups = 0;
    for (int i = 0; i < n_layers; ++i) {
        const sgdfr_style_grad_layer& ly = layers[i];
        SGDFR_REQUIRE(ly.cin > 0 && ly.latent_index >= 0 && ly.latent_index < L && ly.mod_w && ly.ds, "styles_batched_bwd: layer %d: bad cin / latent row / null pointer", i);
        SGDFR_REQUIRE(ly.rgb_r ? (ly.rgb_w != nullptr) : (ly.gs != nullptr), "styles_batched_bwd: layer %d: needs gs, or rgb_r with rgb_w", i);
        SGDFR_REQUIRE(!ly.a || (ly.d && ly.s && ly.qt && ly.cout > 0 && ly.cout <= 512 && ly.a_stride >= 1), "styles_batched_bwd: layer %d: a needs d, s, qt and cout <= 512", i);
        sb.layer[i] = ly;
        sb.tile_start[i] = groups;
        groups += (ly.cin + 3) / 4;
    }
    sb.tile_start[n_layers] = groups;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(styles_batched_bwd_ds_kernel, dim3((groups + 3) / 4), dim3(256), 0, st, sb);
    if (int rc = check_launch("styles_batched_bwd(ds)")) return rc;
    bool want_w = false;
    for (int i = 0; i < n_layers; ++i) want_w = want_w || layers[i].gmod_w || layers[i].gmod_b;
    SGDFR_REQUIRE(!want_w || latent, "styles_batched_bwd: modulation weight gradients need the latent");
    if (want_w) {
        hipLaunchKernelGGL(styles_batched_bwd_w_kernel, dim3((groups + 3) / 4), dim3(256), 0, st, sb, latent);
        if (int rc = check_launch("styles_batched_bwd(weights)")) return rc;
    }
    if (!glat) return 0;
    hipLaunchKernelGGL(styles_batched_bwd_lat_kernel, dim3((D + 63) / 64, L, (B + SG_BU - 1) / SG_BU), dim3(256), 0, st, sb);
    return check_launch("styles_batched_bwd(latent)");
}

extern "C" int sgdfr_demod_grad_f32(const float* gd, const float* d, const float* qt, const float* s, const float* gs,
                                    float* ds, int B, int Cin, int Cout, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0, "demod_grad: bad shape %d %d %d", B, Cin, Cout);
    if (B == 0) return 0;
    SGDFR_REQUIRE(gd && d && qt && s && gs && ds, "demod_grad: null pointer");
    // ds[b,i] = gs[b,i] + s[b,i] * sum_o (-gd[b,o] d[b,o]^3) * Q[o,i]      (qt = Q^T, [Cin, Cout])
    if (B <= 128 && Cout <= 512 && Cin >= 64) {      // a batch of rows through a small layer: wave-per-column form
        hipLaunchKernelGGL((linear_skinny_kernel<2, 2>), dim3((Cin + 3) / 4), dim3(256), 0, as_stream(stream), gd, (int64_t)Cout, qt,
                           (const float*)nullptr, ds, (int64_t)Cin, B, Cin, Cout, 1.0f, 0.f, 0, 0.f, 1.f, LinearExtra{d, gs, s});
        return check_launch("demod_grad(skinny)");
    }
    return launch_linear<2, 2>(gd, Cout, qt, nullptr, ds, Cin, B, Cin, Cout, 1.0f, 0.f, 0, 0.f, 1.f, stream,
                               LinearExtra{d, gs, s});
}
