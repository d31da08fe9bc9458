// Small dense layers of the path: mapping-network EqualLinear (+ fused leaky-ReLU), the per-layer
// style modulation, the demodulation coefficients and DirectionMatrix.  M = batch (1..512),
// N, K <= 4096: a few hundred MFLOP per forward against ~2 TFLOP of convolution, so these are
// LDS-tiled fp32 VALU kernels (exact fmaf chains) sized for launch latency, not MFMA.
#include "common.h"

namespace sgdfr {

constexpr int LBM = 32, LBN = 64, LBK = 32;

// EPI 0: y = act(acc*wscale + bias*bscale);  EPI 1: y = rsqrt(acc + eps)  (wscale carries eps)
// SQX: use x^2 instead of x (demodulation: sum_i s^2 q)
template <bool SQX, int EPI>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int64_t ldx,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, int64_t ldy, int M, int N, int K,
                                                    float wscale, float bscale, int act, float slope, float gain) {
    __shared__ float xs[LBK][LBM + 1];
    __shared__ float ws[LBK][LBN + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15;   // N direction: 4 columns each (tx + 16*j)
    const int ty = tid >> 4;   // M direction: 2 rows each (ty + 16*i)
    const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
    float acc[2][4] = {};
    for (int k0 = 0; k0 < K; k0 += LBK) {
        // x tile: LBM x LBK, lanes along k (contiguous in memory)
        for (int e = tid; e < LBM * LBK; e += 256) {
            const int kk = e & (LBK - 1), mm = e >> 5;
            float v = 0.f;
            if (m0 + mm < M && k0 + kk < K) v = x[(int64_t)(m0 + mm) * ldx + k0 + kk];
            xs[kk][mm] = SQX ? v * v : v;
        }
        for (int e = tid; e < LBN * LBK; e += 256) {
            const int kk = e & (LBK - 1), nn = e >> 5;
            float v = 0.f;
            if (n0 + nn < N && k0 + kk < K) v = w[(int64_t)(n0 + nn) * K + k0 + kk];
            ws[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < LBK; ++kk) {
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = xs[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = ws[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx + 16 * j;
            if (n >= N) continue;
            float v;
            if (EPI == 0) {
                v = acc[i][j] * wscale + (bias ? bias[n] * bscale : 0.f);
                if (act == SGDFR_ACT_LRELU) v = lrelu_gain(v, slope, gain);
            } else {
                v = rsqrtf(acc[i][j] + wscale);
            }
            y[(int64_t)m * ldy + n] = v;
        }
    }
}

template <bool SQX, int EPI>
static int launch_linear(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int M,
                         int N, int K, float wscale, float bscale, int act, float slope, float gain, void* stream) {
    dim3 grid((N + LBN - 1) / LBN, (M + LBM - 1) / LBM);
    hipLaunchKernelGGL((linear_kernel<SQX, EPI>), grid, dim3(256), 0, as_stream(stream), x, ldx, w, bias, y, ldy, M, N,
                       K, wscale, bscale, act, slope, gain);
    return check_launch("linear");
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
    return launch_linear<false, 0>(x, ldx, w, bias, y, ldy, M, N, K, wscale, bscale, act, slope, gain, stream);
}

extern "C" int sgdfr_style_demod_f32(const float* style, int64_t ld_style, const float* mod_w, const float* mod_b,
                                     const float* q, float* s, float* d, int B, int D, int Cin, int Cout,
                                     void* stream) {
    SGDFR_REQUIRE(B >= 0 && D > 0 && Cin > 0, "style_demod: bad shape B=%d D=%d Cin=%d", B, D, Cin);
    if (B == 0) return 0;
    SGDFR_REQUIRE(style && mod_w && mod_b && s, "style_demod: null pointer");
    SGDFR_REQUIRE(ld_style >= D, "style_demod: ld_style < D");
    int rc = launch_linear<false, 0>(style, ld_style, mod_w, mod_b, s, Cin, B, Cin, D, 1.0f / sqrtf((float)D), 1.0f,
                                     SGDFR_ACT_NONE, 0.f, 1.f, stream);
    if (rc || !d) return rc;
    SGDFR_REQUIRE(q && Cout > 0, "style_demod: d requested without q/Cout");
    return launch_linear<true, 1>(s, Cin, q, nullptr, d, Cout, B, Cout, Cin, 1e-8f, 0.f, 0, 0.f, 1.f, stream);
}
