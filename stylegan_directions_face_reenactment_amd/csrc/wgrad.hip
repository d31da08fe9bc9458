// Weight gradient of the shared-weight modulated conv (needed by PTI, libs/optimization.py:32-68):
//     dWc[o,i,ky,kx] = sum_{b,p} (d*g)[b,o, shifted p] * (x*s)[b,i,p]
// as a split-K fp32-MFMA GEMM: rows = output channels o, columns = input channels i, nine accumulators
// (one per tap), reduction over pixels.  The pixel stream is cut into chunks of PC pixels inside one image
// row; per chunk the block stages the style-scaled input row segment [64 i][PC] and the demod-scaled gradient
// neighbourhood [64 o][segments][PC+2]:
//     PLAIN3: 3 segments = gradient rows y+1, y, y-1  (tap (ky,kx) reads row 2-ky at column j+2-kx)
//     UP3   : 8 segments = 4 parity planes x 2 plane rows (tap (ky,kx) reads plane (ky&1,kx&1), row y+(ky>>1),
//             column j+(kx>>1)) -- the planes produced by sgdfr_blur_adjoint_f32
// Channel strides in LDS are odd so the MFMA fragment reads (32 lanes = 32 channels) are conflict-free.
// Blocks own a (64 o x 64 i) tile and a contiguous range of chunks; partial sums are added to the packed
// gradient dwp[Cin][9][Cout] with fp32 atomics.
#include "common.h"

namespace sgdfr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const float* g;
    const float* d;
    const float* x;
    int64_t x_bstride;
    const float* s;
    float* dwp;
    float* part;       // != NULL: slice ks stores its sums to part[ks][9][Cout][Cin] (no atomics), reduced by the finish kernel
    int B, Cin, Cout, H, W, P, R;
    int PC, chunks_per_row, total_chunks, chunks_per_block, n_ot, n_it;
    int lr4;           // log2(PC / 4): a staged row is 2^lr4 float4 pieces (PC is a power of two >= 4 on the MFMA path)
};

template <int MODE, int PCMAX>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradParams p) {
    constexpr int NSEG = (MODE == SGDFR_MODE_UP3) ? 8 : 3;
    constexpr int SEG = PCMAX + 2;
    constexpr int GST = (NSEG * SEG) | 1;   // odd channel stride
    constexpr int UST = PCMAX | 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lu = smem;                 // [64][UST]
    float* lg = smem + 64 * UST;      // [64][GST]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave >> 1, wi = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntile = p.n_ot * p.n_it;
    const int tile = blockIdx.x % ntile, ks = blockIdx.x / ntile;
    const int ot = tile / p.n_it, it = tile - ot * p.n_it;
    const int HW = p.H * p.W, RP = p.R * p.P;
    const int PC = p.PC;

    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    const int c_lo = ks * p.chunks_per_block;
    const int c_hi = min(p.total_chunks, c_lo + p.chunks_per_block);
    for (int c = c_lo; c < c_hi; ++c) {
        const int img = c / (p.H * p.chunks_per_row);
        const int rem = c - img * p.H * p.chunks_per_row;
        const int y = rem / p.chunks_per_row;
        const int x0 = (rem - y * p.chunks_per_row) * PC;
        __syncthreads();   // previous chunk's fragments are consumed
        // Staging by float4 pieces, no divisions: item e of a pass = (channel e >> lr4, piece e & (R4-1)).  (The first version
        // walked single elements with three runtime divisions and two scale loads each: ~40 us per chunk against 9 us of
        // MFMA -- 3.0 ms per 64-channel layer at B=16.)
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const int R4 = 1 << p.lr4;
        // ---- style-scaled inputs  u[i][j] = x[img,i,y,x0+j] * s[img,i]
        for (int e = tid; e < (64 << p.lr4); e += 256) {
            const int il = e >> p.lr4, q = e & (R4 - 1);
            const int i = it * 64 + il;
            f32x4u v = {0.f, 0.f, 0.f, 0.f};
            if (i < p.Cin) {
                v = *reinterpret_cast<const f32x4u*>(p.x + (int64_t)img * p.x_bstride + (int64_t)i * HW + y * p.W + x0 + 4 * q);
                v *= p.s[img * p.Cin + i];
            }
            float* dst = lu + il * UST + 4 * q;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        // ---- demod-scaled gradient neighbourhood
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
            for (int e = tid; e < (64 << p.lr4); e += 256) {
                const int ol = e >> p.lr4, q = e & (R4 - 1);
                const int o = ot * 64 + ol;
                f32x4u v = {0.f, 0.f, 0.f, 0.f};
                float edge = 0.f;            // PLAIN3: column x0-1 (piece 0) / x0+PC (last piece); UP3: column x0+PC (last piece)
                float* dst = lg + ol * GST + sg * SEG;
                if (o < p.Cout) {
                    const float dv = p.d ? p.d[img * p.Cout + o] : 1.f;
                    if (MODE == SGDFR_MODE_UP3) {
                        const float* src = p.g + (((int64_t)img * p.Cout + o) * 4 + (sg >> 1)) * RP + (y + (sg & 1)) * p.P + x0;
                        v = *reinterpret_cast<const f32x4u*>(src + 4 * q) * dv;
                        if (q == R4 - 1) edge = src[PC] * dv;
                    } else {
                        const int yy = y + sg - 1;
                        if (yy >= 0 && yy < p.H) {
                            const float* src = p.g + ((int64_t)img * p.Cout + o) * HW + yy * p.W + x0;
                            v = *reinterpret_cast<const f32x4u*>(src + 4 * q) * dv;
                            if (q == 0 && x0 > 0) edge = src[-1] * dv;
                            if (q == R4 - 1 && x0 + PC < p.W) edge = src[PC] * dv;      // (R4 == 1: the right edge wins below)
                        }
                    }
                }
                if (MODE == SGDFR_MODE_UP3) {
                    dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
                    if (q == R4 - 1) dst[PC] = edge;
                } else {
                    dst[1 + 4 * q] = v[0]; dst[2 + 4 * q] = v[1]; dst[3 + 4 * q] = v[2]; dst[4 + 4 * q] = v[3];
                    if (R4 == 1) {            // one piece per row: this thread owns both edges
                        float le = 0.f, re = 0.f;
                        const int yy = y + sg - 1;
                        if (o < p.Cout && yy >= 0 && yy < p.H) {
                            const float dv = p.d ? p.d[img * p.Cout + o] : 1.f;
                            const float* src = p.g + ((int64_t)img * p.Cout + o) * HW + yy * p.W + x0;
                            if (x0 > 0) le = src[-1] * dv;
                            if (x0 + PC < p.W) re = src[PC] * dv;
                        }
                        dst[0] = le; dst[PC + 1] = re;
                    } else {
                        if (q == 0) dst[0] = edge;
                        if (q == R4 - 1) dst[PC + 1] = edge;
                    }
                }
            }
        }
        __syncthreads();
        const float* lub = lu + (wi * 32 + l31) * UST + hi;
        const float* lgb = lg + (wo * 32 + l31) * GST + hi;
        for (int j = 0; j < PC; j += 2) {
            const float b = lub[j];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ky = k / 3, kx = k - ky * 3;
                int off;
                if (MODE == SGDFR_MODE_UP3)
                    off = ((2 * (ky & 1) + (kx & 1)) * 2 + (ky >> 1)) * SEG + (kx >> 1);
                else
                    off = (2 - ky) * SEG + 2 - kx;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(lgb[off + j], b, acc[k], 0, 0, 0);
            }
        }
    }
    // ---- split-K combine.  Atomics (dwp[i][tap][o] += acc) cost ~0.35 ms per layer whatever its size: every layer ends up with
    // ~9.4 M of them (256-way contention per address on the 64x64 layers).  With a partials buffer each slice stores its
    // sums as [tap][o][i] rows (the 32 lanes of a half-wave = 32 consecutive i = one 128-byte line) and the finish kernel adds
    // the slices in fixed order (deterministic).
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ot * 64 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int i = it * 64 + wi * 32 + l31;
            if (o < p.Cout && i < p.Cin) {
                if (p.part) p.part[(((int64_t)ks * 9 + k) * p.Cout + o) * p.Cin + i] = acc[k][r];
                else atomicAdd(&p.dwp[((int64_t)i * 9 + k) * p.Cout + o], acc[k][r]);
            }
        }
}

// Any shape (odd widths, channel counts that are not multiples of 4 ...): one thread per (i, tap, o).
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgradParams p, int mode) {
    const int64_t total = (int64_t)p.Cin * 9 * p.Cout;
    const int HW = p.H * p.W, RP = p.R * p.P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % p.Cout);
        const int k = (int)((idx / p.Cout) % 9);
        const int i = (int)(idx / ((int64_t)p.Cout * 9));
        const int ky = k / 3, kx = k - ky * 3;
        float acc = 0.f;
        for (int b = 0; b < p.B; ++b) {
            const float sc = p.s[b * p.Cin + i] * (p.d ? p.d[b * p.Cout + o] : 1.f);
            const float* xp = p.x + (int64_t)b * p.x_bstride + (int64_t)i * HW;
            float part = 0.f;
            for (int y = 0; y < p.H; ++y)
                for (int x = 0; x < p.W; ++x) {
                    float gv;
                    if (mode == SGDFR_MODE_UP3) {
                        gv = p.g[(((int64_t)b * p.Cout + o) * 4 + 2 * (ky & 1) + (kx & 1)) * RP + (y + (ky >> 1)) * p.P + x +
                                 (kx >> 1)];
                    } else {
                        const int yy = y + 1 - ky, xx = x + 1 - kx;
                        if (yy < 0 || yy >= p.H || xx < 0 || xx >= p.W) continue;
                        gv = p.g[((int64_t)b * p.Cout + o) * HW + yy * p.W + xx];
                    }
                    part = fmaf(xp[y * p.W + x], gv, part);
                }
            acc = fmaf(part, sc, acc);
        }
        p.dwp[idx] = acc;
    }
}

// dW[o][i][k] = scale * ( dwp[i][k][o] + 2 * wp[i][k][o] * dq[o][i] )      (dq = dL/dQ from the demodulation, may be NULL)
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ dwp, const float* __restrict__ wp,
                                                          const float* __restrict__ dq, float* __restrict__ dw, int Cout,
                                                          int Cin, float scale) {
    const int64_t total = (int64_t)Cout * Cin * 9;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % Cout);     // o fastest: coalesced reads of the packed tensors
        const int k = (int)((idx / Cout) % 9);
        const int i = (int)(idx / ((int64_t)Cout * 9));
        float v = dwp[idx];
        if (dq) v = fmaf(2.f * wp[idx], dq[(int64_t)o * Cin + i], v);
        dw[((int64_t)o * Cin + i) * 9 + k] = v * scale;
    }
}

// dW[o][i][k] = scale * ( sum_ks part[ks][k][o][i] + 2 * wp[i][k][o] * dq[o][i] )      (i fastest: coalesced partial reads)
__global__ __launch_bounds__(256) void wgrad_finish_parts_kernel(const float* __restrict__ part, int ksplit,
                                                                const float* __restrict__ wp, const float* __restrict__ dq,
                                                                float* __restrict__ dw, int Cout, int Cin, float scale) {
    const int64_t total = (int64_t)Cout * Cin * 9;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % Cin);
        const int o = (int)((idx / Cin) % Cout);
        const int k = (int)(idx / ((int64_t)Cin * Cout));
        float v = 0.f;
        for (int ks = 0; ks < ksplit; ++ks) v += part[(int64_t)ks * total + idx];
        if (dq) v = fmaf(2.f * wp[((int64_t)i * 9 + k) * Cout + o], dq[(int64_t)o * Cin + i], v);
        dw[((int64_t)o * Cin + i) * 9 + k] = v * scale;
    }
}

// The same finish with every stream coalesced.  wgrad_finish_parts_kernel walks the OUTPUT order of the slices (i fastest), so its
// reads of wp[i][k][o] are 4-byte gathers at a stride of 9*Cout floats and its writes of dW[o][i][k] 4-byte scatters at a stride of 9
// (512 x 512: 151 MB of cache lines for 9.4 MB of weights; 28 us per launch, a tenth of a PTI step for 13 layers).  Here a block owns
// one cout and 64 input channels: thread (ii, kg) sums the slices ks = kg (mod 4) of its channel for all nine taps (64-float rows of
// `part`), the four partial sums meet in LDS in fixed order, and the 576 outputs leave as ONE contiguous run of dW[o][i0..i0+63][0..8]
// -- the demodulation term read from the ORIGINAL weight tensor w[o][i][k] (wp = w * scale, the same product prepack_kernel formed) at
// the same addresses.
constexpr int WF_IT = 64;
// dq may also be formed here: with a (= d * dL/dd, element stride a_stride), d [B,Cout] and s [B,Cin] given instead of dq,
// dq[o,i] = sum_b (-0.5 * (a/d) * d^3)[b,o] * s[b,i]^2 (sgdfr_demod_dq_f32's expression and order) -- one launch less per layer.
__global__ __launch_bounds__(256) void wgrad_finish_parts_oik_kernel(const float* __restrict__ part, int ksplit,
                                                                    const float* __restrict__ w, const float* __restrict__ dq,
                                                                    const float* __restrict__ a, int64_t a_stride,
                                                                    const float* __restrict__ d, const float* __restrict__ sty, int B,
                                                                    float* __restrict__ dw, int Cout, int Cin, float scale) {
    __shared__ float red[4][9][WF_IT];
    __shared__ float dql[WF_IT];
    const int o = blockIdx.y, i0 = blockIdx.x * WF_IT;
    const int ii = threadIdx.x & (WF_IT - 1), kg = threadIdx.x >> 6;
    const int64_t total = (int64_t)Cout * Cin * 9;
    const bool in = i0 + ii < Cin;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (in) {
        for (int ks = kg; ks < ksplit; ks += 4) {
            const float* src = part + (int64_t)ks * total + (int64_t)o * Cin + i0 + ii;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += src[(int64_t)k * Cout * Cin];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[kg][k][ii] = acc[k];
    if (a && kg == 0) {
        float t = 0.f;
        if (in) {
            for (int b = 0; b < B; ++b) {
                const float dv = d[(int64_t)b * Cout + o];
                const float coeff = (a[((int64_t)b * Cout + o) * a_stride] / dv) * (dv * dv * dv) * -0.5f;
                const float sv = sty[(int64_t)b * Cin + i0 + ii];
                t = fmaf(coeff, sv * sv, t);
            }
        }
        dql[ii] = t;
    }
    __syncthreads();
    const int n_i = min(WF_IT, Cin - i0);
    const int64_t obase = ((int64_t)o * Cin + i0) * 9;
    for (int j = threadIdx.x; j < n_i * 9; j += 256) {
        const int i = j / 9, k = j - i * 9;
        float v = ((red[0][k][i] + red[1][k][i]) + red[2][k][i]) + red[3][k][i];
        if (dq) v = fmaf(2.f * (w[obase + j] * scale), dq[(int64_t)o * Cin + i0 + i], v);
        else if (a) v = fmaf(2.f * (w[obase + j] * scale), dql[i], v);
        dw[obase + j] = v * scale;
    }
}

// K slices of a shape: ~1024 blocks, at least 4 pixel chunks per block
static int wgrad_ksplit(const WgradParams& p, int* chunks_per_block) {
    const int ntile = ((p.Cout + 63) / 64) * ((p.Cin + 63) / 64);
    int ksplit = (1024 + ntile - 1) / ntile;
    int maxsplit = (p.total_chunks + 3) / 4;
    if (maxsplit < 1) maxsplit = 1;
    if (ksplit > maxsplit) ksplit = maxsplit;
    const int cpb = (p.total_chunks + ksplit - 1) / ksplit;
    if (chunks_per_block) *chunks_per_block = cpb;
    return (p.total_chunks + cpb - 1) / cpb;
}

template <int MODE, int PCMAX>
static int launch_wgrad(WgradParams& p, hipStream_t st) {
    constexpr int NSEG = (MODE == SGDFR_MODE_UP3) ? 8 : 3;
    constexpr int GST = (NSEG * (PCMAX + 2)) | 1, UST = PCMAX | 1;
    p.n_ot = (p.Cout + 63) / 64;
    p.n_it = (p.Cin + 63) / 64;
    const int ntile = p.n_ot * p.n_it;
    const int ksplit = wgrad_ksplit(p, &p.chunks_per_block);
    const size_t lds = (size_t)(64 * UST + 64 * GST) * sizeof(float);
    auto kern = wgrad_mfma_kernel<MODE, PCMAX>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return check_launch("wgrad(lds attribute)");
    }
    hipLaunchKernelGGL(kern, dim3(ntile * ksplit), dim3(256), lds, st, p);
    return check_launch("modconv_wgrad");
}

}  // namespace sgdfr

using namespace sgdfr;

// pixel chunking of a shape; false: only the direct kernel applies
static bool wgrad_shape(WgradParams& p, int B, int Cin, int Cout, int H, int W, int mode) {
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    const int pcmax = (mode == SGDFR_MODE_UP3) ? 32 : 64;
    const int PC = W < pcmax ? W : pcmax;
    if (!((W % PC == 0) && PC >= 4 && (PC & (PC - 1)) == 0)) return false;      // float4 pieces, shift arithmetic
    p.PC = PC;
    p.lr4 = 0;
    while ((4 << p.lr4) < PC) ++p.lr4;
    p.chunks_per_row = W / PC;
    p.total_chunks = B * H * p.chunks_per_row;
    return true;
}

extern "C" int sgdfr_modconv_wgrad_ksplit(int B, int Cin, int Cout, int H, int W, int mode) {
    WgradParams p{};
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || !wgrad_shape(p, B, Cin, Cout, H, W, mode)) return 0;
    return wgrad_ksplit(p, nullptr);
}

extern "C" int sgdfr_modconv_wgrad_parts_f32(const float* g, const float* d, const float* x, int64_t x_bstride, const float* s,
                                             float* part, int B, int Cin, int Cout, int H, int W, int mode, void* stream) {
    SGDFR_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "wgrad_parts: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin,
                  Cout, H, W);
    SGDFR_REQUIRE(mode == SGDFR_MODE_PLAIN3 || mode == SGDFR_MODE_UP3, "wgrad_parts: mode must be PLAIN3 or UP3, got %d", mode);
    SGDFR_REQUIRE(g && x && s && part, "wgrad_parts: null pointer");
    WgradParams p{};
    p.g = g; p.d = d; p.x = x; p.x_bstride = x_bstride; p.s = s; p.part = part;
    SGDFR_REQUIRE(wgrad_shape(p, B, Cin, Cout, H, W, mode), "wgrad_parts: shape needs the direct kernel (sgdfr_modconv_wgrad_ksplit "
                  "returned 0): use sgdfr_modconv_wgrad_f32");
    hipStream_t st = as_stream(stream);
    if (mode == SGDFR_MODE_UP3) return launch_wgrad<SGDFR_MODE_UP3, 32>(p, st);
    return launch_wgrad<SGDFR_MODE_PLAIN3, 64>(p, st);
}

extern "C" int sgdfr_modconv_wgrad_finish_parts_f32(const float* part, int ksplit, const float* wp, const float* dq,
                                                    float* dweight, int Cout, int Cin, void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && ksplit > 0, "wgrad_finish_parts: bad shape %d %d x%d", Cout, Cin, ksplit);
    SGDFR_REQUIRE(part && dweight && (wp || !dq), "wgrad_finish_parts: null pointer");
    const int64_t total = (int64_t)Cout * Cin * 9;
    int64_t gsz = (total + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(wgrad_finish_parts_kernel, dim3((int)gsz), dim3(256), 0, as_stream(stream), part, ksplit, wp, dq, dweight,
                       Cout, Cin, 1.0f / sqrtf((float)Cin * 9));
    return check_launch("modconv_wgrad_finish_parts");
}

extern "C" int sgdfr_modconv_wgrad_finish_parts_oik_f32(const float* part, int ksplit, const float* weight, const float* dq,
                                                        const float* a, int64_t a_stride, const float* d, const float* s, int B,
                                                        float* dweight, int Cout, int Cin, void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && ksplit > 0, "wgrad_finish_parts_oik: bad shape %d %d x%d", Cout, Cin, ksplit);
    SGDFR_REQUIRE(part && dweight && (weight || (!dq && !a)), "wgrad_finish_parts_oik: null pointer");
    SGDFR_REQUIRE(!(dq && a), "wgrad_finish_parts_oik: give dq OR (a, d, s), not both");
    SGDFR_REQUIRE(!a || (d && s && B > 0 && a_stride >= 1), "wgrad_finish_parts_oik: a needs d, s, B and a_stride >= 1");
    hipLaunchKernelGGL(wgrad_finish_parts_oik_kernel, dim3((Cin + WF_IT - 1) / WF_IT, Cout), dim3(256), 0, as_stream(stream), part, ksplit,
                       weight, dq, a, a_stride, d, s, B, dweight, Cout, Cin, 1.0f / sqrtf((float)Cin * 9));
    return check_launch("modconv_wgrad_finish_parts_oik");
}

extern "C" int sgdfr_modconv_wgrad_f32(const float* g, const float* d, const float* x, int64_t x_bstride, const float* s,
                                       float* dwp, int B, int Cin, int Cout, int H, int W, int mode, void* stream) {
    SGDFR_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "wgrad: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin,
                  Cout, H, W);
    SGDFR_REQUIRE(mode == SGDFR_MODE_PLAIN3 || mode == SGDFR_MODE_UP3, "wgrad: mode must be PLAIN3 or UP3, got %d", mode);
    SGDFR_REQUIRE(g && x && s && dwp, "wgrad: null pointer");
    hipStream_t st = as_stream(stream);
    WgradParams p{};
    p.g = g; p.d = d; p.x = x; p.x_bstride = x_bstride; p.s = s; p.dwp = dwp;
    const bool mfma_ok = wgrad_shape(p, B, Cin, Cout, H, W, mode);
    if (!mfma_ok) {
        const int64_t total = (int64_t)Cin * 9 * Cout;
        int64_t gsz = (total + 255) / 256;
        if (gsz > 256 * 16) gsz = 256 * 16;
        hipLaunchKernelGGL(wgrad_direct_kernel, dim3((int)gsz), dim3(256), 0, st, p, mode);
        return check_launch("modconv_wgrad(direct)");
    }
    if (hipMemsetAsync(dwp, 0, sizeof(float) * (size_t)Cin * 9 * Cout, st) != hipSuccess) return check_launch("memset");
    if (mode == SGDFR_MODE_UP3) return launch_wgrad<SGDFR_MODE_UP3, 32>(p, st);
    return launch_wgrad<SGDFR_MODE_PLAIN3, 64>(p, st);
}

extern "C" int sgdfr_modconv_wgrad_finish_f32(const float* dwp, const float* wp, const float* dq, float* dweight, int Cout,
                                              int Cin, void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0, "wgrad_finish: bad shape %d %d", Cout, Cin);
    SGDFR_REQUIRE(dwp && dweight && (wp || !dq), "wgrad_finish: null pointer");
    const int64_t total = (int64_t)Cout * Cin * 9;
    int64_t gsz = (total + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((int)gsz), dim3(256), 0, as_stream(stream), dwp, wp, dq, dweight, Cout,
                       Cin, 1.0f / sqrtf((float)Cin * 9));
    return check_launch("modconv_wgrad_finish");
}
