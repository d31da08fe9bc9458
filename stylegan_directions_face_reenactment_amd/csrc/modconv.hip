// Shared-weight modulated convolution for StyleGAN2 on gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Algebra (reference formulation libs/gan/StyleGAN2/model.py:232-273):
//     y[b,o] = d[b,o] * conv( x[b,i] * s[b,i], Wc[o,i,ky,kx] )        Wc = W / sqrt(Cin*9)
// The reference materialises per-sample weights [B,Cout,Cin,3,3] and runs a grouped conv; here
// the batch folds into the GEMM pixel dimension and ONE weight tensor is shared by every image:
//     D[cout, pixel] += A[cout, k] * B[k, pixel],   k = (cin, ky, kx)
// A (weights) and B (style-scaled inputs) are staged through LDS; the MFMA 32x32x2 k-pair is two
// adjacent input channels at the same tap.  C/D layout puts pixels on lanes, so output rows are
// written as 128-byte contiguous segments of NCHW.
//
// Input staging is im2col-free: all images live in one zero-padded "flat" space
//     q = (img*(H+1) + row+1) * (W+1) + col+1
// where one zero column / zero row is SHARED between neighbouring rows / images.  Every 3x3 (or,
// for the stride-2 transposed conv, 2x2) neighbourhood of an output pixel is then a constant
// offset in q, and the LDS tile of a block is simply a contiguous q-range [q0, q0+xlen) -- no
// halo logic, any image size, several small images per tile.
//
// MODE_UP3 computes conv_transpose2d(stride 2) as its four output-parity phases in one pass:
// super-pixel (a,b) of the (H+1)x(W+1) grid owns T[2a+py, 2b+px]; weight tap (ky,kx) feeds phase
// (ky&1, kx&1) from input (a - (ky==2), b - (kx==2)), i.e. 9 MFMA groups per channel pair -- the
// same 9 MACs per input pixel the transposed conv costs, with no zero-stuffed work.
#include "common.h"

namespace sgdfr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ModconvParams {
    const float* x;
    int64_t x_bstride;
    const float* wp;
    const float* s;
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    float* y;
    int B, Cin, Cout, H, W;
    int P, R;             // padded pitch W+1, padded rows per image H+1
    int n_pix_tiles, n_cout_tiles;
    int xs;               // LDS floats per staged channel (multiple of 4, >= xlen)
    int xlen;             // q-range length staged per channel
    int seglen, segpitch; // PLAIN3 on wide images: the range is 3 row segments of seglen, segpitch (= P) apart; else (xlen, 0)
    int64_t total_pix;    // PLAIN: B*H*W dense pixels ; UP: B*R*P super-pixels
    int act;
    float slope, gain;
    int ksplit;           // K (input-channel) slices; > 1: every slice writes d*partial sums to y + slice*split_stride
    int64_t split_stride; // and sgdfr's reduce kernel adds them up and applies noise / bias / activation
};

constexpr int CK = 4;  // input channels per LDS stage (2 MFMA k-pairs)


// Global -> registers for one K stage (CK input channels): inputs + styles of the block's q-range
// and the [CK*9, NT] weight slab.  Issued BEFORE the MFMAs of the previous stage so HBM/L2 latency
// hides under them; written to LDS after the next barrier (register-staged pipeline).
// Branch-free per lane: zero-fill positions load a safe address and are masked at store time.
template <int NT, int EX, int WV, int NPL>
__device__ __forceinline__ void load_stage(const ModconvParams& p, int c0, int tid, int n0, int cstride, int pstride,
                                           int nex, const int64_t (&xoff)[EX], const int (&soff)[EX],
                                           float (&xr)[EX][CK * NPL], float4 (&sr)[EX], float4 (&wr)[WV]) {
    constexpr int WF4 = CK * 9 * NT / 4;
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        if (e < nex) {  // block-uniform
            const float* xp = p.x + xoff[e] + (int64_t)c0 * cstride;
            sr[e] = *reinterpret_cast<const float4*>(p.s + soff[e] + c0);
#pragma unroll
            for (int c = 0; c < CK; ++c)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) xr[e][c * NPL + pl] = xp[(int64_t)c * cstride + pl * pstride];
        }
    }
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int f = tid + v * 256;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < WF4) {
            const int row = f / (NT / 4), col = (f - row * (NT / 4)) * 4;
            if (n0 + col < p.Cout)
                w4 = *reinterpret_cast<const float4*>(p.wp + ((int64_t)c0 * 9 + row) * p.Cout + n0 + col);
        }
        wr[v] = w4;
    }
}

template <int NT, int EX, int WV, int NPL>
__device__ __forceinline__ void store_stage(const ModconvParams& p, int tid, int nex, unsigned okmask, float* lx,
                                            float* lw, const float (&xr)[EX][CK * NPL], const float4 (&sr)[EX],
                                            const float4 (&wr)[WV]) {
    constexpr int WF4 = CK * 9 * NT / 4;
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        if (e < nex) {
            const int j = tid + e * 256;
            const bool ok = (okmask >> e) & 1u;
            if (j < p.xlen) {
                const float sc[4] = {sr[e].x, sr[e].y, sr[e].z, sr[e].w};
#pragma unroll
                for (int c = 0; c < CK; ++c)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        lx[(c * NPL + pl) * p.xs + j] = ok ? xr[e][c * NPL + pl] * sc[c] : 0.f;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int f = tid + v * 256;
        if (f < WF4) reinterpret_cast<float4*>(lw)[f] = wr[v];
    }
}

template <int MODE, int WM, int WN, int MI, int NI, int EX, int OCC>
__global__ __launch_bounds__(256, OCC) void modconv_mfma_kernel(ModconvParams p) {
    constexpr int NT = WM * MI * 32;
    constexpr int PT = WN * NI * 32;
    constexpr int PH = (MODE == SGDFR_MODE_UP3) ? 4 : 1;          // output parity phases (accumulator sets)
    constexpr int NPL = (MODE == SGDFR_MODE_DOWN3) ? 4 : 1;       // input parity planes
    constexpr int WROWS = CK * 9;
    constexpr int WF4 = WROWS * NT / 4;            // float4 per weight stage
    constexpr int WV = (WF4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per block");

    // two LDS stages: [x: CK*xs][w: CK*9*NT] each
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stage_floats = CK * NPL * p.xs + CK * 9 * NT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W;
    const int RP = p.R * p.P;

    // XCD-aware (bijective) remap: the blocks resident on one XCD walk neighbouring pixel tiles of
    // one cout tile, so its weight slice and the overlapping input rows stay in that XCD's L2.
    int lid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int tiles_per_slice = p.n_pix_tiles * p.n_cout_tiles;
    const int ks = lid / tiles_per_slice;                 // K slice of this block (0 when ksplit == 1)
    const int lt = lid - ks * tiles_per_slice;
    const int ct = lt / p.n_pix_tiles, pt = lt - ct * p.n_pix_tiles;
    const int n0 = ct * NT;
    const int p0 = pt * PT;      // host guarantees every pixel / q index fits int32

    // ---- tile origin in the padded flat space
    int q0;
    if (MODE == SGDFR_MODE_PLAIN3) {
        const int img = p0 / HW;
        const int rem = p0 - img * HW;
        const int a = rem / p.W, b = rem - a * p.W;
        q0 = (img * p.R + a + 1) * p.P + b + 1 - p.P - 1;
    } else if (MODE == SGDFR_MODE_DOWN3) {
        const int img = p0 / HW;
        const int rem = p0 - img * HW;
        const int a = rem / p.W, b = rem - a * p.W;
        q0 = img * RP + a * p.P + b;
    } else {
        q0 = p0;
    }

    // ---- per-lane B-fragment offsets (pixel -> position inside the staged q-range)
    int boff[NI];
    int pixv[NI];
    const int total_pix = (int)p.total_pix;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        int pix = p0 + (wn * NI + ni) * 32 + l31;
        pixv[ni] = pix;
        if (pix >= total_pix) pix = total_pix - 1;
        int qb;
        if (MODE == SGDFR_MODE_PLAIN3) {
            const int img = pix / HW;
            const int rem = pix - img * HW;
            const int a = rem / p.W, b = rem - a * p.W;
            qb = (img * p.R + a + 1) * p.P + b + 1;
        } else if (MODE == SGDFR_MODE_DOWN3) {
            const int img = pix / HW;
            const int rem = pix - img * HW;
            const int a = rem / p.W, b = rem - a * p.W;
            qb = img * RP + a * p.P + b;
        } else {
            qb = pix;
        }
        boff[ni] = (p.segpitch ? (pix - p0) + p.seglen + 1 : (qb - q0)) + hi * NPL * p.xs;
    }

    // ---- staging descriptors (fixed for the whole K loop): element offsets into x / s; positions that
    // are padding (or beyond the batch) point at offset 0 and are zeroed through okmask
    int64_t xoff[EX];
    int soff[EX];
    unsigned okmask = 0;
    const int nex = (p.xlen + 255) >> 8;
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        const int j = tid + e * 256;
        const int sg = j / p.seglen;
        const int q = q0 + sg * p.segpitch + (j - sg * p.seglen);
        bool ok;
        int img;
        int64_t off;
        if (MODE == SGDFR_MODE_DOWN3) {      // input = parity planes [B, Cin, 4, R, P]: every plane entry is real data
            img = q / RP;
            ok = (j < p.xlen) && img < p.B;
            off = (int64_t)img * p.x_bstride + (q - img * RP);
        } else {
            const int pir = q / p.P;
            const int pc = q - pir * p.P;
            img = pir / p.R;
            const int pr = pir - img * p.R;
            ok = (j < p.xlen) && pc >= 1 && pr >= 1 && img < p.B;
            off = (int64_t)img * p.x_bstride + (pr - 1) * p.W + (pc - 1);
        }
        xoff[e] = ok ? off : 0;
        soff[e] = ok ? img * p.Cin : 0;
        okmask |= ok ? (1u << e) : 0u;
    }

    f32x16 acc[PH][MI][NI];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][mi][ni][r] = 0.f;

    float xr[EX][CK * NPL];
    const int cstride = (MODE == SGDFR_MODE_DOWN3) ? 4 * RP : HW;   // input channel stride
    float4 sr[EX];
    float4 wr[WV];

    // tap offsets inside the q-range (uniform)
    int tapoff[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ky = k / 3, kx = k - ky * 3;
        if (MODE == SGDFR_MODE_PLAIN3)
            tapoff[k] = (ky - 1) * (p.segpitch ? p.seglen : p.P) + (kx - 1);
        else if (MODE == SGDFR_MODE_UP3)
            tapoff[k] = (ky == 2 ? 0 : p.P) + (kx == 2 ? 0 : 1);
        else   // DOWN3: T[2a+ky, 2b+kx] lives in plane (ky&1, kx&1) at [a + (ky>>1), b + (kx>>1)]
            tapoff[k] = (2 * (ky & 1) + (kx & 1)) * p.xs + (ky >> 1) * p.P + (kx >> 1);
    }
    const int aoff = wm * MI * 32 + l31;

    // Software pipeline, ONE barrier per K stage: while the MFMAs consume LDS stage `cur`, the registers that
    // were filled during the previous iteration are written to stage `cur^1` and the global loads of the stage
    // after that are issued (their latency hides under this iteration's MFMAs).
    const int nstage_all = p.Cin / CK;
    const int st0 = (int)((int64_t)nstage_all * ks / p.ksplit), st1 = (int)((int64_t)nstage_all * (ks + 1) / p.ksplit);
    const int nstage = st1 - st0;                          // stages of this K slice
    load_stage<NT, EX, WV, NPL>(p, st0 * CK, tid, n0, cstride, RP, nex, xoff, soff, xr, sr, wr);
    store_stage<NT, EX, WV, NPL>(p, tid, nex, okmask, smem, smem + CK * NPL * p.xs, xr, sr, wr);
    if (nstage > 1) load_stage<NT, EX, WV, NPL>(p, (st0 + 1) * CK, tid, n0, cstride, RP, nex, xoff, soff, xr, sr, wr);
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        float* lx = smem + (st & 1) * stage_floats;
        float* lw = lx + CK * NPL * p.xs;
        if (st + 1 < nstage) {
            float* nx = smem + ((st + 1) & 1) * stage_floats;
            store_stage<NT, EX, WV, NPL>(p, tid, nex, okmask, nx, nx + CK * NPL * p.xs, xr, sr, wr);
            if (st + 2 < nstage)
                load_stage<NT, EX, WV, NPL>(p, (st0 + st + 2) * CK, tid, n0, cstride, RP, nex, xoff, soff, xr, sr, wr);
        }
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            const float* lxc = lx + (cp * 2) * NPL * p.xs;
            const float* lwc = lw + ((cp * 2 + hi) * 9) * NT + aoff;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int ph = (MODE == SGDFR_MODE_UP3) ? (2 * ((k / 3) & 1) + ((k % 3) & 1)) : 0;
                float a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = lwc[k * NT + mi * 32];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = lxc[boff[ni] + tapoff[k]];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[ph][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[ph][mi][ni], 0, 0, 0);
            }
        }
        __syncthreads();  // stage st fully consumed, stage st+1 fully written
    }

    // ---- epilogue.  C/D layout of 32x32: column (pixel) = lane&31, row (cout) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* const yout = p.y + (int64_t)ks * p.split_stride;
    const bool whole = p.ksplit == 1;      // K slices only scale by d; noise / bias / activation follow the reduction
    // Loads and stores share the in-order vmcnt counter: a `d` / bias load between two stores would make the wave wait
    // for the previous store's HBM round trip.  The coefficients go through LDS (lgkmcnt), noise into registers first.
    const int unit = (MODE == SGDFR_MODE_UP3) ? RP : HW;           // "pixels" per image of this mode
    const int img0 = p0 / unit;
    int nimg;
    {
        int last = p0 + PT - 1;
        if (last >= total_pix) last = total_pix - 1;
        nimg = last / unit - img0 + 1;
    }
    float* const dl = smem;                 // [nimg][NT]; the K loop ended on a barrier, the stage buffers are free
    float* const bl = smem + nimg * NT;     // [NT]
    for (int e = tid; e < nimg * NT; e += 256) {
        const int m = e / NT, c = e - m * NT;
        dl[e] = (p.d && n0 + c < p.Cout) ? p.d[(int64_t)(img0 + m) * p.Cout + n0 + c] : 1.f;
    }
    for (int e = tid; e < NT; e += 256) bl[e] = (whole && p.bias && n0 + e < p.Cout) ? p.bias[n0 + e] : 0.f;
    float nzv[NI];
    {
        const float nw = (MODE != SGDFR_MODE_UP3 && whole && p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int pix = pixv[ni] < total_pix ? pixv[ni] : total_pix - 1;
            const int64_t img = pix / unit;
            nzv[ni] = (MODE != SGDFR_MODE_UP3 && whole && p.noise) ? nw * p.noise[img * p.noise_bstride + (pix - (int)img * unit)] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pix = pixv[ni];
        if (pix >= total_pix) continue;
        const int64_t img = pix / unit;
        const int rem = pix - (int)img * unit;
        const float* dln = dl + ((int)img - img0) * NT;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int co = n0 + cl;
                if (co < p.Cout) {
                    const float dv = dln[cl];
                    if (MODE != SGDFR_MODE_UP3) {
                        float v = acc[0][mi][ni][r] * dv + nzv[ni] + bl[cl];
                        if (whole && p.act) v = lrelu_gain(v, p.slope, p.gain);
                        yout[(img * p.Cout + co) * HW + rem] = v;
                    } else {
                        float* dst = yout + ((img * p.Cout + co) * 4) * RP + rem;
#pragma unroll
                        for (int ph = 0; ph < PH; ++ph) dst[(int64_t)ph * RP] = acc[ph][mi][ni][r] * dv;
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------- split-K reduction
// y[i] = act( sum_s part[s][i] + noise_w*noise + bias[c] ), fixed summation order (deterministic, no atomics)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t stride,
                                                           const float* __restrict__ noise, int64_t noise_bstride,
                                                           const float* __restrict__ noise_w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int64_t n, int C, int inner, int act, float slope, float gain) {
    const float nw = (noise && noise_w) ? noise_w[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += part[s * stride + i];
        const int64_t pl = i / inner;
        const int rem = (int)(i - pl * inner);
        if (noise) v = fmaf(nw, noise[(pl / C) * noise_bstride + rem], v);
        if (bias) v += bias[pl % C];
        if (act) v = lrelu_gain(v, slope, gain);
        y[i] = v;
    }
}

int launch_splitk_reduce(const float* partials, int splits, int64_t n, const float* noise, int64_t noise_bstride,
                         const float* noise_w, const float* bias, float* y, int C, int inner, int act, float slope, float gain,
                         hipStream_t st) {
    int64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, st, partials, splits, n, noise, noise_bstride, noise_w,
                       bias, y, n, C, inner, act, slope, gain);
    return check_launch("splitk_reduce");
}

// ---------------------------------------------------------------- odd-shape path
// Channel counts that are not multiples of 4 never occur in the generator (512..16 channels) but the
// ModulatedConv2d API accepts them: one thread per output element, same packed weights, same output
// layouts as the MFMA kernel.
__global__ __launch_bounds__(256) void modconv_direct_kernel(ModconvParams p, int mode) {
    const int HW = p.H * p.W, RP = p.R * p.P;
    const int64_t per_img = (mode == SGDFR_MODE_UP3) ? (int64_t)p.Cout * 4 * RP : (int64_t)p.Cout * HW;
    const int64_t total = per_img * p.B;
    const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = idx / per_img;
        const int64_t r = idx - img * per_img;
        const float* xb = p.x + img * p.x_bstride;
        const float* sb = p.s + img * p.Cin;
        float acc = 0.f;
        if (mode == SGDFR_MODE_DOWN3) {
            const int co = (int)(r / HW), rem = (int)(r - (int64_t)co * HW);
            const int a = rem / p.W, b = rem - a * p.W;
            for (int i = 0; i < p.Cin; ++i) {
                float part = 0.f;
                const float* pl = xb + (int64_t)i * 4 * RP;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx)
                        part = fmaf(pl[(2 * (ky & 1) + (kx & 1)) * RP + (a + (ky >> 1)) * p.P + b + (kx >> 1)],
                                    p.wp[((int64_t)i * 9 + ky * 3 + kx) * p.Cout + co], part);
                acc = fmaf(part, sb[i], acc);
            }
            float v = acc * (p.d ? p.d[img * p.Cout + co] : 1.f);
            if (p.bias) v += p.bias[co];
            if (p.act) v = lrelu_gain(v, p.slope, p.gain);
            p.y[idx] = v;
        } else if (mode == SGDFR_MODE_PLAIN3) {
            const int co = (int)(r / HW), rem = (int)(r - (int64_t)co * HW);
            const int a = rem / p.W, b = rem - a * p.W;
            for (int i = 0; i < p.Cin; ++i) {
                float part = 0.f;
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = a + ky - 1;
                    if (yy < 0 || yy >= p.H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = b + kx - 1;
                        if (xx < 0 || xx >= p.W) continue;
                        part = fmaf(xb[(int64_t)i * HW + yy * p.W + xx], p.wp[((int64_t)i * 9 + ky * 3 + kx) * p.Cout + co], part);
                    }
                }
                acc = fmaf(part, sb[i], acc);
            }
            float v = acc * (p.d ? p.d[img * p.Cout + co] : 1.f);
            if (p.noise) v = fmaf(nw, p.noise[img * p.noise_bstride + rem], v);
            if (p.bias) v += p.bias[co];
            if (p.act) v = lrelu_gain(v, p.slope, p.gain);
            p.y[idx] = v;
        } else {
            const int co = (int)(r / (4 * RP));
            const int r2 = (int)(r - (int64_t)co * 4 * RP);
            const int ph = r2 / RP, r3 = r2 - ph * RP;
            const int a = r3 / p.P, b = r3 - a * p.P;
            const int oy = 2 * a + (ph >> 1), ox = 2 * b + (ph & 1);
            if (oy <= 2 * p.H && ox <= 2 * p.W) {
                for (int i = 0; i < p.Cin; ++i) {
                    float part = 0.f;
                    for (int ky = (oy & 1); ky < 3; ky += 2) {
                        const int yy = (oy - ky) >> 1;
                        if (oy - ky < 0 || yy >= p.H) continue;
                        for (int kx = (ox & 1); kx < 3; kx += 2) {
                            const int xx = (ox - kx) >> 1;
                            if (ox - kx < 0 || xx >= p.W) continue;
                            part = fmaf(xb[(int64_t)i * HW + yy * p.W + xx], p.wp[((int64_t)i * 9 + ky * 3 + kx) * p.Cout + co], part);
                        }
                    }
                    acc = fmaf(part, sb[i], acc);
                }
                acc *= (p.d ? p.d[img * p.Cout + co] : 1.f);
            }
            p.y[idx] = acc;
        }
    }
}

// ---------------------------------------------------------------- prepack
// w [Cout,Cin,KK] -> wp [Cin,KK,Cout] * scale ; q [Cout,Cin] = sum_t (w*scale)^2 ; qt = q^T [Cin,Cout]
__global__ __launch_bounds__(256) void prepack_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                     float* __restrict__ q, float* __restrict__ qt, int Cout, int Cin,
                                                     int KK, float scale) {
    const int64_t n = (int64_t)Cout * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % Cout);  // o fastest: coalesced wp writes
        const int i = (int)(idx / Cout);
        const float* src = w + ((int64_t)o * Cin + i) * KK;
        float ss = 0.f;
        for (int t = 0; t < KK; ++t) {
            const float v = src[t] * scale;
            wp[((int64_t)i * KK + t) * Cout + o] = v;
            ss = fmaf(v, v, ss);
        }
        if (q) q[(int64_t)o * Cin + i] = ss;
        if (qt) qt[(int64_t)i * Cout + o] = ss;
    }
}

// ---------------------------------------------------------------- ToRGB
// 1x1 modulated conv to 3 channels without demodulation + bias + FIR-upsampled skip.
// HBM-bound: x is read exactly once (16 B/lane when V=4).  A block owns QB pixel groups of one
// image; its 256/QB channel groups split Cin and are summed through LDS.
template <int V>
__global__ __launch_bounds__(256) void torgb_kernel(const float* __restrict__ x, const float* __restrict__ w_rgb,
                                                   const float* __restrict__ s, const float* __restrict__ bias,
                                                   const float* __restrict__ skip, const float* __restrict__ fir,
                                                   float* __restrict__ y, int B, int Cin, int H, int W, int QB,
                                                   int qb_shift, float scale) {
    __shared__ float red[256 * 3 * V];
    __shared__ float kf[16];
    const int tid = threadIdx.x;
    if (tid < 16) kf[tid] = fir ? fir[15 - tid] : 0.f;  // flipped taps
    const int HW = H * W;
    const int np = (HW + V - 1) / V;
    const int qi = tid & (QB - 1), cg = tid >> qb_shift, CG = 256 >> qb_shift;
    const int grp = blockIdx.x * QB + qi;
    const int b = blockIdx.y;
    const bool active = grp < np;
    float acc[3][V];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[j][v] = 0.f;
    if (active) {
        const float* xb = x + (int64_t)b * Cin * HW + (int64_t)grp * V;
        const float* sb = s + (int64_t)b * Cin;
#pragma unroll 4
        for (int i = cg; i < Cin; i += CG) {
            const float sv = sb[i] * scale;
            const float c0 = w_rgb[i] * sv, c1 = w_rgb[Cin + i] * sv, c2 = w_rgb[2 * Cin + i] * sv;
            float xv[V];
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(xb + (int64_t)i * HW);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            } else {
                xv[0] = xb[(int64_t)i * HW];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[0][v] = fmaf(c0, xv[v], acc[0][v]);
                acc[1][v] = fmaf(c1, xv[v], acc[1][v]);
                acc[2][v] = fmaf(c2, xv[v], acc[2][v]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) red[(j * V + v) * 256 + tid] = acc[j][v];
    __syncthreads();
    if (cg == 0 && active) {
        const int Hs = H >> 1, Ws = W >> 1;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float out[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float t = 0.f;
                for (int g = 0; g < CG; ++g) t += red[(j * V + v) * 256 + g * QB + qi];
                t += bias ? bias[j] : 0.f;
                if (skip) {
                    const int pix = grp * V + v;
                    const int oy = pix / W, ox = pix - oy * W;
                    const float* sp = skip + ((int64_t)b * 3 + j) * Hs * Ws;
                    // upfirdn2d(up=2, pad=(2,1)): sample (m,n) of skip sits at padded position (2m+2, 2n+2)
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) {
                        const int uy = oy + ky - 2;
                        if (uy < 0 || (uy & 1) || (uy >> 1) >= Hs) continue;
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const int ux = ox + kx - 2;
                            if (ux < 0 || (ux & 1) || (ux >> 1) >= Ws) continue;
                            t = fmaf(sp[(uy >> 1) * Ws + (ux >> 1)], kf[ky * 4 + kx], t);
                        }
                    }
                }
                out[v] = t;
            }
            float* dst = y + ((int64_t)b * 3 + j) * HW + (int64_t)grp * V;
            if (V == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
            } else {
                dst[0] = out[0];
            }
        }
    }
}

template <int MODE, int WM, int WN, int MI, int NI, int EX, int OCC>
static int launch_modconv(ModconvParams& p, hipStream_t stream) {
    constexpr int NT = WM * MI * 32, PT = WN * NI * 32;
    p.n_cout_tiles = (p.Cout + NT - 1) / NT;
    p.n_pix_tiles = (int)((p.total_pix + PT - 1) / PT);
    // q-range a tile of PT consecutive outputs can touch (see file header)
    int xlen;
    if (MODE == SGDFR_MODE_PLAIN3) {
        // tiles start at multiples of PT: count the row / image boundaries a tile can straddle
        const int HW = p.H * p.W;
        const int rows_crossed = (p.W % PT == 0) ? 0 : ((PT % p.W == 0) ? PT / p.W - 1 : (PT - 1) / p.W + 1);
        const int imgs_crossed = (HW % PT == 0) ? 0 : ((PT % HW == 0) ? PT / HW - 1 : (PT - 1) / HW + 1);
        xlen = (PT - 1) + rows_crossed + imgs_crossed * p.P + 2 * p.P + 3;
    } else if (MODE == SGDFR_MODE_DOWN3) {
        const int HW = p.H * p.W;
        const int rows_crossed = (p.W % PT == 0) ? 0 : ((PT % p.W == 0) ? PT / p.W - 1 : (PT - 1) / p.W + 1);
        const int imgs_crossed = (HW % PT == 0) ? 0 : ((PT % HW == 0) ? PT / HW - 1 : (PT - 1) / HW + 1);
        xlen = (PT - 1) + rows_crossed + imgs_crossed * (p.P + 1) + p.P + 2;
    } else {
        xlen = PT + p.P + 2;
    }
    p.seglen = xlen;
    p.segpitch = 0;
    if (MODE == SGDFR_MODE_PLAIN3 && p.W % PT == 0 && 3 * (PT + 2) < xlen) {
        // wide image: a tile is a piece of ONE row, so stage just the three row segments it touches
        p.seglen = PT + 2;
        p.segpitch = p.P;
        xlen = 3 * (PT + 2);
    }
    SGDFR_REQUIRE(xlen <= EX * 256, "modconv: staged range %d exceeds %d (W=%d too wide for this tile)", xlen, EX * 256,
                  p.W);
    p.xlen = xlen;
    p.xs = (xlen + 3) & ~3;
    constexpr int NPL = (MODE == SGDFR_MODE_DOWN3) ? 4 : 1;
    const size_t lds = 2 * (size_t)(CK * NPL * p.xs + CK * 9 * NT) * sizeof(float);   // double-buffered stages
    SGDFR_REQUIRE(p.total_pix + 4ll * p.P + 8 < (1ll << 31), "modconv: batch too large for 32-bit pixel indices");
    const int64_t nblk = (int64_t)p.n_cout_tiles * p.n_pix_tiles * p.ksplit;
    SGDFR_REQUIRE(nblk < (1ll << 31), "modconv: grid too large");
    SGDFR_REQUIRE(lds <= 160 * 1024, "modconv: LDS request %zu too large", lds);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(modconv_mfma_kernel<MODE, WM, WN, MI, NI, EX, OCC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return check_launch("modconv(lds attribute)");
    hipLaunchKernelGGL((modconv_mfma_kernel<MODE, WM, WN, MI, NI, EX, OCC>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
    return check_launch("modconv2d_fwd");
}

}  // namespace sgdfr

using namespace sgdfr;

extern "C" int sgdfr_modconv_prepack_f32(const float* weight, float* wp, float* q, float* qt, int Cout, int Cin, int k,
                                         void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && k > 0, "prepack: bad shape %d %d %d", Cout, Cin, k);
    SGDFR_REQUIRE(weight && wp, "prepack: null pointer");
    const int64_t n = (int64_t)Cout * Cin;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prepack_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wp, q, qt, Cout, Cin, k * k,
                       1.0f / sqrtf((float)Cin * k * k));
    return check_launch("modconv_prepack");
}

extern "C" int sgdfr_modconv2d_fwd_f32(const float* x, int64_t x_bstride, const float* wp, const float* s,
                                       const float* d, const float* noise, int64_t noise_bstride, const float* noise_w,
                                       const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int mode,
                                       int act, float slope, float gain, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B,
                  Cin, Cout, H, W);
    SGDFR_REQUIRE(mode == SGDFR_MODE_PLAIN3 || mode == SGDFR_MODE_UP3 || mode == SGDFR_MODE_DOWN3,
                  "modconv: unknown mode %d", mode);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && wp && s && y, "modconv: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "modconv: noise without noise_w");
    SGDFR_REQUIRE(x_bstride == 0 || x_bstride >= (int64_t)Cin * H * W * (mode == SGDFR_MODE_DOWN3 ? 0 : 1) +
                                                     (mode == SGDFR_MODE_DOWN3 ? (int64_t)Cin * 4 * (H + 1) * (W + 1) : 0),
                  "modconv: x_bstride too small");
    ModconvParams p{};
    p.x = x; p.x_bstride = x_bstride; p.wp = wp; p.s = s; p.d = d;
    p.noise = noise; p.noise_bstride = noise_bstride; p.noise_w = noise_w; p.bias = bias; p.y = y;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    const bool mfma_ok = (Cin % 4 == 0) && (Cout % 4 == 0) &&
                         (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(wp)) & 15) == 0);
    p.act = act; p.slope = slope; p.gain = gain;
    p.ksplit = 1; p.split_stride = 0;
    hipStream_t st = as_stream(stream);
    if (!mfma_ok) {
        const int64_t total = (mode == SGDFR_MODE_UP3) ? (int64_t)B * Cout * 4 * (H + 1) * (W + 1)
                                                       : (int64_t)B * Cout * H * W;
        int64_t g = (total + 255) / 256;
        if (g > 256 * 16) g = 256 * 16;
        hipLaunchKernelGGL(modconv_direct_kernel, dim3((int)g), dim3(256), 0, st, p, mode);
        return check_launch("modconv2d_fwd(direct)");
    }
    if (mode == SGDFR_MODE_PLAIN3) {
        p.total_pix = (int64_t)B * H * W;
        const int64_t big_blocks = ((p.total_pix + 127) / 128) * ((Cout + 127) / 128);
        if (Cout % 128 == 0 && big_blocks >= 512) return launch_modconv<SGDFR_MODE_PLAIN3, 2, 2, 2, 2, 3, 3>(p, st);
        if (Cout % 64 == 0 && p.total_pix >= 256 * 1024)
            return launch_modconv<SGDFR_MODE_PLAIN3, 1, 4, 2, 2, 4, 3>(p, st);                    // NT 64, PT 256
        if (Cout % 64 == 0 && p.total_pix >= 128 * 512)
            return launch_modconv<SGDFR_MODE_PLAIN3, 2, 2, 1, 2, 3, 3>(p, st);                    // NT 64, PT 128
        if (Cout > 32) return launch_modconv<SGDFR_MODE_PLAIN3, 2, 2, 1, 1, 3, 3>(p, st);            // NT 64, PT 64
        return launch_modconv<SGDFR_MODE_PLAIN3, 1, 4, 1, 1, 3, 3>(p, st);                           // NT 32, PT 128
    } else if (mode == SGDFR_MODE_DOWN3) {
        p.total_pix = (int64_t)B * H * W;
        const int64_t big_blocks = ((p.total_pix + 127) / 128) * ((Cout + 127) / 128);
        if (Cout % 128 == 0 && big_blocks >= 512) return launch_modconv<SGDFR_MODE_DOWN3, 2, 2, 2, 2, 3, 2>(p, st);
        if (Cout > 32) return launch_modconv<SGDFR_MODE_DOWN3, 2, 2, 1, 1, 3, 2>(p, st);           // NT 64, PT 64
        return launch_modconv<SGDFR_MODE_DOWN3, 1, 4, 1, 1, 3, 2>(p, st);                          // NT 32, PT 128
    } else {
        p.total_pix = (int64_t)B * (H + 1) * (W + 1);
        const int64_t big_blocks = ((p.total_pix + 63) / 64) * ((Cout + 127) / 128);
        if (Cout % 128 == 0 && big_blocks >= 512) return launch_modconv<SGDFR_MODE_UP3, 4, 1, 1, 2, 3, 2>(p, st);  // NT 128, PT 64
        if (Cout > 32) return launch_modconv<SGDFR_MODE_UP3, 2, 2, 1, 1, 3, 3>(p, st);               // NT 64, PT 64
        return launch_modconv<SGDFR_MODE_UP3, 1, 4, 1, 1, 3, 3>(p, st);                              // NT 32, PT 128
    }
}

// ToRGB finish: the 1x1 conv was accumulated per cout tile in the feeding conv's epilogue (split.hip); what is left is
// y[b,j,p] = sum_t part[b, t*3+j, p] + bias[j] + upfirdn2d(skip[b,j], fir, up=2, pad=(2,1))[p].  One thread = 4 pixels of
// a row x 3 channels; the polyphase upsample reads a 2x3 skip window per channel.
// U8: the image leaves as uint8 HWC with the reference's tensor_to_image scaling (libs/utilities/image_utils.py:87-110, the
// same expression as image_to_u8_kernel) instead of fp32 planes: y8[b][oy][x_off + ox][c], `pitch` bytes per row, so the
// frame can be one panel of a wider video grid; swap_rb writes channel 2-c (the writers' cvtColor).
template <bool U8>
__global__ __launch_bounds__(256) void torgb_finish_kernel(const float* __restrict__ part, int T, const float* __restrict__ bias,
                                                          const float* __restrict__ skip, const float* __restrict__ fir,
                                                          float* __restrict__ y, unsigned char* __restrict__ y8, int64_t pitch,
                                                          int x_off, int swap_rb, int B, int H, int W) {
    __shared__ float kf[16];
    if (threadIdx.x < 16) kf[threadIdx.x] = fir ? fir[15 - threadIdx.x] : 0.f;  // flipped taps
    __syncthreads();
    const int HW = H * W, W4 = W >> 2;
    const int64_t n = (int64_t)B * H * W4;
    const int Hs = H >> 1, Ws = W >> 1;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(idx % W4);
        const int oy = (int)((idx / W4) % H);
        const int b = (int)(idx / ((int64_t)W4 * H));
        const int ox = xq * 4;
        float px[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float bj = bias ? bias[j] : 0.f;
            float4 acc = make_float4(bj, bj, bj, bj);
            for (int t = 0; t < T; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)b * T * 3 + t * 3 + j) * HW + (int64_t)oy * W + ox);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (skip) {
                // upfirdn2d(up=2, pad=(2,1)): skip sample (m,n) sits at padded (2m+2, 2n+2); output (oy,ox) sums taps (ky,kx) with
                // oy+ky-2 = 2m, ox+kx-2 = 2n  ->  ky of parity oy&1, kx of parity ox&1: 2x2 taps per output
                const float* sp = skip + ((int64_t)b * 3 + j) * Hs * Ws;
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ky = (oy & 1) + 2 * a;
                    const int m = (oy + ky - 2) >> 1;
                    if (oy + ky - 2 < 0 || m >= Hs) continue;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int xx = ox + v;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int kx = (xx & 1) + 2 * c;
                            const int nn = (xx + kx - 2) >> 1;
                            if (xx + kx - 2 < 0 || nn >= Ws) continue;
                            o[v] = fmaf(sp[m * Ws + nn], kf[ky * 4 + kx], o[v]);
                        }
                    }
                }
                acc.x += o[0]; acc.y += o[1]; acc.z += o[2]; acc.w += o[3];
            }
            if (!U8) *reinterpret_cast<float4*>(y + ((int64_t)b * 3 + j) * HW + (int64_t)oy * W + ox) = acc;
            px[0][j] = acc.x; px[1][j] = acc.y; px[2][j] = acc.z; px[3][j] = acc.w;
        }
        if (U8) {
            unsigned char q[12];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f = fminf(fmaxf(px[v][swap_rb ? 2 - c : c], -1.f), 1.f);
                    f = (f + 1.f) / (2.f + 1e-5f) * 255.f;
                    q[v * 3 + c] = (unsigned char)f;
                }
            unsigned* dst = reinterpret_cast<unsigned*>(y8 + ((int64_t)b * H + oy) * pitch + (int64_t)(x_off + ox) * 3);   // 12-byte runs: 4-aligned
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dst[k] = (unsigned)q[4 * k] | ((unsigned)q[4 * k + 1] << 8) | ((unsigned)q[4 * k + 2] << 16) | ((unsigned)q[4 * k + 3] << 24);
        }
    }
}

extern "C" int sgdfr_torgb_finish_f32(const float* part, int T, const float* bias, const float* skip, const float* fir, float* y,
                                      int B, int H, int W, void* stream) {
    SGDFR_REQUIRE(B >= 0 && T > 0 && H > 0 && W > 0 && W % 4 == 0, "torgb_finish: bad shape B=%d T=%d H=%d W=%d (W %% 4 == 0)", B, T,
                  H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(part && y, "torgb_finish: null pointer");
    SGDFR_REQUIRE(!skip || (fir && H % 2 == 0), "torgb_finish: skip needs fir taps and even H");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "torgb_finish: 16-byte alignment");
    int64_t g = ((int64_t)B * H * (W / 4) + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(torgb_finish_kernel<false>, dim3((int)g), dim3(256), 0, as_stream(stream), part, T, bias, skip, fir, y, nullptr,
                       (int64_t)0, 0, 0, B, H, W);
    return check_launch("torgb_finish");
}

extern "C" int sgdfr_torgb_finish_u8_f32(const float* part, int T, const float* bias, const float* skip, const float* fir,
                                         unsigned char* y, int64_t row_pitch, int x_offset, int swap_rb, int B, int H, int W,
                                         void* stream) {
    SGDFR_REQUIRE(B >= 0 && T > 0 && H > 0 && W > 0 && W % 4 == 0, "torgb_finish_u8: bad shape B=%d T=%d H=%d W=%d (W %% 4 == 0)", B, T,
                  H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(part && y, "torgb_finish_u8: null pointer");
    SGDFR_REQUIRE(!skip || (fir && H % 2 == 0), "torgb_finish_u8: skip needs fir taps and even H");
    SGDFR_REQUIRE(x_offset >= 0 && x_offset % 4 == 0 && row_pitch >= (int64_t)(x_offset + W) * 3 && row_pitch % 4 == 0,
                  "torgb_finish_u8: panel at pixel %d (multiple of 4) of %d does not fit rows of %lld bytes (multiple of 4)", x_offset, W,
                  (long long)row_pitch);
    SGDFR_REQUIRE((reinterpret_cast<uintptr_t>(part) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 3) == 0, "torgb_finish_u8: alignment");
    int64_t g = ((int64_t)B * H * (W / 4) + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(torgb_finish_kernel<true>, dim3((int)g), dim3(256), 0, as_stream(stream), part, T, bias, skip, fir, nullptr, y,
                       row_pitch, x_offset, swap_rb, B, H, W);
    return check_launch("torgb_finish_u8");
}

extern "C" int sgdfr_torgb_fwd_f32(const float* x, const float* w_rgb, const float* s, const float* bias,
                                   const float* skip, const float* fir, float* y, int B, int Cin, int H, int W,
                                   void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && H > 0 && W > 0, "torgb: bad shape %d %d %d %d", B, Cin, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && w_rgb && s && y, "torgb: null pointer");
    SGDFR_REQUIRE(!skip || (fir && H % 2 == 0 && W % 2 == 0), "torgb: skip needs fir taps and even H, W");
    const int HW = H * W;
    const bool vec = (HW % 4 == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    const int V = vec ? 4 : 1;
    const int np = (HW + V - 1) / V;
    int QB = 4, sh = 2;
    while (QB < 64 && QB < np) { QB <<= 1; ++sh; }
    dim3 grid((np + QB - 1) / QB, B);
    const float scale = 1.0f / sqrtf((float)Cin);
    if (vec)
        hipLaunchKernelGGL(torgb_kernel<4>, grid, dim3(256), 0, as_stream(stream), x, w_rgb, s, bias, skip, fir, y, B,
                           Cin, H, W, QB, sh, scale);
    else
        hipLaunchKernelGGL(torgb_kernel<1>, grid, dim3(256), 0, as_stream(stream), x, w_rgb, s, bias, skip, fir, y, B,
                           Cin, H, W, QB, sh, scale);
    return check_launch("torgb_fwd");
}

// ---- K-sliced variant for launches that cannot fill the chip (small batch x small resolution) --------------------
// The (64 cout x 64 pixel)-tile grid is replicated `splits` times; slice k accumulates input channels
// [k*Cin/splits, (k+1)*Cin/splits) and writes d * partial sums to partials[k]; a second launch adds the slices in a
// fixed order and applies noise / bias / activation (PLAIN3) or just adds (UP3 planes).
extern "C" int sgdfr_modconv2d_splitk_hint(int B, int Cin, int Cout, int H, int W, int mode) {
    if (Cin % 4 || Cout % 4 || Cout <= 32 || mode == SGDFR_MODE_DOWN3) return 1;
    const int64_t pix = (mode == SGDFR_MODE_UP3) ? (int64_t)B * (H + 1) * (W + 1) : (int64_t)B * H * W;
    const int64_t blocks = ((pix + 63) / 64) * ((Cout + 63) / 64);
    if (blocks >= 192) return 1;
    int s = (int)(1024 / blocks);           // aim at ~4 K-slice blocks per CU
    const int max_by_k = Cin / CK / 8;      // keep at least 8 stages per slice
    if (s > max_by_k) s = max_by_k;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

extern "C" int sgdfr_modconv2d_splitk_f32(const float* x, int64_t x_bstride, const float* wp, const float* s,
                                          const float* d, const float* noise, int64_t noise_bstride,
                                          const float* noise_w, const float* bias, float* y, float* partials, int splits,
                                          int B, int Cin, int Cout, int H, int W, int mode, int act, float slope, float gain,
                                          void* stream) {
    SGDFR_REQUIRE(B > 0 && Cin > 0 && Cout > 32 && H > 0 && W > 0, "modconv_splitk: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B,
                  Cin, Cout, H, W);
    SGDFR_REQUIRE(mode == SGDFR_MODE_PLAIN3 || mode == SGDFR_MODE_UP3, "modconv_splitk: mode must be PLAIN3 or UP3");
    SGDFR_REQUIRE(splits >= 2 && splits <= Cin / CK, "modconv_splitk: splits %d not in 2..%d", splits, Cin / CK);
    SGDFR_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "modconv_splitk: Cin and Cout must be multiples of 4");
    SGDFR_REQUIRE(x && wp && s && y && partials, "modconv_splitk: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "modconv_splitk: noise without noise_w");
    ModconvParams p{};
    p.x = x; p.x_bstride = x_bstride; p.wp = wp; p.s = s; p.d = d;
    p.y = partials;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    p.ksplit = splits;
    hipStream_t st = as_stream(stream);
    int64_t n;
    int rc;
    if (mode == SGDFR_MODE_PLAIN3) {
        p.total_pix = (int64_t)B * H * W;
        n = (int64_t)B * Cout * H * W;
        p.split_stride = n;
        rc = launch_modconv<SGDFR_MODE_PLAIN3, 2, 2, 1, 1, 3, 3>(p, st);
    } else {
        p.total_pix = (int64_t)B * (H + 1) * (W + 1);
        n = (int64_t)B * Cout * 4 * (H + 1) * (W + 1);
        p.split_stride = n;
        rc = launch_modconv<SGDFR_MODE_UP3, 2, 2, 1, 1, 3, 3>(p, st);
    }
    if (rc) return rc;
    const bool plain = mode == SGDFR_MODE_PLAIN3;
    return launch_splitk_reduce(partials, splits, n, plain ? noise : nullptr, noise_bstride, noise_w, plain ? bias : nullptr, y,
                                Cout, plain ? H * W : 1, plain ? act : 0, slope, gain, st);
}
