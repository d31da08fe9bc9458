// upfirdn2d (zero-stuff up, pad/crop, 2-D FIR with the flipped kernel, decimate) and the
// blur + noise + bias + leaky-ReLU pass that finishes an upsampling modulated conv.
// Both are HBM-bound: every output is written once, inputs are read once from HBM and the
// FIR overlap is served by L1/L2 (generic op) or by registers across the 2x2 output quad (blur).
#include "common.h"
#include "wsplit_common.h"      // the fp8 cross-term chunks of SGDFR_SPLIT_FP16F8 (Winograd hand-over only)

namespace sgdfr {

constexpr int kMaxTaps = 16;  // per axis

struct UpfirdnParams {
    const float* x;
    const float* k;
    float* y;
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

// Generic path: one thread per output element (minor fastest, then x: coalesced stores), polyphase
// tap stepping so only the kh/up * kw/up taps that hit real samples are visited.
// The FIR taps live in LDS (broadcast reads).
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirdnParams p) {
    __shared__ float sk[kMaxTaps * kMaxTaps];
    for (int i = threadIdx.x; i < p.kh * p.kw; i += blockDim.x) sk[i] = p.k[i];
    __syncthreads();
    const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int mi = (int)(idx % p.minor);
        int64_t r = idx / p.minor;
        const int ox = (int)(r % p.out_w);
        r /= p.out_w;
        const int oy = (int)(r % p.out_h);
        const int mj = (int)(r / p.out_h);
        // position of tap (ky,kx) in up-sampled coordinates: uy = oy*down_y + ky - pad_y0, must be a multiple of up_y
        const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
        int ky0 = (-by) % p.up_y;
        if (ky0 < 0) ky0 += p.up_y;
        int kx0 = (-bx) % p.up_x;
        if (kx0 < 0) kx0 += p.up_x;
        const float* xp = p.x + (int64_t)mj * p.in_h * p.in_w * p.minor + mi;
        float acc = 0.f;
        for (int ky = ky0; ky < p.kh; ky += p.up_y) {
            const int uy = by + ky;
            if (uy < 0) continue;
            const int iy = uy / p.up_y;
            if (iy >= p.in_h) break;
            for (int kx = kx0; kx < p.kw; kx += p.up_x) {
                const int ux = bx + kx;
                if (ux < 0) continue;
                const int ix = ux / p.up_x;
                if (ix >= p.in_w) break;
                acc = fmaf(xp[((int64_t)iy * p.in_w + ix) * p.minor], sk[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)], acc);
            }
        }
        p.y[idx] = acc;
    }
}

// The same generic path for the other dtypes of the reference's dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF, upfirdn2d_kernel.cu:225):
// T = storage type, A = accumulator (float for half, double for double); taps straight from global memory (L1-resident).
template <typename T, typename A>
__global__ __launch_bounds__(256) void upfirdn2d_typed_kernel(const T* __restrict__ x, const T* __restrict__ k, T* __restrict__ y,
                                                             UpfirdnParams p) {
    const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int mi = (int)(idx % p.minor);
        int64_t r = idx / p.minor;
        const int ox = (int)(r % p.out_w);
        r /= p.out_w;
        const int oy = (int)(r % p.out_h);
        const int mj = (int)(r / p.out_h);
        const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
        int ky0 = (-by) % p.up_y;
        if (ky0 < 0) ky0 += p.up_y;
        int kx0 = (-bx) % p.up_x;
        if (kx0 < 0) kx0 += p.up_x;
        const T* xp = x + (int64_t)mj * p.in_h * p.in_w * p.minor + mi;
        A acc = (A)0;
        for (int ky = ky0; ky < p.kh; ky += p.up_y) {
            const int uy = by + ky;
            if (uy < 0) continue;
            const int iy = uy / p.up_y;
            if (iy >= p.in_h) break;
            for (int kx = kx0; kx < p.kw; kx += p.up_x) {
                const int ux = bx + kx;
                if (ux < 0) continue;
                const int ix = ux / p.up_x;
                if (ix >= p.in_w) break;
                acc += (A)xp[((int64_t)iy * p.in_w + ix) * p.minor] * (A)k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
            }
        }
        y[idx] = (T)acc;
    }
}

// Blur after the stride-2 transposed conv, reading the parity planes written by MODE_UP3.
//   t: [planes, 4, H+1, W+1], plane ph = 2*(row&1)+(col&1) holds T[row,col] at [row>>1, col>>1]
//   y: [planes, 2H, 2W];  y[oy,ox] = sum_{ky,kx} Tpad[oy+ky, ox+kx] * K[3-ky][3-kx],  Tpad[i,j] = T[i-1,j-1]
// One thread owns a vertical strip of QV 2x2 output quads and slides a 5-row x 5-col register window of
// T down the strip: 2 new T rows (10 coalesced dword loads) per quad after the first 5 rows, two float2
// stores per quad.  Lanes run along x so neighbouring windows share cache lines.
// Epilogue: + noise_w*noise + bias[c], leaky-ReLU * gain.
constexpr int BLUR_QV = 4;

__global__ __launch_bounds__(256) void blur_bias_act_kernel(const float* __restrict__ t, const float* __restrict__ fir,
                                                           const float* __restrict__ noise, int64_t noise_bstride,
                                                           const float* __restrict__ noise_w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B,
                                                           int C, int H, int W, int act, float slope, float gain,
                                                           unsigned* __restrict__ y_absmax, int wave_planes) {
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = fir[15 - i];  // flipped: kf[ky*4+kx] = K[3-ky][3-kx] (uniform -> SGPRs)
    const int GW = W + 1, GH = H + 1;
    const int64_t plane_t = (int64_t)4 * GH * GW;
    const int HS = (H + BLUR_QV - 1) / BLUR_QV;  // strips per column
    const int64_t strips = (int64_t)B * C * HS * W;
    const float nw = (noise && noise_w) ? noise_w[0] : 0.f;
    const int OW = 2 * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < strips;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % W);
        int64_t r = idx / W;
        const int ms = (int)(r % HS) * BLUR_QV;
        const int64_t pl = r / HS;  // b*C + c
        const float* tp = t + pl * plane_t;
        const int c = (int)(pl % C);
        const int b = (int)(pl / C);
        const float bv = bias ? bias[c] : 0.f;
        // column descriptors of the 5-wide window: T col 2n-1+v -> plane parity (v+1)&1, plane col (2n-1+v)>>1
        int coff[5];
        bool cok[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const int tc = 2 * n - 1 + v;
            cok[v] = tc >= 0;
            coff[v] = (tc & 1) * GH * GW + (tc >> 1);
        }
        unsigned gm = 0u;       // bit pattern of max |y| over this strip (y_absmax: the range plan of the conv that reads y)
        float win[5][5];
        auto load_row = [&](int tr, float (&dst)[5]) {  // T row tr in [-1, 2H+1]; rows 2H+1 are stored zeros
            const bool rok = tr >= 0;
            const int roff = (tr & 1) * 2 * GH * GW + (tr >> 1) * GW;
#pragma unroll
            for (int v = 0; v < 5; ++v) dst[v] = (rok && cok[v]) ? tp[roff + coff[v]] : 0.f;
        };
#pragma unroll
        for (int u = 0; u < 3; ++u) load_row(2 * ms - 1 + u, win[u]);
#pragma unroll
        for (int qv = 0; qv < BLUR_QV; ++qv) {
            const int m = ms + qv;
            if (m >= H) break;
            load_row(2 * m + 2, win[3]);
            load_row(2 * m + 3, win[4]);
            float nzq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            if (noise) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const float2 t2 = *reinterpret_cast<const float2*>(noise + (int64_t)b * noise_bstride + (int64_t)(2 * m + rr) * OW + 2 * n);
                    nzq[rr][0] = t2.x; nzq[rr][1] = t2.y;
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                float o0 = 0.f, o1 = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        o0 = fmaf(win[rr + ky][kx], kf[ky * 4 + kx], o0);
                        o1 = fmaf(win[rr + ky][1 + kx], kf[ky * 4 + kx], o1);
                    }
                const int oy = 2 * m + rr;
                float v0 = o0 + bv, v1 = o1 + bv;
                v0 = fmaf(nw, nzq[rr][0], v0);
                v1 = fmaf(nw, nzq[rr][1], v1);
                if (act) {
                    v0 = lrelu_gain(v0, slope, gain);
                    v1 = lrelu_gain(v1, slope, gain);
                }
                *reinterpret_cast<float2*>(y + (pl * 2 * H + oy) * OW + 2 * n) = make_float2(v0, v1);
                gm = max(gm, max(__float_as_uint(fabsf(v0)), __float_as_uint(fabsf(v1))));
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) win[u][v] = win[u + 2][v];
        }
        if (y_absmax) {
            // one word per (image, channel) plane, as sgdfr_act_grad_reduce_f32 keeps them.  wave_planes: a wave's 64 strips lie
            // inside one plane and all 64 lanes are here (strips per plane and the grid stride are multiples of 64): one atomic
            // per wave; otherwise (tiny planes) one per strip.
            if (wave_planes) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));      // (integer max: NaN / Inf patterns survive)
                if ((threadIdx.x & 63) == 0 && gm != 0u) atomicMax(y_absmax + pl, gm);
            } else if (gm != 0u) {
                atomicMax(y_absmax + pl, gm);
            }
        }
    }
}


// The same blur + noise + bias + leaky-ReLU, but the result is multiplied by the NEXT layer's modulation and written in
// that layer's split input form ("XS": [B][C/8][hi,lo][2H*2W][8] 16-bit pairs, see split.hip) instead of fp32 NCHW.
// A block = one 8-channel group x 8 output rows x 128 output columns: thread (channel, quad column) slides the same
// 5x5 window down 4 quads, the fp32 results meet in LDS as [row][px][8 channels], and every thread then scales, splits and
// writes whole 16-byte pixel chunks (2 KB runs per row and part).
typedef _Float16 bl_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bl_f32x2 __attribute__((ext_vector_type(2)));

__device__ unsigned int g_blur_saturated = 0;    // operand pairs clamped to the fp16 range (see sgdfr_split_saturation_count)

template <int ET>
__device__ __forceinline__ void blur_split2(float a, float b, unsigned& hi, unsigned& lo, unsigned& sat) {     // as split.hip's split_pair
    if (ET == SGDFR_SPLIT_FP16 || ET == SGDFR_SPLIT_FP16F8) {
        sat += (!(fabsf(a) <= 65504.f) || !(fabsf(b) <= 65504.f)) ? 1u : 0u;      // NaN counts too; flushed once per thread: no branch per pair
        a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
        b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
        const bl_f16x2 h = __builtin_convertvector((bl_f32x2){a, b}, bl_f16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const bl_f32x2 hf = __builtin_convertvector(h, bl_f32x2);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((bl_f32x2){a - hf[0], b - hf[1]}, bl_f16x2));
    } else {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector((bl_f32x2){a, b}, bl_bf16x2));
        const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((bl_f32x2){a - ha, b - hb}, bl_bf16x2));
    }
}

// QC: quad columns per channel (64: a full wave per channel, for W >= 64; 32 / 16 / 8 / 4 for the narrower levels, so no lane
// idles on a 4- or 8-wide plane); NG: (image, 8-channel group, row tile) groups per block, 256 / (8 * QC) for QC < 32 -- the block
// stays 256 threads wide on the small levels (4 -> 8: 37 -> 9 us, 8 -> 16: 58 -> 17 us per launch at B=64); block = 8 * QC * NG
// threads.
// A block walks `nseg` consecutive row segments (BLUR_QV quad rows each) of its column tile with the SAME sliding window,
// handing each segment over through LDS: only the first segment re-reads the 3 plane rows above it, so the planes are
// fetched 1 + 3/(8*nseg) times instead of 1.375 (rocprofv3 FETCH_SIZE of the 128 -> 256 level: 2.03 x the planes with
// nseg = 1, mostly served by the Infinity Cache -- the run time is the same, the DRAM traffic is not).
// Round 3, tried on the two wide levels and dropped (same-process A/B at B=64, scripts/blur_ab.py, all bit-identical): 4-wave
// blocks (QC = 32, four per CU) 491 -> 519 us; each lane loading only its own super-pixel column and taking the window's three
// neighbour entries from lanes -1 / +1 by DPP wave shifts (2 loads per row instead of 5 overlapping ones; the wave's edge
// lanes fetch the outside column in divergent branches) 491 -> 825 us; a producer / consumer block (8 waves load + filter into
// one of two LDS tiles, 8 waves split + store the other: a wave's in-order vmcnt then holds loads OR stores) 514 -> 682 us --
// half as many loading waves per CU.  The kernel is bound by loads in flight and DRAM locality of its 32 streams per block,
// not by load instructions or by the shared load / store queue.
// WINO = 2 | 4: the hand-over writes the Winograd F(WINO,3) input transform of the result instead ("WS" form of wsplit.hip:
// [B][C/8][t WINO+2][hi,lo][2H * 2W / WINO][8], V = B^T (y * s_next) per tile of WINO outputs; bit-identical to
// sgdfr_to_wsplit_f32 of the fp32 result).  A thread takes one tile (WINO = 4: three of its six positions) and reads the
// tile's two row neighbours from the LDS tile, so the tile must span whole rows (one column tile: 2W <= 128).
// (WINO variants: 3 waves per SIMD = 168 registers -- at 128 the sliding window spills.  The 8-wave blocks of the 64 -> 128 level then
// fit a CU only once; tried for that level: requesting the next segment's plane rows after the hand-over instead of before it, to
// get back under 128 registers and two blocks per CU: 88 bytes of scratch remained and the level went from 367 to 513 us.)
// IL: the planes arrive interleaved, t [planes][pstride positions][px][py] (what split.hip's MODE_UP3 writes when it is given a
// plane_stride): a super-pixel's four phases are one 16-byte load, so a new super row of the sliding window costs three vector
// loads (8 + 16 + 16 bytes: left neighbour's px = 1 pair, own position, right neighbour) instead of ten dword loads from four
// planes -- four times the bytes in flight per load instruction on a kernel that is bound by loads in flight.
template <int ET, int QC, int NG, int WINO = 0, bool IL = false>
__global__ __launch_bounds__(8 * QC * NG, WINO ? 3 : 4) void blur_split_kernel(const float* __restrict__ t, const float* __restrict__ fir,
                                                        const float* __restrict__ noise, int64_t noise_bstride,
                                                        const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                        const float* __restrict__ s_next, unsigned char* __restrict__ xs,
                                                        int B, int C, int H, int W, int pstride, int nseg, int act, float slope,
                                                        float gain, unsigned* __restrict__ sat_word) {
    // fp32 results of the block's tile as [row 8][px 128][channel 8 (+1 pad: the lanes of a wave write px 2 apart ->
    // 18-dword stride, conflict-free)]; the hand-over to 16-byte chunks happens when the tile is read back.
    // 8 waves = 8 channels, a wave = 64 quad columns (256-byte coalesced plane reads, as the fp32 kernel).
    __shared__ float tile_all[NG][8][2 * QC][9];
    __shared__ float sv_all[NG][8];
    // WINO = 4 on rows of two column tiles (W = 2 QC = 128): a four-pixel Winograd tile reads one pixel on either side, so the
    // block also computes the ONE output column beyond its inner edge (px 2 QC for the left half, px 2 QC - 1 for the right half)
    // -- 64 threads, 16 plane loads each, the main path's tap order and epilogue, so the pixel has the bits the other half gives it
    __shared__ float edge_all[(WINO == 4 && NG == 1) ? 8 : 1][8];
    static_assert(NG == 1 || QC < 32, "several groups per block only for the narrow levels");
    float kf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = fir[15 - i];
    const int GW = W + 1, GH = H + 1;
    const int64_t plane_t = (int64_t)4 * pstride;          // pstride = floats between parity planes (>= GH*GW)
    const int G = C / 8, OW = 2 * W, OHW = 4 * H * W;
    const int col_tiles = (W + QC - 1) / QC, row_tiles = (H + BLUR_QV * nseg - 1) / (BLUR_QV * nseg);
    // Two column tiles per row (W = 2 QC): ids i and i + 8 are the two halves of one row group.  Blocks i and i + 8 run on
    // the same XCD at the same time, so the cache lines the halves share (odd row pitch: every 64-float run straddles lines)
    // are fetched from memory once and found in that XCD's L2 by the other half.
    const bool paired = col_tiles == 2;
    const int64_t n_groups = (int64_t)B * G * row_tiles;
    const int64_t n_tiles = paired ? ((n_groups + 7) / 8) * 16 : ((n_groups + NG - 1) / NG) * col_tiles;      // (NG > 1: one column tile)
    const float nw = (noise && noise_w) ? noise_w[0] : 0.f;
    const float xsc = (ET == SGDFR_SPLIT_FP16 || ET == SGDFR_SPLIT_FP16F8) ? 0.0625f : 1.f;
    const float f8_mul_lo = exp2f((float)WS_F8_XLO), f8_mul_hi = exp2f((float)WS_F8_XHI);
    const int sub = threadIdx.x / (8 * QC), lt = threadIdx.x % (8 * QC);      // group of this thread inside the block, thread inside the group
    const int c8 = lt / QC, nx = lt % QC;
    float (*const tile)[2 * QC][9] = tile_all[sub];
    float* const sv = sv_all[sub];
    unsigned sat = 0;
    for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
        const int ct = paired ? (int)((tile_id >> 3) & 1) : (int)(tile_id % col_tiles);
        int64_t r = paired ? (tile_id >> 4) * 8 + (tile_id & 7) : (tile_id / col_tiles) * NG + sub;
        // (a group beyond the last one -- padding of the last 8 pairs / of the last block's NG groups -- runs the barriers of the
        // loop with its loads and stores masked; for NG = 1 the test is block-uniform)
        if (NG == 1 && r >= n_groups) continue;
        const bool valid = NG == 1 || r < n_groups;
        if (!valid) r = n_groups - 1;
        const int rt = (int)(r % row_tiles);
        r /= row_tiles;
        const int g = (int)(r % G);
        const int b = (int)(r / G);
        const int c = g * 8 + c8;
        const int n = ct * QC + nx;            // quad column
        if (lt < 8) sv[lt] = s_next[(int64_t)b * C + g * 8 + lt] * xsc;
        const float* tp = t + ((int64_t)b * C + c) * plane_t;
        const float bv = bias ? bias[c] : 0.f;
        int coff[5];
        bool cok[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const int tc = 2 * n - 1 + v;
            cok[v] = tc >= 0 && n < W && valid;
            coff[v] = (tc & 1) * pstride + (tc >> 1);
        }
        float win[5][5];
        auto load_row = [&](int tr, float (&dst)[5]) {
            const bool rok = tr >= 0;
            const int roff = (tr & 1) * 2 * pstride + (tr >> 1) * GW;
#if defined(SGDFR_BLUR_PROBE) && SGDFR_BLUR_PROBE == 2      // ablation: no plane loads
#pragma unroll
            for (int v = 0; v < 5; ++v) dst[v] = (float)(roff + v);
#else
#pragma unroll
            for (int v = 0; v < 5; ++v) dst[v] = (rok && cok[v]) ? tp[roff + coff[v]] : 0.f;
#endif
        };
        // The 2 * BLUR_QV new plane rows of a segment are requested together, and those of the NEXT segment right after this
        // segment's arithmetic -- before its hand-over -- so a block's plane reads run under its own stores (measured on the
        // 128 -> 256 level: loads alone 250 us, compute + stores alone 276 us, both in sequence 524 us).
        float rows[2 * BLUR_QV][5];
        // IL: super row a -> T rows 2a (d0) and 2a+1 (d1), window columns 2n-1 .. 2n+3
        const bool c_ok = n < W && valid, l_ok = c_ok && n >= 1;
        auto load_srow = [&](int a, float (&d0)[5], float (&d1)[5]) {
            const float* q = tp + ((int64_t)a * GW + n) * 4;
            const bool rok = a >= 0;
#if defined(SGDFR_BLUR_PROBE) && SGDFR_BLUR_PROBE == 2      // ablation: no plane loads
            for (int v = 0; v < 5; ++v) { d0[v] = (float)(a + v); d1[v] = (float)(n - v); }
            if (rok) return;
#endif
#if defined(SGDFR_BLUR_PROBE) && (SGDFR_BLUR_PROBE == 4 || SGDFR_BLUR_PROBE == 6)      // A/B: non-temporal plane loads
            typedef float bl_f4 __attribute__((ext_vector_type(4)));
            typedef float bl_f2 __attribute__((ext_vector_type(2)));
            float2 lf = make_float2(0.f, 0.f);
            float4 cf = make_float4(0.f, 0.f, 0.f, 0.f), rf = cf;
            if (rok && l_ok) { const bl_f2 t2 = __builtin_nontemporal_load(reinterpret_cast<const bl_f2*>(q - 2)); lf = make_float2(t2[0], t2[1]); }
            if (rok && c_ok) {
                const bl_f4 t4 = __builtin_nontemporal_load(reinterpret_cast<const bl_f4*>(q)), u4 = __builtin_nontemporal_load(reinterpret_cast<const bl_f4*>(q + 4));
                cf = make_float4(t4[0], t4[1], t4[2], t4[3]); rf = make_float4(u4[0], u4[1], u4[2], u4[3]);
            }
#else
            const float2 lf = (rok && l_ok) ? *reinterpret_cast<const float2*>(q - 2) : make_float2(0.f, 0.f);        // (n-1: px 1, py 0 | 1)
            const float4 cf = (rok && c_ok) ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 rf = (rok && c_ok) ? *reinterpret_cast<const float4*>(q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            d0[0] = lf.x; d0[1] = cf.x; d0[2] = cf.z; d0[3] = rf.x; d0[4] = rf.z;
            d1[0] = lf.y; d1[1] = cf.y; d1[2] = cf.w; d1[3] = rf.y; d1[4] = rf.w;
        };
        auto load_segment = [&](int ms) {
            if (IL) {
#pragma unroll
                for (int i = 0; i < BLUR_QV; ++i)
                    if (ms + i < H) load_srow(ms + i + 1, rows[2 * i], rows[2 * i + 1]);
                return;
            }
#pragma unroll
            for (int i = 0; i < 2 * BLUR_QV; ++i)
                if (ms + (i >> 1) < H) load_row(2 * ms + 2 + i, rows[i]);
        };
        {
            const int ms0 = rt * BLUR_QV * nseg;
            if (IL) {
                float drop[5];
                load_srow(ms0 - 1, drop, win[0]);
                load_srow(ms0, win[1], win[2]);
            } else {
#pragma unroll
                for (int u = 0; u < 3; ++u) load_row(2 * ms0 - 1 + u, win[u]);
            }
            if (n < W) load_segment(ms0);
        }
        for (int sg = 0; sg < nseg; ++sg) {
        const int ms = (rt * nseg + sg) * BLUR_QV;           // first quad row of the segment
        if (NG == 1 && ms >= H) break;                        // block-uniform (NG > 1: groups may sit in different row tiles)
        if (n < W && (NG == 1 || ms < H)) {
#pragma unroll
            for (int qv = 0; qv < BLUR_QV; ++qv) {
                const int m = ms + qv;
                if (m >= H) break;
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    win[3][v] = rows[2 * qv][v];
                    win[4][v] = rows[2 * qv + 1][v];
                }
                float nzq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
                if (noise) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const float2 t2 = *reinterpret_cast<const float2*>(noise + (int64_t)b * noise_bstride + (int64_t)(2 * m + rr) * OW + 2 * n);
                        nzq[rr][0] = t2.x; nzq[rr][1] = t2.y;
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    float o0 = 0.f, o1 = 0.f;
#if defined(SGDFR_BLUR_PROBE) && SGDFR_BLUR_PROBE == 3      // ablation: 5 adds per output instead of 16 FMAs (is the kernel VALU-bound?)
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) { o0 += win[rr + ky][ky]; o1 += win[rr + ky][1 + ky]; }
                    o0 += win[rr][3]; o1 += win[rr + 3][1];
                    for (int v = 0; v < 5; ++v) o0 += win[4 - rr][v] * 1e-9f;      // (keep every window entry alive)
#else
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            o0 = fmaf(win[rr + ky][kx], kf[ky * 4 + kx], o0);
                            o1 = fmaf(win[rr + ky][1 + kx], kf[ky * 4 + kx], o1);
                        }
#endif
                    float v0 = fmaf(nw, nzq[rr][0], o0 + bv), v1 = fmaf(nw, nzq[rr][1], o1 + bv);
                    if (act) {
                        v0 = lrelu_gain(v0, slope, gain);
                        v1 = lrelu_gain(v1, slope, gain);
                    }
                    tile[2 * qv + rr][2 * nx][c8] = v0;
                    tile[2 * qv + rr][2 * nx + 1][c8] = v1;
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int v = 0; v < 5; ++v) win[u][v] = win[u + 2][v];
            }
            if (sg + 1 < nseg && ms + BLUR_QV < H) load_segment(ms + BLUR_QV);
        }
        if (WINO == 4 && NG == 1 && IL && col_tiles == 2 && lt < 64) {
            const int row = lt >> 3, ch = lt & 7;
            const int oy = 2 * ms + row, ox = ct == 0 ? 2 * QC : 2 * QC - 1;      // the neighbour half's first / last pixel
            float v = 0.f;
            if (oy < 2 * H && valid) {
                const float* tq = t + ((int64_t)b * C + g * 8 + ch) * plane_t;
                float o = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int r = oy - 1 + ky, c2 = ox - 1 + kx;           // (columns 126 .. 130 of 0 .. 2W+1: always inside)
                        const float tv = r >= 0 ? tq[((int64_t)(r >> 1) * GW + (c2 >> 1)) * 4 + (c2 & 1) * 2 + (r & 1)] : 0.f;
                        o = fmaf(tv, kf[ky * 4 + kx], o);
                    }
                const float nzv = noise ? noise[(int64_t)b * noise_bstride + (int64_t)oy * OW + ox] : 0.f;
                v = fmaf(nw, nzv, o + (bias ? bias[g * 8 + ch] : 0.f));
                if (act) v = lrelu_gain(v, slope, gain);
            }
            edge_all[row][ch] = v;
        }
        __syncthreads();
        if (WINO == 4) {
            // 8 rows x QC/2 four-pixel tiles, two threads per tile: each takes four of the eight channels through all six positions
            // (neighbouring lanes write the two 8-byte halves of one 16-byte chunk)
            constexpr int TPR = QC / 2;                        // tiles per row
            const int chalf = lt & 1, item = lt >> 1;
            const int tc = item % TPR, row = item / TPR;
            const int oy = 2 * ms + row;
            const int tcg = ct * TPR + tc;                // tile column of the image (two column tiles: W = 2 QC)
            if (oy < 2 * H && 4 * tcg < OW && valid && (NG == 1 || ms < H)) {
                const int HT = H * W;                     // tiles per channel (2H rows x 2W/4)
                unsigned char* dst = xs + ((((int64_t)b * G + g) * 12) * HT + (int64_t)oy * (W / 2) + tcg) * 16 + 8 * chalf;
                float v[6][4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    float d[6], vv[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int px = 4 * tc - 1 + j, pxg = ct * 2 * QC + px;      // inside the block's tile / of the image
                        float val = 0.f;
                        if (pxg >= 0 && pxg < OW)
                            val = (px >= 0 && px < 2 * QC) ? tile[row][px][4 * chalf + cc]
                                                           : edge_all[(WINO == 4 && NG == 1) ? row : 0][4 * chalf + cc];
                        d[j] = val * sv[4 * chalf + cc];
                    }
                    ws_input_transform<6>(d, vv);
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt) v[tt][cc] = vv[tt];
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    unsigned h01, l01, h23, l23;
                    blur_split2<ET>(v[tt][0], v[tt][1], h01, l01, sat);
                    blur_split2<ET>(v[tt][2], v[tt][3], h23, l23, sat);
                    *reinterpret_cast<uint2*>(dst + (int64_t)(2 * tt) * HT * 16) = make_uint2(h01, h23);
                    // (fp8 cross terms: this thread's half of the lo chunk = (4 x lo | 4 x hi) in e4m3, wsplit_common.h)
                    *reinterpret_cast<uint2*>(dst + (int64_t)(2 * tt + 1) * HT * 16) =
                        ET == SGDFR_SPLIT_FP16F8 ? ws_f8_half(h01, h23, l01, l23, f8_mul_lo, f8_mul_hi, false, sat) : make_uint2(l01, l23);
                }
            }
        } else if (WINO == 2) {
            // 8 rows x QC output pairs -> 8 channels each: d_j = y[2*tc - 1 + j] * s_next, V = B^T d, split, 4 x 2 chunks
            const int tc = lt % QC, row = lt / QC;
            const int oy = 2 * ms + row;
            if (oy < 2 * H && 2 * tc < OW && valid && (NG == 1 || ms < H)) {
                const int HT = 2 * H * W;                 // output pairs per channel
                unsigned char* dst = xs + (((((int64_t)b * G + g) * 4) * 2) * HT + (int64_t)oy * W + tc) * 16;
                // one transform position at a time (V1 and V2 share d1, d2; then d0, then d3), so at most two pixels of the
                // tile are live beside the sliding window: with all four at once the window spilled (112 B of scratch per lane)
                auto emit = [&](int tt, const float (&vv)[8]) {
                    uint4 vh, vl;
                    unsigned* ph = reinterpret_cast<unsigned*>(&vh);
                    unsigned* pl = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) blur_split2<ET>(vv[2 * cc], vv[2 * cc + 1], ph[cc], pl[cc], sat);
                    *reinterpret_cast<uint4*>(dst + (int64_t)(2 * tt) * HT * 16) = vh;
                    *reinterpret_cast<uint4*>(dst + (int64_t)(2 * tt + 1) * HT * 16) = vl;
                };
                float d1[8], d2[8], vv[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    d1[cc] = tile[row][2 * tc][cc] * sv[cc];
                    d2[cc] = tile[row][2 * tc + 1][cc] * sv[cc];
                }
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) vv[cc] = d1[cc] + d2[cc];
                emit(1, vv);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) vv[cc] = d2[cc] - d1[cc];
                emit(2, vv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) vv[cc] = (tc > 0 ? tile[row][2 * tc - 1][cc] * sv[cc] : 0.f) - d2[cc];
                emit(0, vv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) vv[cc] = d1[cc] - (2 * tc + 2 < OW ? tile[row][2 * tc + 2][cc] * sv[cc] : 0.f);
                emit(3, vv);
            }
        } else {
        // 8 rows x 64 px pixels -> 8 channels each: multiply by the next layer's style, split, two 16-byte chunks
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = lt + 8 * QC * k;
            const int px = item % (2 * QC), row = item / (2 * QC);
            const int oy = 2 * ms + row, ox = ct * 2 * QC + px;
            if (oy < 2 * H && ox < OW && valid && (NG == 1 || ms < H)) {
                uint4 vh, vl;
                unsigned* ph = reinterpret_cast<unsigned*>(&vh);
                unsigned* pl = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    blur_split2<ET>(tile[row][px][2 * cc] * sv[2 * cc], tile[row][px][2 * cc + 1] * sv[2 * cc + 1], ph[cc], pl[cc], sat);
                if (ET == SGDFR_SPLIT_FP16F8) ws_f8_lo_chunk(vh, vl, f8_mul_lo, f8_mul_hi, false, sat);      // (fp8 cross-term operands, common.h)
                unsigned char* dst = xs + ((((int64_t)b * G + g) * 2) * OHW + (int64_t)oy * OW + ox) * 16;
#if defined(SGDFR_BLUR_PROBE) && SGDFR_BLUR_PROBE == 1      // ablation: no stores (unless a value is NaN: keeps the work alive)
                if (vh.x == 0x7fc07fc0u)
#endif
                {
#if defined(SGDFR_BLUR_PROBE) && (SGDFR_BLUR_PROBE == 5 || SGDFR_BLUR_PROBE == 6)      // A/B: non-temporal hand-over stores
                    typedef unsigned bl_u4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((bl_u4){vh.x, vh.y, vh.z, vh.w}, reinterpret_cast<bl_u4*>(dst));
                    __builtin_nontemporal_store((bl_u4){vl.x, vl.y, vl.z, vl.w}, reinterpret_cast<bl_u4*>(dst + (int64_t)OHW * 16));
#else
                    *reinterpret_cast<uint4*>(dst) = vh;
                    *reinterpret_cast<uint4*>(dst + (int64_t)OHW * 16) = vl;
#endif
                }
            }
        }
        }
        __syncthreads();
        }       // segments
    }
    if ((ET == SGDFR_SPLIT_FP16 || ET == SGDFR_SPLIT_FP16F8) && sat != 0) atomicAdd(sat_word ? sat_word : &g_blur_saturated, sat);
}



// Adjoint of the blur feeding the split-kernel backward of the transposed conv (autograd of Blur, model.py:72-88, consumers
// trainer.py:188): g [B,C,2H,2W] = dL/d(blur output) -> the gradient of the four parity planes gT[2a+py, 2b+px] =
// sum_{dy,dx} g[2a+py+1-dy, 2b+px+1-dx] K[3-dy][3-dx], written ONLY in the phase-major split form of gT*d that
// split_mfma_kernel<DOWN3> stages ([B][(ph*C+c)/8][hi,lo][(H+1)(W+1)][8]); asum[b,c] += sum gT*T (the demodulation
// gradient) when the forward planes t are given.  Block = 8 channels x QC super-pixel columns x BLUR_QV rows, same
// sliding 5x5 window as backward.hip's blur_adjoint_kernel; the fp32 results meet in LDS as [phase][row][column][8
// channels] and leave as whole 16-byte chunks.  When W is a multiple of QC, the planes' extra column W (only its even
// phases are real) is computed by the thread of column W-1 from the same window, so no tile is nearly empty.
template <int ET, int QC>
__global__ __launch_bounds__(8 * QC) void blur_adjoint_split_kernel(const float* __restrict__ g, const float* __restrict__ fir,
                                                                const float* __restrict__ t, const float* __restrict__ d,
                                                                unsigned char* __restrict__ xs, float* __restrict__ asum,
                                                                int B, int C, int H, int W, unsigned* __restrict__ sat_word) {
    __shared__ float tile[4][BLUR_QV][QC + 1][9];
    __shared__ float sv[8];
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = fir[i];
    const int GH = H + 1, GW = W + 1, OH = 2 * H, OW = 2 * W, RP = GH * GW;
    const int G = C / 8;
    const bool extra = (W % QC) == 0;                                  // column W rides on the thread of column W-1
    const int col_tiles = extra ? W / QC : (GW + QC - 1) / QC, row_tiles = (GH + BLUR_QV - 1) / BLUR_QV;
    const int64_t n_tiles = (int64_t)B * G * row_tiles * col_tiles;
    const float xsc = (ET == SGDFR_SPLIT_FP16) ? 0.0625f : 1.f;
    const int c8 = threadIdx.x / QC, nx = threadIdx.x % QC;
    unsigned sat = 0;
    for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
        const int ct = (int)(tile_id % col_tiles);
        int64_t r = tile_id / col_tiles;
        const int rt = (int)(r % row_tiles);
        r /= row_tiles;
        const int gi = (int)(r % G);
        const int b = (int)(r / G);
        const int c = gi * 8 + c8;
        const int bb = ct * QC + nx;
        const int as = rt * BLUR_QV;
        const int64_t pl = (int64_t)b * C + c;
        const bool last_ct = ct == col_tiles - 1;
        const int ncols = min(QC, GW - ct * QC) + ((extra && last_ct) ? 1 : 0);     // columns this tile hands over
        const bool two = extra && last_ct && nx == QC - 1;                            // this thread also owns column W
        if (threadIdx.x < 8) sv[threadIdx.x] = (d ? d[(int64_t)b * C + gi * 8 + threadIdx.x] : 1.f) * xsc;
        float acc_a = 0.f;
        if (bb < (extra ? W : GW)) {
            const float* gp = g + pl * (int64_t)OH * OW;
            const float* tp = t ? t + pl * (int64_t)4 * RP : nullptr;
            bool cok[5];
#pragma unroll
            for (int v = 0; v < 5; ++v) cok[v] = (2 * bb - 2 + v) >= 0 && (2 * bb - 2 + v) < OW;
            float win[5][5];
            // window columns 2bb-2 .. 2bb+2 as two aligned float2 and one float (each pair is in range or not as a whole)
            auto load_row = [&](int gr, float (&dst)[5]) {
                const bool rok = gr >= 0 && gr < OH;
                const float* rp = gp + (int64_t)gr * OW + 2 * bb - 2;
                const float2 z2 = make_float2(0.f, 0.f);
                const float2 p01 = (rok && cok[0]) ? *reinterpret_cast<const float2*>(rp) : z2;
                const float2 p23 = (rok && cok[2]) ? *reinterpret_cast<const float2*>(rp + 2) : z2;
                dst[0] = p01.x; dst[1] = p01.y; dst[2] = p23.x; dst[3] = p23.y;
                dst[4] = (rok && cok[4]) ? rp[4] : 0.f;
            };
#pragma unroll
            for (int u = 0; u < 3; ++u) load_row(2 * as - 2 + u, win[u]);
#pragma unroll
            for (int qv = 0; qv < BLUR_QV; ++qv) {
                const int a = as + qv;
                if (a >= GH) break;
                load_row(2 * a + 1, win[3]);
                load_row(2 * a + 2, win[4]);
#pragma unroll
                for (int py = 0; py < 2; ++py) {
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        float v = 0.f;
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 4; ++dx) v = fmaf(win[py + 3 - dy][px + 3 - dx], k[(3 - dy) * 4 + (3 - dx)], v);
                        // rows / cols 2H+1, 2W+1 of the planes are padding (zeros in the forward planes)
                        if (2 * a + py > 2 * H || 2 * bb + px > 2 * W) v = 0.f;
                        tile[py * 2 + px][qv][nx][c8] = v;
                        if (tp) acc_a = fmaf(v, tp[(py * 2 + px) * RP + a * GW + bb], acc_a);
                    }
                    if (two) {          // column W: phase px = 0 reads g columns 2W-2, 2W-1 = window columns 2, 3; px = 1 is padding
                        float v = 0.f;
#pragma unroll
                        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                            for (int dx = 2; dx < 4; ++dx) v = fmaf(win[py + 3 - dy][5 - dx], k[(3 - dy) * 4 + (3 - dx)], v);
                        if (2 * a + py > 2 * H) v = 0.f;
                        tile[py * 2][qv][QC][c8] = v;
                        tile[py * 2 + 1][qv][QC][c8] = 0.f;
                        if (tp) acc_a = fmaf(v, tp[(py * 2) * RP + a * GW + W], acc_a);
                    }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int v = 0; v < 5; ++v) win[u][v] = win[u + 2][v];
            }
        }
        if (t) {            // the QC threads of a channel are QC consecutive lanes
#pragma unroll
            for (int off = QC / 2; off > 0; off >>= 1) acc_a += __shfl_xor(acc_a, off, 64);
            if (nx == 0) atomicAdd(&asum[pl], acc_a);
        }
        __syncthreads();
        // (phase, row, column) items -> 8 channels each: times d, split, two 16-byte chunks
        const int rows = min(BLUR_QV, GH - as);
        const int n_items = 4 * rows * ncols;
        for (int item = threadIdx.x; item < n_items; item += 8 * QC) {
            const int col = item % ncols;
            const int rq = item / ncols;
            const int qv = rq % rows, ph = rq / rows;
            uint4 vh, vl;
            unsigned* phv = reinterpret_cast<unsigned*>(&vh);
            unsigned* plv = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                blur_split2<ET>(tile[ph][qv][col][2 * cc] * sv[2 * cc], tile[ph][qv][col][2 * cc + 1] * sv[2 * cc + 1], phv[cc], plv[cc], sat);
            unsigned char* dst = xs + ((((int64_t)b * 4 * G + ph * G + gi) * 2) * RP + (int64_t)(as + qv) * GW + ct * QC + col) * 16;
            *reinterpret_cast<uint4*>(dst) = vh;
            *reinterpret_cast<uint4*>(dst + (int64_t)RP * 16) = vl;
        }
        __syncthreads();
    }
    if (ET == SGDFR_SPLIT_FP16 && sat != 0) atomicAdd(sat_word ? sat_word : &g_blur_saturated, sat);
}

}  // namespace sgdfr

using namespace sgdfr;

unsigned int blur_split_saturation_count(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_blur_saturated), sizeof(v)) != hipSuccess) return 0;
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_blur_saturated), &z, sizeof(z));
    }
    return v;
}

extern "C" int sgdfr_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                                   int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                   int pad_y0, int pad_y1, void* stream) {
    SGDFR_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0, "upfirdn2d: bad input shape [%d,%d,%d,%d]", major,
                  in_h, in_w, minor);
    SGDFR_REQUIRE(kh > 0 && kw > 0 && kh <= kMaxTaps && kw <= kMaxTaps, "upfirdn2d: kernel %dx%d unsupported (max %d)",
                  kh, kw, kMaxTaps);
    SGDFR_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "upfirdn2d: up/down factors must be positive");
    const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    SGDFR_REQUIRE(out_h > 0 && out_w > 0, "upfirdn2d: empty output %dx%d", out_h, out_w);
    if (major == 0) return 0;
    SGDFR_REQUIRE(x && k && y, "upfirdn2d: null pointer");
    UpfirdnParams p{x, k, y, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w};
    const int64_t total = (int64_t)major * out_h * out_w * minor;
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), p);
    return check_launch("upfirdn2d");
}

extern "C" int sgdfr_upfirdn2d(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                               int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype, void* stream) {
    if (dtype == SGDFR_DTYPE_F32)
        return sgdfr_upfirdn2d_f32(static_cast<const float*>(x), static_cast<const float*>(k), static_cast<float*>(y), major, in_h, in_w,
                                   minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, stream);
    SGDFR_REQUIRE(dtype == SGDFR_DTYPE_F16 || dtype == SGDFR_DTYPE_F64, "upfirdn2d: dtype must be SGDFR_DTYPE_F32/F16/F64, got %d", dtype);
    SGDFR_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0, "upfirdn2d: bad input shape [%d,%d,%d,%d]", major, in_h, in_w, minor);
    SGDFR_REQUIRE(kh > 0 && kw > 0 && kh <= kMaxTaps && kw <= kMaxTaps, "upfirdn2d: kernel %dx%d unsupported (max %d)", kh, kw, kMaxTaps);
    SGDFR_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "upfirdn2d: up/down factors must be positive");
    const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    SGDFR_REQUIRE(out_h > 0 && out_w > 0, "upfirdn2d: empty output %dx%d", out_h, out_w);
    if (major == 0) return 0;
    SGDFR_REQUIRE(x && k && y, "upfirdn2d: null pointer");
    UpfirdnParams p{nullptr, nullptr, nullptr, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w};
    const int64_t total = (int64_t)major * out_h * out_w * minor;
    int64_t g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == SGDFR_DTYPE_F16)
        hipLaunchKernelGGL((upfirdn2d_typed_kernel<_Float16, float>), dim3((int)g), dim3(256), 0, as_stream(stream),
                           static_cast<const _Float16*>(x), static_cast<const _Float16*>(k), static_cast<_Float16*>(y), p);
    else
        hipLaunchKernelGGL((upfirdn2d_typed_kernel<double, double>), dim3((int)g), dim3(256), 0, as_stream(stream),
                           static_cast<const double*>(x), static_cast<const double*>(k), static_cast<double*>(y), p);
    return check_launch("upfirdn2d");
}

extern "C" int sgdfr_blur_bias_act_f32(const float* t, const float* fir, const float* noise, int64_t noise_bstride,
                                       const float* noise_w, const float* bias, float* y, int B, int C, int H, int W,
                                       int act, float slope, float gain, unsigned int* y_absmax, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, "blur_bias_act: bad shape %d %d %d %d", B, C, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(t && fir && y, "blur_bias_act: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "blur_bias_act: noise without noise_w");
    const int64_t strips = (int64_t)B * C * ((H + BLUR_QV - 1) / BLUR_QV) * W;
    int64_t g = (strips + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    // (strips of one plane: HS * W; a wave stays inside a plane and is complete when that is a multiple of 64 -- the grid stride,
    //  a multiple of 256, then keeps it so)
    const int wave_planes = ((int64_t)((H + BLUR_QV - 1) / BLUR_QV) * W) % 64 == 0 ? 1 : 0;
    hipLaunchKernelGGL(blur_bias_act_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), t, fir, noise,
                       noise_bstride, noise_w, bias, y, B, C, H, W, act, slope, gain, y_absmax, wave_planes);
    return check_launch("blur_bias_act");
}

extern "C" int sgdfr_blur_bias_act_split_f32(const float* t, const float* fir, const float* noise, int64_t noise_bstride,
                                             const float* noise_w, const float* bias, const float* s_next, unsigned short* xs,
                                             int B, int C, int H, int W, int64_t plane_stride, int arith, int wino, int act,
                                             float slope, float gain, unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "blur_bias_act_split: bad shape %d %d %d %d (C %% 8)", B, C, H, W);
    SGDFR_REQUIRE(wino == 0 || wino == 2 || wino == 4, "blur_bias_act_split: wino is 0, 2 or 4 (outputs per Winograd tile), got %d", wino);
    SGDFR_REQUIRE(!wino || (W >= 8 && W <= 64 && (W & (W - 1)) == 0) || (wino == 4 && W == 128 && plane_stride != 0),
                  "blur_bias_act_split: the Winograd hand-over takes W = 8, 16, 32 or 64 (output rows inside one column tile; F(4,3) on "
                  "interleaved planes also W = 128), got %d", W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || (arith == SGDFR_SPLIT_FP16F8 && wino != 2),
                  "blur_bias_act_split: arith must be SGDFR_SPLIT_BF16/FP16 (or FP16F8 with the plain or the F(4,3) hand-over)");
    if (B == 0) return 0;
    SGDFR_REQUIRE(t && fir && s_next && xs && (reinterpret_cast<uintptr_t>(xs) & 15) == 0, "blur_bias_act_split: null / misaligned pointer");
    SGDFR_REQUIRE(!noise || noise_w, "blur_bias_act_split: noise without noise_w");
    static const int il_env = getenv("SGDFR_PLANE_IL") ? atoi(getenv("SGDFR_PLANE_IL")) : 1;      // (0: padded PLANAR planes, A/B only)
    const bool il = plane_stride != 0 && il_env != 0;       // padded planes are interleaved planes (see sgdfr.h)
    SGDFR_REQUIRE(!(wino && W == 128) || il, "blur_bias_act_split: the Winograd hand-over of W = 128 needs interleaved planes");
    if (plane_stride == 0) plane_stride = (int64_t)(H + 1) * (W + 1);
    SGDFR_REQUIRE(plane_stride >= (int64_t)(H + 1) * (W + 1) && plane_stride < (1 << 30), "blur_bias_act_split: plane_stride < (H+1)*(W+1)");
    SGDFR_REQUIRE(!il || (reinterpret_cast<uintptr_t>(t) & 15) == 0, "blur_bias_act_split: interleaved planes must be 16-byte aligned");
    const int QC = W >= 64 ? 64 : W > 16 ? 32 : W > 8 ? 16 : W > 4 ? 8 : 4;
    const int NG = QC < 32 ? 256 / (8 * QC) : 1;
    // row segments per block: as many as keep >= 8 blocks per CU (the window slides across them: each plane row is read once)
    const int seg_env = getenv("SGDFR_BLUR_SEGMENTS") ? atoi(getenv("SGDFR_BLUR_SEGMENTS")) : 0;
    const int64_t tiles1 = (int64_t)B * (C / 8) * ((H + BLUR_QV - 1) / BLUR_QV) * ((W + QC - 1) / QC);
    int nseg = 1;
    while (nseg < 8 && BLUR_QV * nseg * 2 <= H && tiles1 / (nseg * 2) >= 256 * 8) nseg *= 2;
    if (seg_env > 0) nseg = seg_env;
    const int64_t groups = (int64_t)B * (C / 8) * ((H + BLUR_QV * nseg - 1) / (BLUR_QV * nseg));
    const int col_tiles = (W + QC - 1) / QC;
    const int64_t tiles = col_tiles == 2 ? ((groups + 7) / 8) * 16 : ((groups + NG - 1) / NG) * col_tiles;       // see `paired` / NG in the kernel
    int64_t g = tiles;
    if (g > 256 * 32) g = 256 * 32;           // (a multiple of 16: the grid-stride loop keeps the pairing)
    unsigned char* out = reinterpret_cast<unsigned char*>(xs);
    void (*kern)(const float*, const float*, const float*, int64_t, const float*, const float*, const float*, unsigned char*, int,
                 int, int, int, int, int, int, float, float, unsigned*);
    // (ET, WINO, IL) -> the QC / NG variant of this launch
#define SGDFR_BLUR_PICK(ET_, WINO_, IL_)                                                                                          \
    (QC == 64   ? blur_split_kernel<ET_, 64, 1, WINO_, IL_>                                                                       \
     : QC == 32 ? blur_split_kernel<ET_, 32, 1, WINO_, IL_>                                                                       \
     : QC == 16 ? blur_split_kernel<ET_, 16, 2, WINO_, IL_>                                                                       \
     : (QC == 8 || WINO_ != 0) ? blur_split_kernel<ET_, 8, 4, WINO_, IL_>                                                         \
                : blur_split_kernel<ET_, 4, (WINO_ != 0 ? 4 : 8), WINO_, IL_>)
#define SGDFR_BLUR_PICK_W(ET_, IL_) (wino == 2 ? SGDFR_BLUR_PICK(ET_, 2, IL_) : wino == 4 ? SGDFR_BLUR_PICK(ET_, 4, IL_) : SGDFR_BLUR_PICK(ET_, 0, IL_))
    if (arith == SGDFR_SPLIT_FP16F8)
        kern = wino == 4 ? (il ? SGDFR_BLUR_PICK(SGDFR_SPLIT_FP16F8, 4, true) : SGDFR_BLUR_PICK(SGDFR_SPLIT_FP16F8, 4, false))
                         : (il ? SGDFR_BLUR_PICK(SGDFR_SPLIT_FP16F8, 0, true) : SGDFR_BLUR_PICK(SGDFR_SPLIT_FP16F8, 0, false));
    else if (arith == SGDFR_SPLIT_FP16) kern = il ? SGDFR_BLUR_PICK_W(SGDFR_SPLIT_FP16, true) : SGDFR_BLUR_PICK_W(SGDFR_SPLIT_FP16, false);
    else kern = il ? SGDFR_BLUR_PICK_W(SGDFR_SPLIT_BF16, true) : SGDFR_BLUR_PICK_W(SGDFR_SPLIT_BF16, false);
#undef SGDFR_BLUR_PICK_W
#undef SGDFR_BLUR_PICK
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(8 * QC * NG), 0, as_stream(stream), t, fir, noise, noise_bstride, noise_w, bias, s_next,
                       out, B, C, H, W, (int)plane_stride, nseg, act, slope, gain, sat);
    return check_launch("blur_bias_act_split");
}

extern "C" int sgdfr_blur_adjoint_split_f32(const float* g, const float* fir, const float* t, const float* d, unsigned short* xs,
                                            float* asum, int B, int C, int H, int W, int arith, unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "blur_adjoint_split: bad shape %d %d %d %d (C %% 8)", B, C, H, W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "blur_adjoint_split: arith must be SGDFR_SPLIT_BF16/FP16");
    if (B == 0) return 0;
    SGDFR_REQUIRE(g && fir && xs && (reinterpret_cast<uintptr_t>(xs) & 15) == 0, "blur_adjoint_split: null / misaligned pointer");
    SGDFR_REQUIRE(!t || asum, "blur_adjoint_split: t given without asum");
    hipStream_t st = as_stream(stream);
    if (t && hipMemsetAsync(asum, 0, sizeof(float) * (size_t)B * C, st) != hipSuccess) return check_launch("memset");
    const int QC = W >= 64 ? 64 : 32;
    const int col_tiles = (W % QC) == 0 ? W / QC : (W + 1 + QC - 1) / QC;
    const int64_t tiles = (int64_t)B * (C / 8) * ((H + 1 + BLUR_QV - 1) / BLUR_QV) * col_tiles;
    int64_t grid = tiles;
    if (grid > 256 * 32) grid = 256 * 32;
    unsigned char* out = reinterpret_cast<unsigned char*>(xs);
    void (*kern)(const float*, const float*, const float*, const float*, unsigned char*, float*, int, int, int, int, unsigned*);
    if (arith == SGDFR_SPLIT_FP16) kern = QC == 64 ? blur_adjoint_split_kernel<SGDFR_SPLIT_FP16, 64> : blur_adjoint_split_kernel<SGDFR_SPLIT_FP16, 32>;
    else kern = QC == 64 ? blur_adjoint_split_kernel<SGDFR_SPLIT_BF16, 64> : blur_adjoint_split_kernel<SGDFR_SPLIT_BF16, 32>;
    hipLaunchKernelGGL(kern, dim3((int)grid), dim3(8 * QC), 0, st, g, fir, t, d, out, asum, B, C, H, W, sat);
    return check_launch("blur_adjoint_split");
}
