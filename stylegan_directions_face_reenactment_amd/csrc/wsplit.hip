// Plain 3x3 modulated convolution in 1-D Winograd F(2,3) form on the split-operand 16-bit matrix cores (the arithmetic of
// split.hip: every fp32 operand as two 16-bit terms hi + lo, three MFMA products, fp32 accumulation).
//
// Why: the split conv kernels run at the chip's power budget inside their K loop -- the only way past that roof is
// fewer MFMAs per output.  Along image rows the 3-tap correlation of two neighbouring outputs is computed from four
// transformed inputs (F(2,3): 4 multiplies instead of 6); kernel rows stay direct.  Per 16 input channels and output PAIR that
// is 3 ky x 4 positions = 12 MFMA columns instead of 9 x 2 = 18: 2/3 of the MFMA work.  scripts/mfma16_probe.hip prices it on
// the device with random operands, LDS-fed fragments and the kernel's DMA rates: 586 fp32-equivalent TFLOP/s for this kernel's
// loop against 430 for the plain 128 x 512 tile -- the 1.5x survives as 1.36x because the transformed operands are twice the
// bytes (four values per two pixels) and the transformed weights 4/3.
//
//   V_t[tile]   = B^T d,  d_j = (x*s)[2*tile - 1 + j]:  V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3   (producer)
//   U_t[ky]     = G g,    g = W[ky][0..2]:              U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2      (pack)
//   M_t[co,tile] = sum_ky sum_ci U_t[ky][co,ci] * V_t[ci, tile + (ky-1) * row pitch]                               (MFMA)
//   y[2*tile] = M0 + M1 + M2,  y[2*tile+1] = M1 - M2 - M3                                                          (epilogue)
//
// The transforms are applied to fp32 values BEFORE the hi/lo split (in the producer of the activation: sgdfr_to_wsplit_f32, or
// the blur kernel's hand-over; in the weight pack), so every MFMA operand keeps its 22 (fp16) / 16 (bf16) bits.  |V| <= 2 max|x*s|:
// the range plan of the fp16 terms leaves one more binade for Winograd consumers.
//
// Data path: "WS" input [B][Cin/8][t 4][hi,lo][H * W/2][8 x 16 bit] (8 bytes per input element), staged by global->LDS DMA only
// (no registers, no VALU); weight pack [cout tile 128][cin block][ky][t][hi,lo][k-half][128][8].  Block = 8 waves = 128 couts x
// 128 tiles (256 pixels) as a TR x TCT patch of tile space (TCT = min(16, W/2) tile columns: Winograd tiles do not overlap
// along x, so a patch has only a vertical halo: (TR+2) x TCT staged positions); wave tile 32 couts x 64 tiles x 4 positions = 8
// accumulators.  K loop: 16-channel blocks x 6 barrier-delimited half-stages (kernel row x position pair: 12 MFMAs per wave),
// V double-buffered per channel block (40 KB each), U in a 4-slot ring of 16 KB half-slabs DMA'd three half-stages ahead; the
// first fragments of a half-stage are requested before the barrier that precedes it (see the loop).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sgdfr {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ws_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ws_f16x2 __attribute__((ext_vector_type(2)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef int ws_frag __attribute__((ext_vector_type(4)));

__device__ unsigned int g_wsplit_saturated = 0;     // clamped operand pairs of launches without a saturation word

constexpr float WS_F16_XSCALE = 0.0625f, WS_F16_WSCALE = 64.f, WS_F16_OUT = 0.25f, WS_F16_MAX = 65504.f;     // as split.hip
constexpr int WS_CB = 16;

template <int ET>
__device__ __forceinline__ ws_f32x16 ws_mfma(ws_frag a, ws_frag b, ws_f32x16 c) {
    if (ET == SGDFR_SPLIT_FP16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ws_f16x8, a), __builtin_bit_cast(ws_f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ws_bf16x8, a), __builtin_bit_cast(ws_bf16x8, b), c, 0, 0, 0);
}

// two floats -> packed hi pair, packed lo pair (split.hip's split_pair: same rounding, same clamp-and-count rule)
template <int ET>
__device__ __forceinline__ void ws_pair(float a, float b, unsigned& hi, unsigned& lo, unsigned& sat) {
    if (ET == SGDFR_SPLIT_FP16) {
        sat += (!(fabsf(a) <= WS_F16_MAX) || !(fabsf(b) <= WS_F16_MAX)) ? 1u : 0u;
        a = __builtin_amdgcn_fmed3f(a, -WS_F16_MAX, WS_F16_MAX);
        b = __builtin_amdgcn_fmed3f(b, -WS_F16_MAX, WS_F16_MAX);
        const ws_f16x2 h = __builtin_convertvector((ws_f32x2){a, b}, ws_f16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const ws_f32x2 hf = __builtin_convertvector(h, ws_f32x2);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a - hf[0], b - hf[1]}, ws_f16x2));
    } else {
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a, b}, ws_bf16x2));
        const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector((ws_f32x2){a - ha, b - hb}, ws_bf16x2));
    }
}

struct WsParams {
    const unsigned char* v;      // WS input
    const unsigned char* wsp;    // U pack
    const unsigned char* zeros;  // >= 16 zero bytes
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    float* y;
    const float* rgb_w;          // fused ToRGB, as split.hip: [3][Cout] weights, [B][Cout] styles, partial sums [B][T*3][H*W]
    const float* rgb_s;
    float* rgb_part;
    unsigned char* xs_out;       // the activation in the next (transposed) conv's split input form [B][Cout/8][hi,lo][H*W][8]
    const float* s_next;
    unsigned* sat;
    int B, Cin, Cout, H, W;
    int TW;                      // tiles per image row (W / 2)
    int TCT, TR, tct_shift;      // patch: TR rows x TCT tile columns (TR * TCT = 128)
    int tiles_x, tiles_y;
    int xs;                      // staged positions per (t, part, k-half) run: (TR + 2) * TCT
    int n_pix_tiles, n_cout_tiles;
    int act;
    float slope, gain;
    int dbg;
    FastDiv fd_xs, fd_tiles_x, fd_per_img, fd_npt;
};

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in 0..4 (the instruction takes an immediate)
__device__ __forceinline__ void ws_wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: ws_wait_vmcnt<0>(); break;
        case 1: ws_wait_vmcnt<1>(); break;
        case 2: ws_wait_vmcnt<2>(); break;
        case 3: ws_wait_vmcnt<3>(); break;
        default: ws_wait_vmcnt<4>(); break;
    }
}

template <int ET>
__global__ __launch_bounds__(512, 1) void wsplit_kernel(WsParams p) {
    constexpr int NT = 128, NI = 2, NEXV = 5;
    constexpr int WSLOT = 32768;                  // one kernel row of one cout tile: [t 4][part 2][k-half 2][128][8] x 16 bit
    constexpr int WHALF = 16384;                  // one position pair of it = one half-stage = one ring slot
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xbuf_bytes = 256 * p.xs;            // [t 4][part 2][k-half 2][xs][8] x 16 bit
    unsigned char* const xb0 = smem;
    unsigned char* const wb0 = smem + 2 * xbuf_bytes;
    float* const dl = reinterpret_cast<float*>(wb0 + 2 * WSLOT);     // [NT] d * output scale
    float* const bl = dl + NT;                                        // [NT] bias
    float* const sn = bl + NT;                                        // [NT] next layer's style * range shift
    float* const cw = sn + NT;                                        // [NT][4] ToRGB coefficients
    float* const red = reinterpret_cast<float*>(smem);                // [4][256 px][3] after the K loop (dead staging buffers)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W, HT = p.H * p.TW, G8 = p.Cin / 8;

    // XCD-aware tile order: the blocks resident on one XCD (blockIdx & 7) walk neighbouring patches of one cout tile
    int lid;
    {
        const int nblk = (int)gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ct = fdiv(lid, p.fd_npt), pt = lid - ct * p.n_pix_tiles;
    const int per_img = p.tiles_x * p.tiles_y;
    const int img0 = fdiv(pt, p.fd_per_img);
    const int prem = pt - img0 * per_img;
    const int ty = fdiv(prem, p.fd_tiles_x), tx = prem - ty * p.tiles_x;
    const int row0 = ty * p.TR, col0 = tx * p.TCT, n0 = ct * NT;

    const bool fuse_rgb = p.rgb_part != nullptr, emit_xs = p.xs_out != nullptr;
    // epilogue coefficients: global loads at the top of the tile, LDS writes in the prologue (one exposed latency, hidden
    // behind the descriptor arithmetic); they go through LDS because loads between stores would serialise on vmcnt
    float t_d = 1.f, t_b = 0.f, t_s = 0.f, t_r = 0.f, t_w0 = 0.f, t_w1 = 0.f, t_w2 = 0.f;
    if (tid < NT) {
        const int64_t bc = (int64_t)img0 * p.Cout + n0 + tid;
        t_d = p.d ? p.d[bc] : 1.f;
        t_b = p.bias ? p.bias[n0 + tid] : 0.f;
        if (emit_xs) t_s = p.s_next[bc];
        if (fuse_rgb) {
            t_r = p.rgb_s[bc];
            t_w0 = p.rgb_w[n0 + tid];
            t_w1 = p.rgb_w[p.Cout + n0 + tid];
            t_w2 = p.rgb_w[2 * p.Cout + n0 + tid];
        }
    }

    // this lane's two tile columns: position inside the staged patch, output pixel pair, noise
    int boff[NI], pix[NI];
    float nz[NI][2];
    {
        const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int l = (wn * NI + n) * 32 + l31;
            const int r = l >> p.tct_shift, c = l & (p.TCT - 1);
            boff[n] = (hi * p.xs + l) * 16;                      // (staged row r + ky holds image row row0 - 1 + r + ky)
            pix[n] = (row0 + r) * p.W + 2 * (col0 + c);
            nz[n][0] = nz[n][1] = 0.f;
            if (p.noise) {
                const float2 t2 = *reinterpret_cast<const float2*>(p.noise + (int64_t)img0 * p.noise_bstride + pix[n]);
                nz[n][0] = nw * t2.x;
                nz[n][1] = nw * t2.y;
            }
        }
    }

    // staging descriptors of the V patch: item i = (run (t, part, k-half), position) -> one 16-byte chunk; a wave's piece is
    // 64 consecutive items, so the LDS image is simply item order.  -1: zero page (rows outside the image), -2: no item.
    int64_t vsrc[NEXV];
#pragma unroll
    for (int e = 0; e < NEXV; ++e) {
        const int i = (e * 8 + wave) * 64 + lane;
        if ((e * 8 + wave) * 64 >= 16 * p.xs) { vsrc[e] = -2; continue; }
        const int run = fdiv(i, p.fd_xs), pos = i - run * p.xs;
        const int sr = pos >> p.tct_shift, c = pos & (p.TCT - 1);
        const int row = row0 - 1 + sr;
        const int t = run >> 2, part = (run >> 1) & 1, h = run & 1;
        vsrc[e] = (row >= 0 && row < p.H)
                      ? ((((((int64_t)img0 * G8 + h) * 4 + t) * 2 + part) * HT) + (int64_t)row * p.TW + col0 + c) * 16
                      : -1;
    }
    const int64_t v_cb_stride = (int64_t)256 * HT;      // 2 eight-channel groups x [4][2][HT][16 B]

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto issue_v = [&](int e, int cb, unsigned char* xb) {
        if (vsrc[e] == -2) return;                       // wave-uniform
#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 8) return;
#endif
        const unsigned char* src = vsrc[e] >= 0 ? p.v + vsrc[e] + cb * v_cb_stride : p.zeros;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(xb + (e * 8 + wave) * 1024), 16, 0, 0);
    };
    const int ncb = p.Cin / WS_CB;
    const int nh = ncb * 6;                             // half-stages: (channel block, kernel row, position pair)
    const unsigned char* const wglb = p.wsp + (int64_t)ct * ncb * 3 * WSLOT;
    // weight slab of half-stage h = (cb, ky, tp): 16 KB [2 positions][part][k-half][128][8], 2 pieces per wave
    auto issue_w = [&](int h) {
#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 16) return;
#endif
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int piece = wave + v * 8;
            __builtin_amdgcn_global_load_lds((glb_void*)(wglb + (int64_t)h * WHALF + piece * 1024 + lane * 16),
                                             (lds_void*)(wb0 + (h & 3) * WHALF + piece * 1024), 16, 0, 0);
        }
    };

    ws_f32x16 acc[4][NI];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;

    // V pieces this wave really issues (xs = 144: the fifth piece exists for waves 0-3 only), per half-stage of a channel block:
    // pieces {0, 1}, {2}, {3}, {4} in half-stages 0..3 -- all landed and published by the barrier that ends half-stage 4
    int nvw[4];
    {
        auto ex = [&](int e) { return (e * 8 + wave) * 64 < 16 * p.xs ? 1 : 0; };
        nvw[0] = ex(0) + ex(1); nvw[1] = ex(2); nvw[2] = ex(3); nvw[3] = ex(4);
    }

    // ---- prologue: channel block 0, the first three weight half-slabs
#pragma unroll
    for (int e = 0; e < NEXV; ++e) issue_v(e, 0, xb0);
    issue_w(0);
    if (nh > 1) issue_w(1);
    if (nh > 2) issue_w(2);
    if (tid < NT) {
        const float oscale = (ET == SGDFR_SPLIT_FP16) ? WS_F16_OUT : 1.f;
        const float xsc = (ET == SGDFR_SPLIT_FP16) ? WS_F16_XSCALE : 1.f;
        const float rs = rsqrtf((float)p.Cout);
        dl[tid] = t_d * oscale;
        bl[tid] = t_b;
        sn[tid] = t_s * xsc;
        *reinterpret_cast<float4*>(cw + 4 * tid) = make_float4(t_w0 * (t_r * rs), t_w1 * (t_r * rs), t_w2 * (t_r * rs), 0.f);
    }
    ws_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // K loop.  The weight half-slabs live in a FOUR-slot ring (16 KB each) and are DMA'd three half-stages ahead: the slab of
    // h+1 is published by the barrier that ends h-1, so the first fragments of h+1 are requested in the middle of h -- BEFORE
    // the barrier that ends h -- and no wave starts a half-stage waiting for LDS.  (With one barrier per kernel row and two
    // 32 KB slots every wave of the block asked for its first six fragments right behind the barrier: 48 KB of LDS reads with
    // the matrix cores idle, ~15 % of a 24-MFMA sub-stage.)  The barrier's counted wait leaves the pieces issued during h in
    // flight and completes everything older (the slab of h+2, V pieces of the next channel block).
    const int a_off = (hi * 128 + wm * 32 + l31) * 16;
    const bool late = wave >= 4 && !(p.dbg & 4);      // the two waves of a SIMD issue their DMA pieces at different times (split.hip)
    const int rowstep = p.TCT * 16;
    ws_frag a[2][2], b[2][2][NI];      // [set][part]
    auto fetch = [&](int set, const unsigned char* wsl, const unsigned char* xr, int tl, int t) {
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            a[set][part] = *reinterpret_cast<const ws_frag*>(wsl + (tl * 2 + part) * 4096);
#pragma unroll
            for (int n = 0; n < NI; ++n)
                b[set][part][n] = *reinterpret_cast<const ws_frag*>(xr + (t * 2 + part) * 32 * p.xs + boff[n]);
        }
    };
    fetch(0, wb0 + a_off, xb0, 0, 0);
    int xsel = 0, h = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const unsigned char* xcur = xb0 + xsel * xbuf_bytes;
        unsigned char* xnext = xb0 + (xsel ^ 1) * xbuf_bytes;
        const bool v_next = cb + 1 < ncb;
#pragma unroll
        for (int hh = 0; hh < 6; ++hh, ++h) {
            const int ky = hh >> 1, tp = hh & 1;
            int n_issued = 0;
            auto issue_all = [&]() {
                if (h + 3 < nh) { issue_w(h + 3); n_issued += 2; }
                if (v_next && hh < 4) {
                    if (hh == 0) { issue_v(0, cb + 1, xnext); issue_v(1, cb + 1, xnext); }
                    if (hh == 1) issue_v(2, cb + 1, xnext);
                    if (hh == 2) issue_v(3, cb + 1, xnext);
                    if (hh == 3) issue_v(4, cb + 1, xnext);
                    n_issued += nvw[hh < 4 ? hh : 0];
                }
            };
            if (!late) issue_all();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* wsl = wb0 + (h & 3) * WHALF + a_off;
            const unsigned char* xrow = xcur + ky * rowstep;
            // first position of the pair: fragments were requested during the previous half-stage
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp][n] = ws_mfma<ET>(a[0][0], b[0][0][n], acc[2 * tp][n]);
            __builtin_amdgcn_sched_barrier(0);
            fetch(1, wsl, xrow, 1, 2 * tp + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp][n] = ws_mfma<ET>(a[0][0], b[0][1][n], acc[2 * tp][n]);
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp][n] = ws_mfma<ET>(a[0][1], b[0][0][n], acc[2 * tp][n]);
            __builtin_amdgcn_sched_barrier(0);
            // second position; the first fragments of the NEXT half-stage are requested behind its hi*hi products
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp + 1][n] = ws_mfma<ET>(a[1][0], b[1][0][n], acc[2 * tp + 1][n]);
            __builtin_amdgcn_sched_barrier(0);
            if (late) issue_all();
            if (h + 1 < nh) {
                const int hn = hh + 1;      // (6 = the first half-stage of the next channel block)
                fetch(0, wb0 + ((h + 1) & 3) * WHALF + a_off, (hn == 6 ? xnext : xcur) + (hn == 6 ? 0 : (hn >> 1)) * rowstep, 0,
                      hn == 6 ? 0 : 2 * (hn & 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp + 1][n] = ws_mfma<ET>(a[1][0], b[1][1][n], acc[2 * tp + 1][n]);
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[2 * tp + 1][n] = ws_mfma<ET>(a[1][1], b[1][0][n], acc[2 * tp + 1][n]);
            __builtin_amdgcn_sched_barrier(0);
            if (h + 1 < nh) {
#ifdef SGDFR_WSPLIT_PROBE
                if (!(p.dbg & 32))
#endif
                ws_wait_vmcnt_dyn(n_issued);
                __builtin_amdgcn_s_barrier();
            }
        }
        xsel ^= 1;
    }

#ifdef SGDFR_WSPLIT_PROBE
    if (p.dbg & 2) return;
#endif
    // ---- epilogue.  C/D layout of 32x32: column (tile) = lane & 31, row (cout) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const float e_slope = p.act ? p.slope : 1.f, e_gain = p.act ? p.gain : 1.f;
    unsigned sat = 0;
    float rgb[NI][2][3];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int q = 0; q < 2; ++q) rgb[n][q][0] = rgb[n][q][1] = rgb[n][q][2] = 0.f;
    auto epilogue = [&](auto has_y_t, auto emit_xs_t, auto fuse_rgb_t) {
        constexpr bool HAS_Y = decltype(has_y_t)::value, EMIT_XS = decltype(emit_xs_t)::value, FUSE_RGB = decltype(fuse_rgb_t)::value;
        const int io = wm * 32 + 4 * hi;
        const float4* const d4p = reinterpret_cast<const float4*>(dl + io);
        const float4* const b4p = reinterpret_cast<const float4*>(bl + io);
        const float4* const s4p = reinterpret_cast<const float4*>(sn + io);
        const float4* const cwp = reinterpret_cast<const float4*>(cw) + io;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            float* const yp = p.y + ((int64_t)img0 * p.Cout + n0 + io) * HW + pix[n];
            unsigned char* const xp = p.xs_out + ((((int64_t)img0 * (p.Cout / 8) + (n0 + wm * 32) / 8) * 2) * HW + pix[n]) * 16 + 8 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 dq = d4p[2 * g], bq = b4p[2 * g];
                const float dv[4] = {dq.x, dq.y, dq.z, dq.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
                float v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r];
                    const float y0 = (m0 + m1) + m2, y1 = (m1 - m2) - m3;
                    v0[j] = lrelu_gain(y0 * dv[j] + nz[n][0] + bv[j], e_slope, e_gain);
                    v1[j] = lrelu_gain(y1 * dv[j] + nz[n][1] + bv[j], e_slope, e_gain);
                }
                if (HAS_Y) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float2*>(yp + (int64_t)(8 * g + j) * HW) = make_float2(v0[j], v1[j]);
                }
                if (EMIT_XS) {      // the 4 rows are half of one 8-channel chunk of each of the two pixels
                    const float4 sq = s4p[2 * g];
                    unsigned h01, l01, h23, l23;
                    unsigned char* dst = xp + (int64_t)g * 2 * HW * 16;
                    ws_pair<ET>(v0[0] * sq.x, v0[1] * sq.y, h01, l01, sat);
                    ws_pair<ET>(v0[2] * sq.z, v0[3] * sq.w, h23, l23, sat);
                    *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(dst + (int64_t)HW * 16) = make_uint2(l01, l23);
                    ws_pair<ET>(v1[0] * sq.x, v1[1] * sq.y, h01, l01, sat);
                    ws_pair<ET>(v1[2] * sq.z, v1[3] * sq.w, h23, l23, sat);
                    *reinterpret_cast<uint2*>(dst + 16) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(dst + (int64_t)HW * 16 + 16) = make_uint2(l01, l23);
                }
                if (FUSE_RGB) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 q = cwp[8 * g + j];
                        rgb[n][0][0] = fmaf(v0[j], q.x, rgb[n][0][0]);
                        rgb[n][0][1] = fmaf(v0[j], q.y, rgb[n][0][1]);
                        rgb[n][0][2] = fmaf(v0[j], q.z, rgb[n][0][2]);
                        rgb[n][1][0] = fmaf(v1[j], q.x, rgb[n][1][0]);
                        rgb[n][1][1] = fmaf(v1[j], q.y, rgb[n][1][1]);
                        rgb[n][1][2] = fmaf(v1[j], q.z, rgb[n][1][2]);
                    }
                }
            }
        }
    };
    {
        using yes = std::true_type;
        using no = std::false_type;
        switch ((p.y ? 1 : 0) | (emit_xs ? 2 : 0) | (fuse_rgb ? 4 : 0)) {       // block-uniform
            case 1: epilogue(yes{}, no{}, no{}); break;
            case 2: epilogue(no{}, yes{}, no{}); break;
            case 3: epilogue(yes{}, yes{}, no{}); break;
            case 4: epilogue(no{}, no{}, yes{}); break;
            case 5: epilogue(yes{}, no{}, yes{}); break;
            case 6: epilogue(no{}, yes{}, yes{}); break;
            case 7: epilogue(yes{}, yes{}, yes{}); break;
            default: break;
        }
    }
    if (fuse_rgb) {      // block-uniform: the two lane halves hold different couts of the same pixels; the 4 cout waves meet in LDS
        __syncthreads();      // every wave is done with the staging buffers `red` overlays
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float v = rgb[n][q][j] + __shfl_xor(rgb[n][q][j], 32, 64);
                    if (hi == 0) red[(wm * 256 + ((wn * NI + n) * 32 + l31) * 2 + q) * 3 + j] = v;
                }
        __syncthreads();
        if (wm == 0 && hi == 0) {
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float t2[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) t += red[(w2 * 256 + ((wn * NI + n) * 32 + l31) * 2 + q) * 3 + j];
                        t2[q] = t;
                    }
                    *reinterpret_cast<float2*>(p.rgb_part + (((int64_t)img0 * p.n_cout_tiles + ct) * 3 + j) * HW + pix[n]) =
                        make_float2(t2[0], t2[1]);
                }
        }
    }
    if (ET == SGDFR_SPLIT_FP16 && __builtin_expect(sat != 0, 0)) atomicAdd(p.sat ? p.sat : &g_wsplit_saturated, sat);
}

// x [B,Cin,H,W] fp32 and s [B,Cin] -> WS [B][Cin/8][t 4][hi,lo][H * W/2][8]: B^T (x*s) per row pair, split.
// One thread = one tile of one 8-channel group.
template <int ET>
__global__ __launch_bounds__(256) void to_wsplit_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       unsigned char* __restrict__ vs, int B, int Cin, int H, int W,
                                                       unsigned* __restrict__ sat_word) {
    const int G = Cin / 8, TW = W / 2, HT = H * TW;
    const int64_t n = (int64_t)B * G * HT;
    const float sc = (ET == SGDFR_SPLIT_FP16) ? WS_F16_XSCALE : 1.f;
    unsigned sat = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int pos = (int)(idx % HT);
        const int64_t bg = idx / HT;
        const int g = (int)(bg % G), b = (int)(bg / G);
        const int row = pos / TW, tc = pos - row * TW;
        float v[4][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* xp = x + (((int64_t)b * Cin + g * 8 + c) * H + row) * W + 2 * tc;
            const float sv = s[(int64_t)b * Cin + g * 8 + c] * sc;
            const float d0 = tc > 0 ? xp[-1] * sv : 0.f, d1 = xp[0] * sv, d2 = xp[1] * sv, d3 = tc + 1 < TW ? xp[2] * sv : 0.f;
            v[0][c] = d0 - d2;
            v[1][c] = d1 + d2;
            v[2][c] = d2 - d1;
            v[3][c] = d1 - d3;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 vh, vl;
            unsigned* ph = reinterpret_cast<unsigned*>(&vh);
            unsigned* pl = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
            for (int c = 0; c < 4; ++c) ws_pair<ET>(v[t][2 * c], v[t][2 * c + 1], ph[c], pl[c], sat);
            unsigned char* dst = vs + (((bg * 4 + t) * 2) * HT + pos) * 16;
            *reinterpret_cast<uint4*>(dst) = vh;
            *reinterpret_cast<uint4*>(dst + (int64_t)HT * 16) = vl;
        }
    }
    if (ET == SGDFR_SPLIT_FP16 && sat != 0) atomicAdd(sat_word ? sat_word : &g_wsplit_saturated, sat);
}

// weight [Cout,Cin,3,3] fp32 -> 16-bit hi/lo of U = G (weight/sqrt(9 Cin)) per kernel row, in the kernel's LDS order:
//   [cout tile 128][cin block][ky][t][part][k-half][128 couts][8 cin]
__global__ __launch_bounds__(256) void prepack_wsplit_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                            int Cout, int Cin, float scale, int et, unsigned* __restrict__ sat_word) {
    const int64_t n = (int64_t)Cout * Cin * 3;
    const int ncb = Cin / WS_CB;
    unsigned sat = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ky = (int)(idx % 3);
        const int ci = (int)((idx / 3) % Cin);
        const int co = (int)(idx / (3 * (int64_t)Cin));
        const float* g = w + ((int64_t)co * Cin + ci) * 9 + ky * 3;
        const float g0 = g[0] * scale, g1 = g[1] * scale, g2 = g[2] * scale;
        const float U[4] = {g0, 0.5f * ((g0 + g2) + g1), 0.5f * ((g0 + g2) - g1), g2};
        const int ctile = co / 128, col = co - ctile * 128, cb = ci / WS_CB, h = (ci % WS_CB) / 8, c8 = ci % 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned hp, lp;
            if (et == SGDFR_SPLIT_FP16) ws_pair<SGDFR_SPLIT_FP16>(U[t], 0.f, hp, lp, sat);
            else ws_pair<SGDFR_SPLIT_BF16>(U[t], 0.f, hp, lp, sat);
            const int64_t base = ((((int64_t)ctile * ncb + cb) * 3 + ky) * 4 + t) * 2;      // -> [part]
            out[(((base + 0) * 2 + h) * 128 + col) * 8 + c8] = (unsigned short)(hp & 0xffffu);
            out[(((base + 1) * 2 + h) * 128 + col) * 8 + c8] = (unsigned short)(lp & 0xffffu);
        }
    }
    if (sat != 0) atomicAdd(sat_word ? sat_word : &g_wsplit_saturated, sat);
}

unsigned int wsplit_saturation_count(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wsplit_saturated), sizeof(v)) != hipSuccess) return 0;
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wsplit_saturated), &z, sizeof(z));
    }
    return v;
}

}  // namespace sgdfr

using namespace sgdfr;

// geometry; returns 0 when the shape cannot use the kernel
static int wsplit_geometry(int B, int Cin, int Cout, int H, int W, WsParams* out) {
    if (B < 1 || Cin % WS_CB != 0 || Cout % 128 != 0 || W % 2 != 0 || W < 16 || H < 1) return 0;
    WsParams p{};
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.TW = W / 2;
    p.TCT = p.TW >= 16 ? 16 : 8;
    p.tct_shift = p.TCT == 16 ? 4 : 3;
    p.TR = 128 / p.TCT;
    if (p.TW % p.TCT != 0 || H % p.TR != 0) return 0;
    p.tiles_x = p.TW / p.TCT;
    p.tiles_y = H / p.TR;
    p.xs = (p.TR + 2) * p.TCT;
    if ((16 * p.xs) % 64 != 0 || 16 * p.xs > 5 * 512) return 0;
    p.n_pix_tiles = B * p.tiles_x * p.tiles_y;
    p.n_cout_tiles = Cout / 128;
    if ((int64_t)p.n_pix_tiles * p.n_cout_tiles >= (1ll << 30) || (int64_t)B * Cout * H * W >= (1ll << 40)) return 0;
    p.fd_xs = make_fastdiv(p.xs);
    p.fd_tiles_x = make_fastdiv(p.tiles_x);
    p.fd_per_img = make_fastdiv(p.tiles_x * p.tiles_y);
    p.fd_npt = make_fastdiv(p.n_pix_tiles);
    if (out) *out = p;
    return 1;
}

extern "C" int sgdfr_modconv2d_wsplit_supported(int B, int Cin, int Cout, int H, int W) {
    return wsplit_geometry(B, Cin, Cout, H, W, nullptr);
}

extern "C" int64_t sgdfr_modconv_prepack_wsplit_elems(int Cout, int Cin) { return (int64_t)Cout * Cin * 12 * 2; }

extern "C" int sgdfr_modconv_prepack_wsplit_f32(const float* weight, unsigned short* wsp, int Cout, int Cin, int arith,
                                                unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "prepack_wsplit: arith must be SGDFR_SPLIT_BF16/FP16");
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && Cin % WS_CB == 0 && Cout % 128 == 0,
                  "prepack_wsplit: needs Cin %% 16 == 0 and Cout %% 128 == 0, got Cin=%d Cout=%d", Cin, Cout);
    SGDFR_REQUIRE(weight && wsp, "prepack_wsplit: null pointer");
    int64_t g = ((int64_t)Cout * Cin * 3 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prepack_wsplit_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wsp, Cout, Cin,
                       (arith == SGDFR_SPLIT_FP16 ? WS_F16_WSCALE : 1.f) / sqrtf((float)Cin * 9), arith, sat);
    return check_launch("modconv_prepack_wsplit");
}

extern "C" int sgdfr_to_wsplit_f32(const float* x, const float* s, unsigned short* vs, int B, int Cin, int H, int W, int arith,
                                   unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cin % 8 == 0 && H > 0 && W > 0 && W % 2 == 0, "to_wsplit: bad shape B=%d Cin=%d H=%d W=%d (Cin %% 8, even W)", B, Cin, H, W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "to_wsplit: arith must be SGDFR_SPLIT_BF16/FP16");
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && s && vs && (reinterpret_cast<uintptr_t>(vs) & 15) == 0, "to_wsplit: null or misaligned pointer");
    int64_t g = ((int64_t)B * (Cin / 8) * H * (W / 2) + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (arith == SGDFR_SPLIT_FP16)
        hipLaunchKernelGGL(to_wsplit_kernel<SGDFR_SPLIT_FP16>, dim3((int)g), dim3(256), 0, as_stream(stream), x, s,
                           reinterpret_cast<unsigned char*>(vs), B, Cin, H, W, sat);
    else
        hipLaunchKernelGGL(to_wsplit_kernel<SGDFR_SPLIT_BF16>, dim3((int)g), dim3(256), 0, as_stream(stream), x, s,
                           reinterpret_cast<unsigned char*>(vs), B, Cin, H, W, sat);
    return check_launch("to_wsplit");
}

extern "C" int sgdfr_modconv2d_wsplit_f32(const unsigned short* v, const unsigned short* wsp, const float* d, const float* noise,
                                          int64_t noise_bstride, const float* noise_w, const float* bias, const float* zeros,
                                          float* y, const float* rgb_w, const float* rgb_s, float* rgb_part,
                                          unsigned short* xs_out, const float* s_next, int B, int Cin, int Cout, int H, int W,
                                          int arith, int act, float slope, float gain, unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "modconv_wsplit: arith must be SGDFR_SPLIT_BF16/FP16");
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv_wsplit: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin,
                  Cout, H, W);
    if (B == 0) return 0;
    WsParams p;
    SGDFR_REQUIRE(wsplit_geometry(B, Cin, Cout, H, W, &p),
                  "modconv_wsplit: shape B=%d Cin=%d Cout=%d H=%d W=%d not supported (Cin %% 16, Cout %% 128, W/2 %% min(16, W/2), "
                  "H %% (128 / tile columns)); use sgdfr_modconv2d_split_f32", B, Cin, Cout, H, W);
    SGDFR_REQUIRE(v && wsp && zeros && (y || rgb_part || xs_out), "modconv_wsplit: null pointer");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(wsp)) & 15) == 0,
                  "modconv_wsplit: v and wsp must be 16-byte aligned");
    SGDFR_REQUIRE(!noise || (noise_w && (reinterpret_cast<uintptr_t>(noise) & 7) == 0 && noise_bstride % 2 == 0),
                  "modconv_wsplit: noise needs noise_w and 8-byte alignment");
    SGDFR_REQUIRE(!y || (reinterpret_cast<uintptr_t>(y) & 7) == 0, "modconv_wsplit: y must be 8-byte aligned");
    SGDFR_REQUIRE(!xs_out || (s_next && (reinterpret_cast<uintptr_t>(xs_out) & 15) == 0), "modconv_wsplit: xs_out needs s_next and a 16-byte aligned buffer");
    SGDFR_REQUIRE(!rgb_part || (rgb_w && rgb_s && (reinterpret_cast<uintptr_t>(rgb_part) & 7) == 0), "modconv_wsplit: the fused ToRGB needs rgb_w and rgb_s");
    p.v = reinterpret_cast<const unsigned char*>(v);
    p.wsp = reinterpret_cast<const unsigned char*>(wsp);
    p.zeros = reinterpret_cast<const unsigned char*>(zeros);
    p.d = d; p.noise = noise; p.noise_bstride = noise_bstride; p.noise_w = noise_w; p.bias = bias; p.y = y;
    p.rgb_w = rgb_w; p.rgb_s = rgb_s; p.rgb_part = rgb_part;
    p.xs_out = reinterpret_cast<unsigned char*>(xs_out); p.s_next = s_next; p.sat = sat;
    p.act = act; p.slope = slope; p.gain = gain;
    p.dbg = getenv("SGDFR_WSPLIT_DBG") ? atoi(getenv("SGDFR_WSPLIT_DBG")) : 0;
    const size_t lds = 2 * (size_t)256 * p.xs + 2 * 32768 + 7 * 128 * sizeof(float);
    void (*kern)(WsParams) = arith == SGDFR_SPLIT_FP16 ? wsplit_kernel<SGDFR_SPLIT_FP16> : wsplit_kernel<SGDFR_SPLIT_BF16>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv_wsplit: LDS request %zu B refused", lds);
        return 2;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_pix_tiles * p.n_cout_tiles), dim3(512), lds, as_stream(stream), p);
    return check_launch("modconv2d_wsplit");
}
