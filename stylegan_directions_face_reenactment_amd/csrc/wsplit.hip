// Plain 3x3 modulated convolution in 1-D Winograd form -- F(2,3) or F(4,3) along image rows, kernel rows direct -- on the
// split-operand 16-bit matrix cores (the arithmetic of split.hip: every fp32 operand as two 16-bit terms hi + lo, three MFMA
// products, fp32 accumulation).  The text below describes F(2,3) (POS = 4 transform positions); F(4,3) (POS = 6: four outputs
// from six transformed inputs, HALF the MFMA work, V 1.5x and U 2x the direct operands' bytes, six accumulators per wave, block
// 128 couts x 64 four-pixel tiles) is the same kernel with three positions per half-stage -- see the kernel's template comment.
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
#include "wsplit_common.h"

namespace sgdfr {

__device__ unsigned int g_wsplit_saturated = 0;     // clamped operand pairs of launches without a saturation word

// POS = transform positions of the 1-D Winograd form: 4 = F(2,3) (two outputs per tile), 6 = F(4,3) (four outputs per tile:
// 18 instead of 36 MFMA columns per 16 channels and output quad, V 1.5x and U 2x the direct operands' bytes).
// XSF8: the hand-over's lo chunks leave as fp8 cross-term operands (SGDFR_SPLIT_HANDOVER_F8; its own instantiation, as in wswide.hip)
template <int ET, int POS, bool XSF8 = false>
__global__ __launch_bounds__(512, 1) void wsplit_kernel(WsParams p) {
    constexpr int NT = 128;
    constexpr int OUTP = POS - 2;                 // output pixels per tile
    constexpr int NI = POS == 4 ? 2 : 1;          // 32-tile MFMA column tiles per wave: POS * NI accumulators (8 / 6)
    constexpr int TILES = 64 * NI;                // tiles per block (2 column waves): 256 pixels either way
    constexpr int HPOS = POS / 2;                 // positions per half-stage
    constexpr int RUNS = POS * 4;                 // (position, part, k-half) runs of the V image
    constexpr int NEXV = POS == 4 ? 5 : 4;        // V pieces per wave and channel block (the last one may not exist for every wave)
    constexpr int WHALF = HPOS * 8192;            // one half-stage of one cout tile: [HPOS][part 2][k-half 2][128][8] x 16 bit = one ring slot
    constexpr int WV = HPOS;                      // 1 KB DMA pieces per wave and half-stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xbuf_bytes = RUNS * 16 * p.xs;      // [t POS][part 2][k-half 2][xs][8] x 16 bit
    unsigned char* const xb0 = smem;
    unsigned char* const wb0 = smem + 2 * xbuf_bytes;
    float* const dl = reinterpret_cast<float*>(wb0 + 4 * WHALF);     // [NT] d * output scale (* gain)
    float* const bl = dl + NT;                                        // [NT] bias
    float* const sn = bl + NT;                                        // [NT] next layer's style * range shift
    float* const cw = sn + NT;                                        // [NT][4] ToRGB coefficients
    float* const red = reinterpret_cast<float*>(wb0 + 3 * WHALF);    // [4][256 px][3] after the K loop: the ring slot of the last half-stage

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W, HT = p.H * p.TW, G8 = p.Cin / 8;

    // Persistent blocks: the grid is one block per CU (or fewer).  The tiles are dealt to the XCDs (blockIdx & 7) in eight contiguous
    // ranges -- the blocks resident on one XCD walk neighbouring patches of one cout tile (weights and halo rows shared in that
    // XCD's L2), and the XCDs work on DIFFERENT cout tiles of the same patches at the same time (the second reader of a V patch
    // finds it in the Infinity Cache; with round-robin rounds over all tiles it came back 4 rounds later: 256@64^2 +1.4 %) -- and
    // a block takes every (blocks of its XCD)-th tile of its XCD's range.  With one block per tile this is the plain XCD-aware
    // order.  The last channel block of a tile stages the first channel block and the first three weight half-slabs of the
    // block's NEXT tile (and loads its epilogue coefficients), so a tile does not start with ~100 KB of exposed DMA latency.
    struct Tile { int ct, img0, row0, col0; };
    const int per_img = p.tiles_x * p.tiles_y;
    auto lid_of = [&](int j) -> int {          // j-th tile of this block, -1: none
        const int nblk = (int)gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int bk = (nblk >> 3) + (xcd < (nblk & 7) ? 1 : 0);                              // blocks on this XCD
        const int tq = p.total_blocks >> 3, tr = p.total_blocks & 7;
        const int nk = tq + (xcd < tr ? 1 : 0), sk = xcd * tq + min(xcd, tr);                // this XCD's tiles: [sk, sk + nk)
        const int local = idx + j * bk;
        return local < nk ? sk + local : -1;
    };
    auto tile_of = [&](int lid) -> Tile {
        Tile t;
        t.ct = fdiv(lid, p.fd_npt);
        const int pt = lid - t.ct * p.n_pix_tiles;
        t.img0 = fdiv(pt, p.fd_per_img);
        const int prem = pt - t.img0 * per_img;
        const int ty = fdiv(prem, p.fd_tiles_x), tx = prem - ty * p.tiles_x;
        t.row0 = ty * p.TR;
        t.col0 = tx * p.TCT;
        return t;
    };
    // staging descriptor of V piece e of a tile: item i = (run (t, part, k-half), position) -> one 16-byte chunk; a wave's piece is
    // 64 consecutive items, so the LDS image is simply item order.  -1: zero page (rows outside the image), -2: no item.
    auto vsrc_of = [&](const Tile& T, int e) -> int64_t {
        const int i = (e * 8 + wave) * 64 + lane;
        if ((e * 8 + wave) * 64 >= RUNS * p.xs) return -2;
        const int run = fdiv(i, p.fd_xs), pos = i - run * p.xs;
        const int sr = pos >> p.tct_shift, c = pos & (p.TCT - 1);
        const int row = T.row0 - 1 + sr;
        const int t = run >> 2, part = (run >> 1) & 1, h = run & 1;
        return (row >= 0 && row < p.H)
                   ? ((((((int64_t)T.img0 * G8 + h) * POS + t) * 2 + part) * HT) + (int64_t)row * p.TW + T.col0 + c) * 16
                   : -1;
    };
    const bool fuse_rgb = p.rgb_part != nullptr, emit_xs = p.xs_out != nullptr;
    // epilogue coefficients of a tile: global loads early (top of a tile / first half-stage of the previous tile's last channel
    // block), LDS writes in the prologue; they go through LDS because loads between stores would serialise on vmcnt
    struct TabVals { float d, b, s, r, w0, w1, w2; };
    auto tables_load = [&](const Tile& T) -> TabVals {
        TabVals t{1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tid < NT) {
            const int n0 = T.ct * NT;
            const int64_t bc = (int64_t)T.img0 * p.Cout + n0 + tid;
            t.d = p.d ? p.d[bc] : 1.f;
            t.b = p.bias ? p.bias[n0 + tid] : 0.f;
            if (emit_xs) t.s = p.s_next[bc];
            if (fuse_rgb) {
                t.r = p.rgb_s[bc];
                t.w0 = p.rgb_w[n0 + tid];
                t.w1 = p.rgb_w[p.Cout + n0 + tid];
                t.w2 = p.rgb_w[2 * p.Cout + n0 + tid];
            }
        }
        return t;
    };
    const int ntab = wave < 2 ? ((p.d ? 1 : 0) + (p.bias ? 1 : 0) + (emit_xs ? 1 : 0) + (fuse_rgb ? 4 : 0)) : 0;   // loads tables_load issues
    // (the activation gain is folded into d, bias and noise: lrelu(t) * gain = max(g t, slope * g t) for gain > 0, 0 < slope <= 1)
    const float e_slope = p.act ? p.slope : 1.f, e_gain = p.act ? p.gain : 1.f;
    const int64_t v_cb_stride = (int64_t)(POS * 64) * HT;      // 2 eight-channel groups x [POS][2][HT][16 B]
    const int ncb = p.Cin / WS_CB;
    const int nh = ncb * 6;                             // half-stages: (channel block, kernel row, position half)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto issue_v_at = [&](int64_t src_off, int e, int cb, unsigned char* xb) {
        if (src_off == -2) return;                       // wave-uniform
#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 8) return;
#endif
        const unsigned char* src = src_off >= 0 ? p.v + src_off + cb * v_cb_stride : p.zeros;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(xb + (e * 8 + wave) * 1024), 16, 0, 0);
    };
    // weight slab of half-stage h = (cb, ky, tp) of a cout tile: [HPOS positions][part][k-half][128][8], WV pieces per wave
    auto issue_w_of = [&](const unsigned char* wg, int h) {
#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 16) return;
#endif
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int piece = wave + v * 8;
            __builtin_amdgcn_global_load_lds((glb_void*)(wg + (int64_t)h * WHALF + piece * 1024 + lane * 16),
                                             (lds_void*)(wb0 + (h & 3) * WHALF + piece * 1024), 16, 0, 0);
        }
    };
    // V pieces this wave really issues per half-stage of a channel block (the last piece may exist for the first waves only):
    // piece e goes out in half-stage max(0, e - (NEXV - 4)) -- all landed and published by the barrier that ends half-stage 4
    int nvw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < NEXV; ++e) nvw[e - (NEXV - 4) > 0 ? e - (NEXV - 4) : 0] += ((e * 8 + wave) * 64 < RUNS * p.xs) ? 1 : 0;
    const int a_off = (hi * 128 + wm * 32 + l31) * 16;
    // the two waves of a SIMD issue their DMA pieces at different times, waves 4-7 behind the first MFMAs of the half-stage's last
    // position (split.hip's finding holds here: 2.5-3 % on every F(4,3) layer in a same-process A/B, SGDFR_WSPLIT_DBG=4 switches it off)
    const bool late = wave >= 4 && !(p.dbg & 4);
    const int rowstep = p.TCT * 16;
    // a tile may stage its successor only when the ring and the V buffers come round: an even number of channel blocks
    const bool can_prefetch = (ncb & 1) == 0 && !(p.dbg & 128);

    // First-round desynchronisation (split.hip's transposed conv): equal blocks started together reach their store phase
    // together; spreading the starts of the first round over a fraction of a block time lets later rounds store while other CUs
    // compute.
    if (p.desync > 0 && blockIdx.x < 256) {
        const int slot = (int)((blockIdx.x * 2654435761u) >> 24);          // 0..255, scrambled
        const int n_sleep = (slot * p.desync) >> 8;
        for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(64);
    }

    unsigned sat = 0;
    bool prefetched = false;       // this tile's first channel block, weight half-slabs and coefficients were staged by the previous tile
    TabVals tvn{};
    for (int jt = 0;; ++jt) {
    const int lid = lid_of(jt);
    if (lid < 0) break;
    const int lid_n = lid_of(jt + 1);
    const bool has_next = can_prefetch && lid_n >= 0;
    const Tile T = tile_of(lid), Tn = has_next ? tile_of(lid_n) : T;
    const int ct = T.ct, img0 = T.img0, row0 = T.row0, col0 = T.col0, n0 = ct * NT;
    const TabVals tv = prefetched ? tvn : tables_load(T);

    // this lane's tile columns: position inside the staged patch, first output pixel, noise
    int boff[NI], pix[NI];
    float nz[NI][OUTP];
    {
        const float nw = ((p.noise && p.noise_w) ? p.noise_w[0] : 0.f) * e_gain;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int l = (wn * NI + n) * 32 + l31;
            const int r = l >> p.tct_shift, c = l & (p.TCT - 1);
            boff[n] = (hi * p.xs + l) * 16;                      // (staged row r + ky holds image row row0 - 1 + r + ky)
            pix[n] = (row0 + r) * p.W + OUTP * (col0 + c);
#pragma unroll
            for (int q = 0; q < OUTP; ++q) nz[n][q] = 0.f;
            if (p.noise) {
                const float* np_ = p.noise + (int64_t)img0 * p.noise_bstride + pix[n];
                if (OUTP == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(np_);
                    nz[n][0] = nw * t2.x; nz[n][1] = nw * t2.y;
                } else {
                    const float4 t4 = *reinterpret_cast<const float4*>(np_);
                    nz[n][0] = nw * t4.x; nz[n][1] = nw * t4.y; nz[n][OUTP - 2] = nw * t4.z; nz[n][OUTP - 1] = nw * t4.w;
                }
            }
        }
    }
    int64_t vsrc[NEXV];
#pragma unroll
    for (int e = 0; e < NEXV; ++e) vsrc[e] = vsrc_of(T, e);
    const unsigned char* const wglb = p.wsp + (int64_t)ct * ncb * 6 * WHALF;
    const unsigned char* const wglb_n = p.wsp + (int64_t)Tn.ct * ncb * 6 * WHALF;

    ws_f32x16 acc[POS][NI];
#pragma unroll
    for (int t = 0; t < POS; ++t)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;

    // ---- prologue: channel block 0 and the first three weight half-slabs (unless the previous tile staged them); tables
    if (!prefetched) {
#pragma unroll
        for (int e = 0; e < NEXV; ++e) issue_v_at(vsrc[e], e, 0, xb0);
        issue_w_of(wglb, 0);
        if (nh > 1) issue_w_of(wglb, 1);
        if (nh > 2) issue_w_of(wglb, 2);
    }
    if (tid < NT) {
        const float oscale = (ET == SGDFR_SPLIT_FP16) ? WS_F16_OUT : 1.f;
        const float xsc = (ET == SGDFR_SPLIT_FP16) ? WS_F16_XSCALE : 1.f;
        const float rs = rsqrtf((float)p.Cout);
        dl[tid] = tv.d * oscale * e_gain;
        bl[tid] = tv.b * e_gain;
        sn[tid] = tv.s * xsc;
        *reinterpret_cast<float4*>(cw + 4 * tid) = make_float4(tv.w0 * (tv.r * rs), tv.w1 * (tv.r * rs), tv.w2 * (tv.r * rs), 0.f);
    }
    if (!prefetched) {
        ws_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // (a staged tile starts behind the barrier that ended the previous one: every wave had waited for its pieces before it)

    // K loop.  The weight half-slabs live in a FOUR-slot ring and are DMA'd three half-stages ahead: the slab of h+1 is
    // published by the barrier that ends h-1, so the first fragments of h+1 are requested during the last position of h --
    // BEFORE the barrier that ends h -- and no wave starts a half-stage waiting for LDS.  (One barrier per kernel row with two
    // slots, all eight waves asking for their first fragments right behind it, measured the same: the loop is power-bound.)
    // The barrier's counted wait leaves the pieces issued during h in flight and completes everything older (the slab of h+2,
    // V pieces of the next channel block).
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
        const bool pre_next = !v_next && has_next;      // the last channel block stages the next tile's first one
#pragma unroll
        for (int hh = 0; hh < 6; ++hh, ++h) {
            const int ky = hh >> 1, tp = hh & 1;
            int n_issued = 0;
            auto issue_all = [&]() {
                if (h + 3 < nh) { issue_w_of(wglb, h + 3); n_issued += WV; }
                else if (has_next) { issue_w_of(wglb_n, h + 3 - nh); n_issued += WV; }      // (nh % 4 == 0: the ring comes round)
                if ((v_next || pre_next) && hh < 4) {
#pragma unroll
                    for (int e = 0; e < NEXV; ++e)
                        if ((e - (NEXV - 4) > 0 ? e - (NEXV - 4) : 0) == hh) {
                            if (v_next) issue_v_at(vsrc[e], e, cb + 1, xnext);
                            else issue_v_at(vsrc_of(Tn, e), e, 0, xnext);
                        }
                    n_issued += nvw[hh < 4 ? hh : 0];
                }
                if (pre_next && hh == 0) {      // the next tile's coefficients: in flight with this half-stage's pieces
                    tvn = tables_load(Tn);
                    n_issued += ntab;
                }
            };
            if (!late) issue_all();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* wsl = wb0 + (h & 3) * WHALF + a_off;
            const unsigned char* xrow = xcur + ky * rowstep;
#pragma unroll
            for (int tl = 0; tl < HPOS; ++tl) {
                const int t = tp * HPOS + tl;
                const int cur = (hh * HPOS + tl) & 1;            // (6 * HPOS is even: the parity is the same in every channel block)
                // fragments of this position were requested during the previous one (the first of a half-stage: before the
                // barrier that precedes it)
#pragma unroll
                for (int n = 0; n < NI; ++n) acc[t][n] = ws_mfma<ET>(a[cur][0], b[cur][0][n], acc[t][n]);
                __builtin_amdgcn_sched_barrier(0);
                if (tl == HPOS - 1 && late) issue_all();
                if (tl + 1 < HPOS) {
                    fetch(cur ^ 1, wsl, xrow, tl + 1, t + 1);
                } else if (h + 1 < nh) {
                    const int hn = hh + 1;      // (6 = the first half-stage of the next channel block)
                    fetch(cur ^ 1, wb0 + ((h + 1) & 3) * WHALF + a_off, (hn == 6 ? xnext : xcur) + (hn == 6 ? 0 : (hn >> 1)) * rowstep, 0,
                          hn == 6 ? 0 : HPOS * (hn & 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NI; ++n) acc[t][n] = ws_mfma<ET>(a[cur][0], b[cur][1][n], acc[t][n]);
#pragma unroll
                for (int n = 0; n < NI; ++n) acc[t][n] = ws_mfma<ET>(a[cur][1], b[cur][0][n], acc[t][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (h + 1 < nh) {
                // (the first half-stage of a staged tile waits for nothing: what it publishes landed before the previous tile's
                // epilogue, and what is in flight now are that epilogue's stores -- loads and stores retire through one in-order counter)
                if (!(prefetched && h == 0)) {
#ifdef SGDFR_WSPLIT_PROBE
                    if (!(p.dbg & 32))
#endif
                    ws_wait_vmcnt_dyn(n_issued);
                }
                __builtin_amdgcn_s_barrier();
            } else if (has_next) {
                ws_wait_vmcnt<0>();      // the next tile's first stages have landed before this tile's stores join the queue
                // (its coefficients too: "use" them here, so the compiler's own wait for these loads sits at this point and not
                //  behind the epilogue's stores at the top of the next tile)
                asm volatile("" : "+v"(tvn.d), "+v"(tvn.b), "+v"(tvn.s), "+v"(tvn.r), "+v"(tvn.w0), "+v"(tvn.w1), "+v"(tvn.w2));
            }
        }
        xsel ^= 1;
    }
    prefetched = has_next;

#ifdef SGDFR_WSPLIT_PROBE
    if (p.dbg & 2) { __syncthreads(); continue; }
#endif
    // ---- epilogue.  C/D layout of 32x32: column (tile) = lane & 31, row (cout) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // The element loop works on PAIRS of neighbouring pixels as two-float vectors: scale + noise + bias, the leaky-ReLU product,
    // the next layer's style and the ToRGB sums are v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (one instruction per pair; no MFMA
    // is in flight here, so the packed forms cost nothing extra) -- 13 instead of 18 VALU instructions per output in chain form.
    constexpr int NP = OUTP / 2;
    ws_f32x2 rgb2[NI][NP][3];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) rgb2[n][pp][0] = rgb2[n][pp][1] = rgb2[n][pp][2] = (ws_f32x2){0.f, 0.f};
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
                ws_f32x2 v2[NP][4];      // [pixel pair of the tile][row of the group]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    float y[OUTP];
                    if (POS == 4) {      // A^T of F(2,3)
                        const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[POS - 1][n][r];
                        y[0] = (m0 + m1) + m2;
                        y[1] = (m1 - m2) - m3;
                    } else {             // A^T of F(4,3): rows (1 1 1 1 1 0) (0 1 -1 2 -2 0) (0 1 1 4 4 0) (0 1 -1 8 -8 1)
                        const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r], m4 = acc[POS - 2][n][r],
                                    m5 = acc[POS - 1][n][r];
                        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                        y[0] = (m0 + s12) + s34;
                        y[1] = fmaf(2.f, d34, d12);
                        y[OUTP - 2] = fmaf(4.f, s34, s12);
                        y[OUTP - 1] = fmaf(8.f, d34, d12) + m5;
                    }
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        const ws_f32x2 yy = {y[2 * pp], y[2 * pp + 1]}, nn = {nz[n][2 * pp], nz[n][2 * pp + 1]};
                        const ws_f32x2 t = yy * dv[j] + (nn + bv[j]);
                        const ws_f32x2 ts = t * e_slope;
                        v2[pp][j] = (ws_f32x2){fmaxf(t[0], ts[0]), fmaxf(t[1], ts[1])};
                    }
                }
                if (HAS_Y) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* dst = yp + (int64_t)(8 * g + j) * HW;
                        if (OUTP == 2) *reinterpret_cast<float2*>(dst) = make_float2(v2[0][j][0], v2[0][j][1]);
                        else *reinterpret_cast<float4*>(dst) = make_float4(v2[0][j][0], v2[0][j][1], v2[NP - 1][j][0], v2[NP - 1][j][1]);
                    }
                }
                if (EMIT_XS) {      // the 4 rows are half of one 8-channel chunk of each pixel of the tile
                    const float4 sq = s4p[2 * g];
                    const float sv[4] = {sq.x, sq.y, sq.z, sq.w};
                    unsigned char* dst = xp + (int64_t)g * 2 * HW * 16;
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        ws_f32x2 pr[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) pr[j] = v2[pp][j] * sv[j];
                        unsigned h01[2], l01[2], h23[2], l23[2];      // [pixel of the pair]
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            ws_pair<ET>(pr[0][k], pr[1][k], h01[k], l01[k], sat);
                            ws_pair<ET>(pr[2][k], pr[3][k], h23[k], l23[k], sat);
                        }
                        {
                            // The two lane halves hold channels 4*hi .. 4*hi+3 of the same pixels.  One v_permlane32_swap per
                            // register (lanes 32..63 of the first pixel's <-> lanes 0..31 of the second's) gives the lower half
                            // the first pixel's whole 8-channel chunk and the upper half the second's: 16-byte stores, neighbouring
                            // chunks from the two halves in one instruction (32 contiguous bytes per tile instead of 2 x 8).
                            auto swap32 = [](unsigned& x0, unsigned& x1) {
                                const auto r2 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
                                x0 = r2[0]; x1 = r2[1];
                            };
                            swap32(h01[0], h01[1]); swap32(h23[0], h23[1]); swap32(l01[0], l01[1]); swap32(l23[0], l23[1]);
                            unsigned char* const d16 = dst - 8 * hi + 16 * (2 * pp + hi);
                            const uint4 vh = make_uint4(h01[0], h23[0], h01[1], h23[1]);
                            uint4 vl = make_uint4(l01[0], l23[0], l01[1], l23[1]);
                            if constexpr (XSF8) ws_f8_lo_chunk(vh, vl, exp2f((float)WS_F8_XLO), exp2f((float)WS_F8_XHI), false, sat);
                            *reinterpret_cast<uint4*>(d16) = vh;
                            *reinterpret_cast<uint4*>(d16 + (int64_t)HW * 16) = vl;
                        }
                    }
                }
                if (FUSE_RGB) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 q4 = cwp[8 * g + j];
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) {
                            rgb2[n][pp][0] = v2[pp][j] * q4.x + rgb2[n][pp][0];
                            rgb2[n][pp][1] = v2[pp][j] * q4.y + rgb2[n][pp][1];
                            rgb2[n][pp][2] = v2[pp][j] * q4.z + rgb2[n][pp][2];
                        }
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
            for (int q = 0; q < OUTP; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float mine = rgb2[n][q >> 1][j][q & 1];
                    const float vv = mine + __shfl_xor(mine, 32, 64);
                    if (hi == 0) red[(wm * 256 + ((wn * NI + n) * 32 + l31) * OUTP + q) * 3 + j] = vv;
                }
        __syncthreads();
        if (wm == 0 && hi == 0) {
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float tq[OUTP];
#pragma unroll
                    for (int q = 0; q < OUTP; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) t += red[(w2 * 256 + ((wn * NI + n) * 32 + l31) * OUTP + q) * 3 + j];
                        tq[q] = t;
                    }
                    float* dst = p.rgb_part + (((int64_t)img0 * p.n_cout_tiles + ct) * 3 + j) * HW + pix[n];
                    if (OUTP == 2) *reinterpret_cast<float2*>(dst) = make_float2(tq[0], tq[1]);
                    else *reinterpret_cast<float4*>(dst) = make_float4(tq[0], tq[1], tq[OUTP - 2], tq[OUTP - 1]);
                }
        }
    }
    __syncthreads();      // the next tile refills the tables; every wave has left the staging buffers and `red`
    }
    if (ET == SGDFR_SPLIT_FP16 && __builtin_expect(sat != 0, 0)) atomicAdd(p.sat ? p.sat : &g_wsplit_saturated, sat);
}

template <int POS>
__device__ __forceinline__ void ws_weight_transform(float g0, float g1, float g2, float (&u)[POS]) {
    if (POS == 4) {
        u[0] = g0;
        u[1] = 0.5f * ((g0 + g2) + g1);
        u[2] = 0.5f * ((g0 + g2) - g1);
        u[POS - 1] = g2;
    } else {
        // G rows: (1/4 0 0) (-1/6 -1/6 -1/6) (-1/6 1/6 -1/6) (1/24 1/12 1/6) (1/24 -1/12 1/6) (0 0 1)
        const float s02 = g0 + g2, q = fmaf(4.f, g2, g0);          // g0 + 4 g2
        u[0] = 0.25f * g0;
        u[1] = (-1.f / 6.f) * (s02 + g1);
        u[2] = (-1.f / 6.f) * (s02 - g1);
        u[3] = (1.f / 24.f) * fmaf(2.f, g1, q);
        u[POS - 2] = (1.f / 24.f) * fmaf(-2.f, g1, q);
        u[POS - 1] = g2;
    }
}

// x [B,Cin,H,W] fp32 and s [B,Cin] -> WS [B][Cin/8][t POS][hi,lo][H * W/(POS-2)][8]: B^T (x*s) per output tile, split.
// One thread = one tile of one 8-channel group.
template <int ET, int POS>
__global__ __launch_bounds__(256) void to_wsplit_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       unsigned char* __restrict__ vs, int B, int Cin, int H, int W,
                                                       unsigned* __restrict__ sat_word) {
    constexpr int OUTP = POS - 2;
    constexpr int ETM = ws_main_et<ET>::value;
    const int G = Cin / 8, TW = W / OUTP, HT = H * TW;
    const int64_t n = (int64_t)B * G * HT;
    const float sc = (ETM == SGDFR_SPLIT_FP16) ? WS_F16_XSCALE : 1.f;
    unsigned sat = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int pos = (int)(idx % HT);
        const int64_t bg = idx / HT;
        const int g = (int)(bg % G), b = (int)(bg / G);
        const int row = pos / TW, tc = pos - row * TW;
        float v[POS][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* xp = x + (((int64_t)b * Cin + g * 8 + c) * H + row) * W + OUTP * tc;
            const float sv = s[(int64_t)b * Cin + g * 8 + c] * sc;
            float d[POS], vv[POS];
#pragma unroll
            for (int j = 0; j < POS; ++j) {
                const int col = OUTP * tc - 1 + j;
                d[j] = (col >= 0 && col < W) ? xp[j - 1] * sv : 0.f;
            }
            ws_input_transform<POS>(d, vv);
#pragma unroll
            for (int t = 0; t < POS; ++t) v[t][c] = vv[t];
        }
#pragma unroll
        for (int t = 0; t < POS; ++t) {
            uint4 vh, vl;
            unsigned* ph = reinterpret_cast<unsigned*>(&vh);
            unsigned* pl = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
            for (int c = 0; c < 4; ++c) ws_pair<ETM>(v[t][2 * c], v[t][2 * c + 1], ph[c], pl[c], sat);
            if (ET == SGDFR_SPLIT_FP16F8) ws_f8_lo_chunk(vh, vl, exp2f((float)WS_F8_XLO), exp2f((float)WS_F8_XHI), false, sat);
            unsigned char* dst = vs + (((bg * POS + t) * 2) * HT + pos) * 16;
            *reinterpret_cast<uint4*>(dst) = vh;
            *reinterpret_cast<uint4*>(dst + (int64_t)HT * 16) = vl;
        }
    }
    if (ETM == SGDFR_SPLIT_FP16 && sat != 0) atomicAdd(sat_word ? sat_word : &g_wsplit_saturated, sat);
}

// weight [Cout,Cin,3,3] fp32 -> 16-bit hi/lo of U = G (weight/sqrt(9 Cin)) per kernel row, in the kernel's LDS order:
//   [cout tile 128][cin block][ky][t POS][part][k-half][128 couts][8 cin]
// SGDFR_SPLIT_FP16F8: max |w| (as float bits: non-negative floats order like unsigned ints) into the pack's trailer word
__global__ __launch_bounds__(256) void wsplit_absmax_kernel(const float* __restrict__ w, int64_t n, float scale, unsigned* __restrict__ trailer) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i] * scale));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(trailer, __builtin_bit_cast(unsigned, m));
}

template <int POS>
__global__ __launch_bounds__(256) void prepack_wsplit_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                            int Cout, int Cin, float scale, int et, unsigned* __restrict__ sat_word) {
    const int64_t n = (int64_t)Cout * Cin * 3;
    const int ncb = Cin / WS_CB;
    unsigned sat = 0;
    // fp8 cross terms: the lo chunk of (cout, 8 channels) is bytes (w_hi * 2^-EW) x 8 | (w_lo * 2^(11 - EW)) x 8, EW from the trailer
    float f8_hi = 0.f, f8_lo = 0.f;
    if (et == SGDFR_SPLIT_FP16F8) {
        const int ew = ws_f8_wexp(*reinterpret_cast<const float*>(out + (int64_t)Cout * Cin * 3 * POS * 2));
        f8_hi = exp2f((float)-ew); f8_lo = exp2f((float)(11 - ew));
    }
    unsigned char* const out8 = reinterpret_cast<unsigned char*>(out);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ky = (int)(idx % 3);
        const int ci = (int)((idx / 3) % Cin);
        const int co = (int)(idx / (3 * (int64_t)Cin));
        const float* g = w + ((int64_t)co * Cin + ci) * 9 + ky * 3;
        float U[POS];
        ws_weight_transform<POS>(g[0] * scale, g[1] * scale, g[2] * scale, U);
        const int ctile = co / 128, col = co - ctile * 128, cb = ci / WS_CB, h = (ci % WS_CB) / 8, c8 = ci % 8;
#pragma unroll
        for (int t = 0; t < POS; ++t) {
            unsigned hp, lp;
            if (et != SGDFR_SPLIT_BF16) ws_pair<SGDFR_SPLIT_FP16>(U[t], 0.f, hp, lp, sat);
            else ws_pair<SGDFR_SPLIT_BF16>(U[t], 0.f, hp, lp, sat);
            const int64_t base = ((((int64_t)ctile * ncb + cb) * 3 + ky) * POS + t) * 2;      // -> [part]
            out[(((base + 0) * 2 + h) * 128 + col) * 8 + c8] = (unsigned short)(hp & 0xffffu);
            if (et == SGDFR_SPLIT_FP16F8) {
                const float fh = (float)__builtin_bit_cast(_Float16, (unsigned short)(hp & 0xffffu));
                const float fl = (float)__builtin_bit_cast(_Float16, (unsigned short)(lp & 0xffffu));
                const int two = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(fh * f8_hi, -448.f, 448.f), __builtin_amdgcn_fmed3f(fl * f8_lo, -448.f, 448.f), 0, false);
                unsigned char* const chunk = out8 + (((base + 1) * 2 + h) * 128 + col) * 16 + (c8 >> 2) * 8 + (c8 & 3);      // (ws_f8_half's order)
                chunk[0] = (unsigned char)(two & 0xff);
                chunk[4] = (unsigned char)((two >> 8) & 0xff);
            } else {
                out[(((base + 1) * 2 + h) * 128 + col) * 8 + c8] = (unsigned short)(lp & 0xffffu);
            }
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
    return v + wswide_saturation_count(reset);
}

}  // namespace sgdfr

using namespace sgdfr;

// geometry for F(f,3), f = 2 | 4 outputs per tile; returns 0 when the shape cannot use the kernel
static int wsplit_geometry(int B, int Cin, int Cout, int H, int W, int f, WsParams* out) {
    if ((f != 2 && f != 4) || B < 1 || Cin % WS_CB != 0 || Cout % 128 != 0 || W % f != 0 || W < 16 || H < 1) return 0;
    WsParams p{};
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.TW = W / f;
    const int tiles = f == 2 ? 128 : 64, runs = (f + 2) * 4;          // tiles per block; (position, part, k-half) runs
    const int tct_max = f == 2 ? 16 : 8;
    p.TCT = p.TW >= tct_max ? tct_max : tct_max / 2;
    p.tct_shift = p.TCT == 16 ? 4 : p.TCT == 8 ? 3 : 2;
    p.TR = tiles / p.TCT;
    if (p.TW % p.TCT != 0 || H % p.TR != 0) return 0;
    p.tiles_x = p.TW / p.TCT;
    p.tiles_y = H / p.TR;
    p.xs = (p.TR + 2) * p.TCT;
    if ((runs * p.xs) % 64 != 0 || runs * p.xs > (f == 2 ? 5 : 4) * 512) return 0;
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

extern "C" int sgdfr_modconv2d_wsplit_supported(int B, int Cin, int Cout, int H, int W, int f) {
    return wsplit_geometry(B, Cin, Cout, H, W, f, nullptr);
}

extern "C" int sgdfr_modconv2d_wsplit_wide(int B, int Cin, int Cout, int H, int W) {
    WsParams p;
    return wsplit_geometry(B, Cin, Cout, H, W, 4, &p) && Cin % (2 * WS_CB) == 0 ? wswide_by_tile_count(p) : 0;
}

// (+ 8: a 16-byte trailer; SGDFR_SPLIT_FP16F8 packs keep max |w * scale| there, the source of their fp8 exponent)
extern "C" int64_t sgdfr_modconv_prepack_wsplit_elems(int Cout, int Cin, int f) { return (int64_t)Cout * Cin * 3 * (f + 2) * 2 + 8; }

extern "C" int sgdfr_modconv_prepack_wsplit_f32(const float* weight, unsigned short* wsp, int Cout, int Cin, int f, int arith,
                                                unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || (arith == SGDFR_SPLIT_FP16F8 && f == 4),
                  "prepack_wsplit: arith must be SGDFR_SPLIT_BF16/FP16 (or FP16F8 with f = 4)");
    SGDFR_REQUIRE(f == 2 || f == 4, "prepack_wsplit: f (outputs per Winograd tile) must be 2 or 4, got %d", f);
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && Cin % WS_CB == 0 && Cout % 128 == 0,
                  "prepack_wsplit: needs Cin %% 16 == 0 and Cout %% 128 == 0, got Cin=%d Cout=%d", Cin, Cout);
    SGDFR_REQUIRE(weight && wsp, "prepack_wsplit: null pointer");
    int64_t g = ((int64_t)Cout * Cin * 3 + 255) / 256;
    if (g > 4096) g = 4096;
    const float scale = (arith != SGDFR_SPLIT_BF16 ? WS_F16_WSCALE : 1.f) / sqrtf((float)Cin * 9);
    {
        unsigned* const trailer = reinterpret_cast<unsigned*>(wsp + (int64_t)Cout * Cin * 3 * (f + 2) * 2);
        if (hipMemsetAsync(trailer, 0, 16, as_stream(stream)) != hipSuccess) { (void)hipGetLastError(); set_error("prepack_wsplit: memset failed"); return 2; }
        if (arith == SGDFR_SPLIT_FP16F8)
            hipLaunchKernelGGL(wsplit_absmax_kernel, dim3(256), dim3(256), 0, as_stream(stream), weight, (int64_t)Cout * Cin * 9, scale, trailer);
    }
    if (f == 2)
        hipLaunchKernelGGL(prepack_wsplit_kernel<4>, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wsp, Cout, Cin, scale, arith, sat);
    else
        hipLaunchKernelGGL(prepack_wsplit_kernel<6>, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wsp, Cout, Cin, scale, arith, sat);
    return check_launch("modconv_prepack_wsplit");
}

extern "C" int sgdfr_to_wsplit_f32(const float* x, const float* s, unsigned short* vs, int B, int Cin, int H, int W, int f, int arith,
                                   unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(f == 2 || f == 4, "to_wsplit: f (outputs per Winograd tile) must be 2 or 4, got %d", f);
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cin % 8 == 0 && H > 0 && W > 0 && W % f == 0, "to_wsplit: bad shape B=%d Cin=%d H=%d W=%d (Cin %% 8, W %% f)", B, Cin, H, W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || (arith == SGDFR_SPLIT_FP16F8 && f == 4),
                  "to_wsplit: arith must be SGDFR_SPLIT_BF16/FP16 (or FP16F8 with f = 4)");
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && s && vs && (reinterpret_cast<uintptr_t>(vs) & 15) == 0, "to_wsplit: null or misaligned pointer");
    int64_t g = ((int64_t)B * (Cin / 8) * H * (W / f) + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    unsigned char* out = reinterpret_cast<unsigned char*>(vs);
    void (*kern)(const float*, const float*, unsigned char*, int, int, int, int, unsigned*) =
        arith == SGDFR_SPLIT_FP16F8 ? to_wsplit_kernel<SGDFR_SPLIT_FP16F8, 6>
        : arith == SGDFR_SPLIT_FP16 ? (f == 2 ? to_wsplit_kernel<SGDFR_SPLIT_FP16, 4> : to_wsplit_kernel<SGDFR_SPLIT_FP16, 6>)
                                    : (f == 2 ? to_wsplit_kernel<SGDFR_SPLIT_BF16, 4> : to_wsplit_kernel<SGDFR_SPLIT_BF16, 6>);
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(256), 0, as_stream(stream), x, s, out, B, Cin, H, W, sat);
    return check_launch("to_wsplit");
}

extern "C" int sgdfr_modconv2d_wsplit_f32(const unsigned short* v, const unsigned short* wsp, const float* d, const float* noise,
                                          int64_t noise_bstride, const float* noise_w, const float* bias, const float* zeros,
                                          float* y, const float* rgb_w, const float* rgb_s, float* rgb_part,
                                          unsigned short* xs_out, const float* s_next, int B, int Cin, int Cout, int H, int W,
                                          int f, int arith, int act, float slope, float gain, unsigned int* sat, void* stream) {
    const int xs_f8 = (arith & SGDFR_SPLIT_HANDOVER_F8) ? 1 : 0;      // (a flag beside the input's arithmetic)
    arith &= ~SGDFR_SPLIT_HANDOVER_F8;
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || (arith == SGDFR_SPLIT_FP16F8 && f == 4),
                  "modconv_wsplit: arith must be SGDFR_SPLIT_BF16/FP16 (or FP16F8 with f = 4)");
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv_wsplit: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin,
                  Cout, H, W);
    if (B == 0) return 0;
    WsParams p;
    SGDFR_REQUIRE(wsplit_geometry(B, Cin, Cout, H, W, f, &p),
                  "modconv_wsplit: shape B=%d Cin=%d Cout=%d H=%d W=%d f=%d not supported (f = 2 | 4, Cin %% 16, Cout %% 128, W/f a "
                  "multiple of the patch width, H of the patch height); use sgdfr_modconv2d_split_f32", B, Cin, Cout, H, W, f);
    SGDFR_REQUIRE(v && wsp && zeros && (y || rgb_part || xs_out), "modconv_wsplit: null pointer");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(wsp)) & 15) == 0,
                  "modconv_wsplit: v and wsp must be 16-byte aligned");
    const uintptr_t amask = (uintptr_t)(4 * f - 1);      // a lane stores / loads f consecutive floats
    SGDFR_REQUIRE(!noise || (noise_w && (reinterpret_cast<uintptr_t>(noise) & amask) == 0 && noise_bstride % f == 0),
                  "modconv_wsplit: noise needs noise_w and %d-byte alignment", 4 * f);
    SGDFR_REQUIRE(!y || (reinterpret_cast<uintptr_t>(y) & amask) == 0, "modconv_wsplit: y must be %d-byte aligned", 4 * f);
    SGDFR_REQUIRE(!xs_out || (s_next && (reinterpret_cast<uintptr_t>(xs_out) & 15) == 0), "modconv_wsplit: xs_out needs s_next and a 16-byte aligned buffer");
    SGDFR_REQUIRE(!rgb_part || (rgb_w && rgb_s && (reinterpret_cast<uintptr_t>(rgb_part) & amask) == 0), "modconv_wsplit: the fused ToRGB needs rgb_w and rgb_s");
    p.v = reinterpret_cast<const unsigned char*>(v);
    p.wsp = reinterpret_cast<const unsigned char*>(wsp);
    p.zeros = reinterpret_cast<const unsigned char*>(zeros);
    p.d = d; p.noise = noise; p.noise_bstride = noise_bstride; p.noise_w = noise_w; p.bias = bias; p.y = y;
    p.rgb_w = rgb_w; p.rgb_s = rgb_s; p.rgb_part = rgb_part;
    p.xs_out = reinterpret_cast<unsigned char*>(xs_out); p.s_next = s_next; p.sat = sat;
    p.act = act; p.slope = slope; p.gain = gain;
    p.dbg = getenv("SGDFR_WSPLIT_DBG") ? atoi(getenv("SGDFR_WSPLIT_DBG")) : 0;
    p.xs_f8 = xs_f8;
    SGDFR_REQUIRE(!xs_f8 || (xs_out && arith != SGDFR_SPLIT_BF16 && f == 4), "modconv_wsplit: SGDFR_SPLIT_HANDOVER_F8 needs xs_out, f = 4 and an fp16 arithmetic");
    if (f == 4) {      // 128 couts x 128 tiles per block where the tile count allows it (csrc/wswide.hip: the same outputs, bit for bit)
        const int rc = wswide_try_launch(p, arith, stream);
        if (rc >= 0) return rc;
    }
    SGDFR_REQUIRE(arith != SGDFR_SPLIT_FP16F8,
                  "modconv_wsplit: SGDFR_SPLIT_FP16F8 runs on the wide-tile kernel only (Cin %% 32 == 0, Cin >= 64, Cout %% 128 == 0, W %% 32 == 0, "
                  "H %% 16 == 0, d and bias given); got Cin=%d Cout=%d H=%d W=%d", Cin, Cout, H, W);
    {
        // same-process A/B at B=64 (scripts/wsplit_env_ab.py, 60 % of the block-time estimate): 256@64^2 (8 rounds of blocks) 618 ->
        // 607 us, 128@128^2 (16 rounds) 733 -> 713 us, 512@32^2 (4 rounds: the spread costs part of a round) 557 -> 568 us
        const int pct = getenv("SGDFR_WSPLIT_DESYNC") ? atoi(getenv("SGDFR_WSPLIT_DESYNC")) : 60;
        const double block_clk = (double)(Cin / WS_CB) * (f == 2 ? 72 : 54) * 32 * 2 / 0.7 + 12000.0;
        const int blocks = p.n_pix_tiles * p.n_cout_tiles;
        p.desync = (pct > 0 && blocks >= 2048) ? (int)(block_clk * pct / 100 / 4096) : 0;
    }
    // V double buffer + four weight half-slabs + tables: f = 2: 80 + 64 + 3.5 KB, f = 4: 60 + 96 + 3.5 KB (of 160)
    const size_t lds = 2 * (size_t)(f + 2) * 64 * p.xs + 4 * (size_t)((f + 2) / 2) * 8192 + 7 * 128 * sizeof(float);
    void (*kern)(WsParams) = arith == SGDFR_SPLIT_FP16 ? (f == 2 ? wsplit_kernel<SGDFR_SPLIT_FP16, 4> : xs_f8 ? wsplit_kernel<SGDFR_SPLIT_FP16, 6, true> : wsplit_kernel<SGDFR_SPLIT_FP16, 6>)
                                                        : (f == 2 ? wsplit_kernel<SGDFR_SPLIT_BF16, 4> : wsplit_kernel<SGDFR_SPLIT_BF16, 6>);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv_wsplit: LDS request %zu B refused", lds);
        return 2;
    }
    // Persistent blocks, one per CU, from two tiles per CU on: a block's tile stages the first channel block, the first three
    // weight half-slabs and the epilogue coefficients of the block's next tile (see the kernel).
    const int persist = getenv("SGDFR_WSPLIT_PERSIST") ? atoi(getenv("SGDFR_WSPLIT_PERSIST")) : 256;      // (read per launch: scripts/wsplit_env_ab.py)
    p.total_blocks = p.n_pix_tiles * p.n_cout_tiles;
    // (a persistent grid deals its tiles to the 8 XCDs -- blockIdx & 7 -- so it needs a block on each: fewer than 8 would skip ranges)
    const int grid = (persist >= 8 && p.total_blocks >= 2 * persist) ? persist : p.total_blocks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, as_stream(stream), p);
    return check_launch("modconv2d_wsplit");
}
