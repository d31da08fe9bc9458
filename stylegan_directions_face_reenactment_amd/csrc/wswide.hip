// The F(4,3) split-Winograd plain conv of wsplit.hip on a 128 couts x 128 tiles (512 pixels) block tile -- twice the pixels per
// staged weight slab.  Same algebra, same operand formats (the WS input of sgdfr_to_wsplit_f32 / the blur's hand-over, the pack of
// sgdfr_modconv_prepack_wsplit_f32(f = 4)), same epilogue arithmetic: the outputs are the same bits as wsplit_kernel<ET, 6>'s.
//
// Why (round 5, scripts/wsplit_form_probe.hip): at 128 x 64 the transformed weights U are 85 % of the operand bytes a block stages
// (144 KB of U + 30 KB of V per 16 input channels for 432 MFMAs) and every U byte is used by two waves only; the K loop of that form
// runs 668 algorithmic TFLOP/s in a bare loop at its own instruction ratios against 880 without the DMA.  With 128 tiles per block
// the same U slab feeds four column waves: 144 + 54 KB per 864 MFMAs (0.57x the bytes per MFMA), and the wave tile grows to 64 couts
// x 32 tiles (12 accumulators, 6 fragment reads per 6 MFMAs instead of 4 per 3): 784 TFLOP/s in the probe, +17 %.  The forms the
// verdict proposed instead -- four waves with two accumulator sets and the epilogue as MFMA-gap filler -- price at 662 (K loop alone)
// and 608 (with 1.4 filler instructions per MFMA) in the same probe: no faster than today's loop, DESIGN 4.10.
//
// What had to change to make it fit (160 KB of LDS, 256 registers per wave at two waves per SIMD):
//  * POSITION-OUTER K loop: (Winograd position t -> channel block -> kernel row).  One position is accumulated over ALL channel
//    blocks -- two live MFMA accumulators per wave (64 couts x 32 tiles of M_t) -- and the finished M_t is folded into the output
//    transform A^T M at once, in wsplit_kernel's expression tree, so at most 160 registers hold accumulators or partial outputs
//    (all twelve M_t at once = 192 of 256: hipcc then rotated one through scratch every channel block).  Every accumulator still
//    sums (channel block, kernel row, product term) in wsplit_kernel's order: same bits.
//  * A group = (position, channel block) needs the three (ky, t) chunks of U (24 KB: contiguous 8 KB chunks of the existing pack)
//    and only V_t = [part][k-half][144 staged positions] (9 KB), so neither operand is double-buffered per channel block: two rings
//    of four group slots (96 + 36 KB), every group's operands DMA'd three groups (~2.5-3k clocks of MFMA issue) ahead, one s_barrier
//    and one counted vmcnt per group.  wsplit_kernel's V double buffer alone would be 108 KB at this tile.
//  * The rings run ACROSS tiles: the look-ahead of a tile's last three groups stages the first three groups of the block's next
//    tile (persistent blocks), so a tile starts with its operands in LDS.
//  * Per-lane state cut to a handful of registers: operand DMA by BUFFER loads (descriptor in SGPRs, 32-bit lane offset; rows
//    outside the image are out-of-range offsets = zeros from the bounds check), epilogue coefficients by 4-byte LDS-DMA (no staging
//    registers), noise fetched in the epilogue, lane coordinates laundered through an asm so the epilogue's addresses are not
//    hoisted above the K loop.
#include "wsplit_common.h"

namespace sgdfr {

__device__ unsigned int g_wswide_saturated = 0;     // clamped operand pairs of launches without a saturation word

constexpr int WW_USLAB = 3 * 8192;                  // [ky 3][part 2][k-half 2][128 couts][8 x 16 bit]
constexpr int WW_XS = 144;                          // staged positions of a patch: (TR + 2) x TCT = 18 x 8
constexpr int WW_VSLAB = 4 * WW_XS * 16;            // [part 2][k-half 2][144][8 x 16 bit] = 9216 B = 9 DMA pieces
constexpr int WW_RING = 4;
constexpr int WW_TCT = 8, WW_TR = 16;

// LDS traffic of this wave done + workgroup barrier -- without the vmcnt(0) a __syncthreads() may carry (the epilogue's stores and
// the look-ahead's DMAs stay in flight across it)
__device__ __forceinline__ void ws_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// XSF8: the hand-over's lo chunks leave as fp8 cross-term operands (SGDFR_SPLIT_HANDOVER_F8) -- its own instantiation: the conversion
// inside the epilogue of the plain kernels (behind a block-uniform branch) cost them 870 spilled registers
template <int ET, bool XSF8 = false>
__global__ __launch_bounds__(512, 1) void wswide_kernel(WsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the buffer-resource type does not exist in the host pass: without this the host stub is never instantiated)
    constexpr int POS = 6, OUTP = 4, NT = 128, MI = 2, NP = OUTP / 2;
    // F8 (SGDFR_SPLIT_FP16F8): the main term on the fp16 matrix path, BOTH cross terms of two (kernel row, channel block) slices in
    // one v_mfma_scale_f32_32x32x64_f8f6f4 -- 6 + 3 x 2 = 12 MFMA units per channel-block pair and 32-cout tile instead of 18.  The
    // hi chunks, every scale and the hand-over are the fp16 arithmetic's.
    constexpr bool F8 = (ET == SGDFR_SPLIT_FP16F8);
    constexpr int ETM = ws_main_et<ET>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ub0 = smem;
    unsigned char* const vb0 = smem + WW_RING * WW_USLAB;
    float* const tab = reinterpret_cast<float*>(vb0 + WW_RING * WW_VSLAB);      // [7][128] raw: d, bias, s_next, rgb style, rgb weights 0..2

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W, HT = p.H * p.TW, G8 = p.Cin / 8;
    const int ncb = p.Cin / WS_CB;
    const bool fuse_rgb = p.rgb_part != nullptr, emit_xs = p.xs_out != nullptr;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    // ---- tiles of this block (persistent; the XCD-aware order of wsplit_kernel)
    struct Tile { int ct, img0, row0, col0; };
    const int per_img = p.tiles_x * p.tiles_y;
    auto lid_of = [&](int j) -> int {
        const int nblk = (int)gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int bk = (nblk >> 3) + (xcd < (nblk & 7) ? 1 : 0);
        const int tq = p.total_blocks >> 3, tr = p.total_blocks & 7;
        const int nk = tq + (xcd < tr ? 1 : 0), sk = xcd * tq + min(xcd, tr);
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
        t.row0 = ty * WW_TR;
        t.col0 = tx * WW_TCT;
        return t;
    };
    // Operand DMA goes through BUFFER loads (buffer_load_dwordx4 ... lds: resource descriptor in SGPRs + a 32-bit lane offset + a
    // uniform offset) -- the flat form needs a 64-bit address per lane and piece, and with 192 of 256 registers holding
    // accumulators those temporaries spilled (and a scratch reload waits vmcnt(0): it drains the DMA queue).  Rows of a patch that
    // lie outside the image are lane offsets beyond the descriptor's size: the hardware's bounds check writes zeros to LDS.
    constexpr int OOB = (int)0x80000000;
    // this lane's item of V piece e of a tile: byte offset inside one image of the WS tensor at channel block 0, position 0.
    // Item i = e * 64 + lane = (run = (part, k-half), staged position).
    auto voff_of = [&](const Tile& T, int e) -> int {
        const int i = e * 64 + lane;
        const int run = i / WW_XS, pos = i - run * WW_XS;
        const int row = T.row0 - 1 + (pos >> 3), c = pos & 7;
        const int part = run >> 1, h = run & 1;
        return (row >= 0 && row < p.H) ? (((h * 12 + part) * HT + row * p.TW + T.col0 + c) * 16) : OOB;
    };
    const int img_bytes = G8 * 12 * HT * 16;                           // one image of the WS tensor (< 2^31: checked by the launcher)
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wsp), 0, p.n_cout_tiles * ncb * (18 * 8192), 0x00020000);
    // DMA sources of the tile the look-ahead currently points into
    struct Src { int w_off; const unsigned char* v; int voff, voff8; };
    const int lane_piece = wave * 1024 + lane * 16;
    auto src_of = [&](const Tile& T) -> Src {
        Src s;
        s.w_off = T.ct * ncb * (18 * 8192);
        s.v = p.v + (int64_t)T.img0 * img_bytes;
        s.voff = voff_of(T, wave);
        s.voff8 = voff_of(T, 8);
        return s;
    };
    // operands of group (cb, t) of a tile -> ring slot: U = the three (ky, t) chunks of the pack (8 KB each, one piece per wave and
    // kernel row), V_t = 9 pieces (wave 0 takes the ninth).
    auto issue_group = [&](const Src& S, int cb, int t, int slot) {
#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 16) return;
#endif
        // (wave-uniform values, said so explicitly: left in a VGPR an soffset makes hipcc wrap the load in a waterfall loop)
        const int wg = __builtin_amdgcn_readfirstlane(S.w_off + (cb * 18 + t) * 8192);
        const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(S.v), 0, img_bytes, 0x00020000);
        const int vg = __builtin_amdgcn_readfirstlane((cb * 24 + t * 2) * HT * 16);
        auto go = [&](auto unt, auto vnt) {      // cache policy of the two operand streams (aux = 2: non-temporal)
            constexpr int UA = decltype(unt)::value, VA = decltype(vnt)::value;
#pragma unroll
            for (int v = 0; v < 3; ++v)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(ub0 + slot * WW_USLAB + (v * 8 + wave) * 1024), 16, lane_piece,
                                                         wg + v * (6 * 8192), 0, UA);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lds_void*)(vb0 + slot * WW_VSLAB + wave * 1024), 16, S.voff, vg, 0, VA);
            if (wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lds_void*)(vb0 + slot * WW_VSLAB + 8 * 1024), 16, S.voff8, vg, 0, VA);
        };
        using c0 = std::integral_constant<int, 0>;
        using c2 = std::integral_constant<int, 2>;
#ifdef SGDFR_WSPLIT_PROBE
        switch ((p.dbg >> 13) & 3) {
            case 1: go(c0{}, c2{}); break;
            case 2: go(c2{}, c0{}); break;
            case 3: go(c2{}, c2{}); break;
            default: go(c0{}, c0{}); break;
        }
#else
        go(c0{}, c0{});
#endif
    };
    // epilogue coefficients of a tile by 4-byte LDS-DMA: array k of `tab` is fetched by wave k (two 64-lane halves)
    auto issue_tables = [&](const Tile& T) {
        const int n0 = T.ct * NT;
        const int64_t bc = (int64_t)T.img0 * p.Cout + n0;
        const float* src = nullptr;
        switch (wave) {
            case 0: src = p.d + bc; break;
            case 1: src = p.bias + n0; break;
            case 2: src = emit_xs ? p.s_next + bc : nullptr; break;
            case 3: src = fuse_rgb ? p.rgb_s + bc : nullptr; break;
            case 4: src = fuse_rgb ? p.rgb_w + n0 : nullptr; break;
            case 5: src = fuse_rgb ? p.rgb_w + p.Cout + n0 : nullptr; break;
            case 6: src = fuse_rgb ? p.rgb_w + 2 * p.Cout + n0 : nullptr; break;
            default: break;
        }
        if (src == nullptr) return;      // wave-uniform
#pragma unroll
        for (int half = 0; half < 2; ++half)
            __builtin_amdgcn_global_load_lds((glb_void*)(src + half * 64 + lane), (lds_void*)(tab + wave * 128 + half * 64), 4, 0, 0);
    };

    // first-round start spread (wsplit_kernel's): equal blocks started together reach their store phase together
    if (p.desync > 0 && blockIdx.x < 256) {
        const int slot = (int)((blockIdx.x * 2654435761u) >> 24);
        const int n_sleep = (slot * p.desync) >> 8;
        for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(64);
    }

    const float e_slope = p.act ? p.slope : 1.f, e_gain = p.act ? p.gain : 1.f;
    const float d_mul = ((ETM == SGDFR_SPLIT_FP16) ? WS_F16_OUT : 1.f) * e_gain;
    const float s_mul = (ETM == SGDFR_SPLIT_FP16) ? WS_F16_XSCALE : 1.f;
    // F8: the cross-term MFMA's constant exponent, 2^(EW - 3) (wsplit_common.h); EW from the largest weight, kept in the pack's trailer
    int f8_sa = 127;
    if (F8) {
        const float maxw = *reinterpret_cast<const float*>(p.wsp + (size_t)p.n_cout_tiles * ncb * (18 * 8192));
        f8_sa = 127 + ws_f8_wexp(maxw) - WS_F8_XLO;
    }
    const float rgb_mul = rsqrtf((float)p.Cout);
    const float f8_mul_lo = exp2f((float)WS_F8_XLO), f8_mul_hi = exp2f((float)WS_F8_XHI);
    const int a_off = (hi * 128 + wm * 64 + l31) * 16;                   // + ((ky * 2 + part) * 2) * 2048 + mi * 512
    const int b_off = (hi * WW_XS + wn * 32 + l31) * 16;                 // + (part * 2) * 144 * 16 + ky * 8 * 16

    // where in a group a wave issues its look-ahead: 0 = before the first kernel row's MFMAs, 1 / 2 = behind the first / second row's
#ifdef SGDFR_WSPLIT_PROBE
    const int issue_at = (p.dbg & 0x100) ? (wave >= 4 ? (p.dbg >> 11) & 3 : (p.dbg >> 9) & 3) : 1;
#else
    constexpr int issue_at = 1;
#endif
    unsigned sat = 0;
    int G = 0;                    // groups issued to the rings so far = slot counter (runs across tiles)
    int lid = lid_of(0);
    if (lid < 0) return;
    Tile T = tile_of(lid);
    Src L = src_of(T);            // the tile the look-ahead points into
    // prime the rings: the first three groups of the first tile
    issue_group(L, 0, 0, 0);      // (channel block, position, slot): position 0 of channel blocks 0, 1, 2
    issue_group(L, 1, 0, 1);
    issue_group(L, 2, 0, 2);
    int grace = 0, prev_issued = 0;
    bool primed = false;
    for (int jt = 0;; ++jt) {
        const int lid_n = lid_of(jt + 1);
        const bool has_next = lid_n >= 0;
        const Tile Tn = has_next ? tile_of(lid_n) : T;
        const int img0 = T.img0, n0 = T.ct * NT;
        if (!primed) {
            ws_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            primed = true;
        }
        // POSITION-OUTER K loop: position t is accumulated over ALL channel blocks before the next one starts, so only two MFMA
        // accumulators (64 couts x 32 tiles of M_t) are live in the inner loop; a finished M_t is folded into the output transform
        // A^T M at once, in wsplit_kernel's expression tree -- y0 = (m0 + s12) + s34, y1 = fma(2, d34, d12), y2 = fma(4, s34, s12),
        // y3 = fma(8, d34, d12) + m5 with s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4 -- so the outputs are the same
        // bits, and at most 160 registers hold accumulators or partial transforms (all twelve M_t at once would be 192 of 256: the
        // register allocator then rotated one of them through scratch every channel block, each reload a vmcnt(0) that drains the DMA
        // queue).  Every accumulator still sums (channel block, kernel row, product term) in wsplit_kernel's order.
        ws_f32x16 Y[OUTP][MI];      // t = 0: Y0 = m0 | 1: Y1 = m1 | 2: Y1 = s12, Y2 = d12, Y0 = m0 + s12 | 3: Y3 = m3 | 4: Y0..Y3 = y0, y1, y2, y3 - m5 | 5: y3
#pragma unroll
        for (int t = 0; t < POS; ++t) {
            ws_f32x16 acc[MI];
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            ws_frag carry_a[MI], carry_b;      // F8: kernel row 2's lo fragments of an even channel block wait for row 0 of the next one
            auto group = [&](const int cb, auto phase_t) {
                constexpr int PH = decltype(phase_t)::value;      // 0: three fp16 products; 1 / 2: fp8 cross terms, even / odd channel block
                // look-ahead: the operands of the group three ahead (same position, or the next position's / the next TILE's first
                // channel blocks) go into the slot the previous group has just left
                int cur_issued = 4;       // pieces per wave and group (wave 0: 5 -- it then waits for one more of its own)
                auto look_ahead = [&]() {
                    int cbl = cb + 3, tl = t;
                    if (cbl >= ncb) { cbl -= ncb; tl = t + 1; }
                    if (tl < POS) {
                        issue_group(L, cbl, tl, (G + 3) & 3);
                    } else if (has_next) {
                        if (cbl == 0) L = src_of(Tn);
                        issue_group(L, cbl, 0, (G + 3) & 3);
                    } else {
                        cur_issued = 0;   // the block's last tile: nothing left to stage
                    }
                };
                // (every wave issues its pieces BEHIND the first kernel row's MFMAs, not in front of them: a group then opens with
                //  fragment reads + MFMAs on every SIMD.  Same-process A/B of the position per wave half, probe build, us per launch at
                //  512@32^2 / 256@64^2 / 128@128^2: both halves in front 475 / 519 / 604 (other box), waves 4-7 behind row 0 (wsplit.hip's
                //  arrangement) 469 / 515 / 593, BOTH behind row 0 456 / 501 / 593, waves 4-7 behind row 1 472 / 519 / 607)
                if (issue_at == 0) look_ahead();
                if (t == 0 && cb == 0) issue_tables(T);      // (the previous tile's epilogue is behind a barrier)
                const unsigned char* us = ub0 + (G & 3) * WW_USLAB + a_off;
                const unsigned char* vs = vb0 + (G & 3) * WW_VSLAB + b_off;
                ws_frag keep_a[MI], keep_b;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    ws_frag a[MI][2], b[2];
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        b[part] = *reinterpret_cast<const ws_frag*>(vs + part * (2 * WW_XS * 16) + ky * (WW_TCT * 16));
#pragma unroll
                        for (int m = 0; m < MI; ++m)
                            a[m][part] = *reinterpret_cast<const ws_frag*>(us + (ky * 2 + part) * 4096 + m * 512);
                    }
#ifdef SGDFR_WSPLIT_PROBE
                    if (p.dbg & 64) continue;
#endif
#pragma unroll
                    for (int m = 0; m < MI; ++m) acc[m] = ws_mfma<ETM>(a[m][0], b[0], acc[m]);
                    if (PH == 0) {
#pragma unroll
                        for (int m = 0; m < MI; ++m) acc[m] = ws_mfma<ETM>(a[m][0], b[1], acc[m]);
#pragma unroll
                        for (int m = 0; m < MI; ++m) acc[m] = ws_mfma<ETM>(a[m][1], b[0], acc[m]);
                    } else if ((PH == 1 && ky == 0) || (PH == 2 && ky == 1)) {      // first of a pair: keep
#pragma unroll
                        for (int m = 0; m < MI; ++m) keep_a[m] = a[m][1];
                        keep_b = b[1];
                    } else if (PH == 1 && ky == 2) {                                // left over: the next channel block's row 0 takes it
#pragma unroll
                        for (int m = 0; m < MI; ++m) carry_a[m] = a[m][1];
                        carry_b = b[1];
                    } else if (PH == 2 && ky == 0) {
#pragma unroll
                        for (int m = 0; m < MI; ++m) acc[m] = ws_mfma_f8(carry_a[m], a[m][1], carry_b, b[1], acc[m], f8_sa);
                    } else {
#pragma unroll
                        for (int m = 0; m < MI; ++m) acc[m] = ws_mfma_f8(keep_a[m], a[m][1], keep_b, b[1], acc[m], f8_sa);
                    }
                    if (ky + 1 == issue_at) look_ahead();
                }
                // The next group's operands (issued two groups ago) must have landed and be published; what this group and the
                // previous one issued stays in flight.  Loads and
                // stores retire through ONE in-order counter: after an epilogue the first two groups wait for nothing -- their
                // successors landed before the stores were issued (the vmcnt(0) of the tile's last group).
                const bool last = (t == POS - 1) && (cb == ncb - 1);
                if (last) {
                    ws_wait_vmcnt<0>();
                    grace = 2;
                } else if (grace > 0) {
                    --grace;
                } else {
#ifdef SGDFR_WSPLIT_PROBE
                    if (!(p.dbg & 32))
#endif
                    {
                        // (exactly what this group and the previous one issued may stay in flight -- a constant 8 would let the
                        //  NEXT group's operands count as "recent" once the look-ahead has run dry at the end of the last tile)
                        const int n = prev_issued + cur_issued;
                        if (n == 8) ws_wait_vmcnt<8>();
                        else if (n == 4) ws_wait_vmcnt<4>();
                        else ws_wait_vmcnt<0>();
                    }
                }
                prev_issued = cur_issued;
                __builtin_amdgcn_s_barrier();
            };
            using ph0 = std::integral_constant<int, 0>;
            using ph1 = std::integral_constant<int, 1>;
            using ph2 = std::integral_constant<int, 2>;
            if constexpr (!F8) {
#pragma unroll 1
                for (int cb = 0; cb < ncb; ++cb, ++G) group(cb, ph0{});
            } else {                                  // (Cin % 32 == 0: the launcher's condition)
#pragma unroll 1
                for (int cb = 0; cb < ncb; cb += 2) {
                    group(cb, ph1{}); ++G;
                    group(cb + 1, ph2{}); ++G;
                }
            }
            // fold M_t into the output transform (see above)
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                if (t == 0) Y[0][m] = acc[m];
                if (t == 1) Y[1][m] = acc[m];
                if (t == 2) {
                    const ws_f32x16 m1 = Y[1][m];
                    Y[1][m] = m1 + acc[m];                  // s12
                    Y[2][m] = m1 - acc[m];                  // d12
                    Y[0][m] = Y[0][m] + Y[1][m];            // m0 + s12
                }
                if (t == 3) Y[3][m] = acc[m];
                if (t == 4) {
                    const ws_f32x16 s34 = Y[3][m] + acc[m], d34 = Y[3][m] - acc[m], s12 = Y[1][m], d12 = Y[2][m];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        Y[0][m][r] = Y[0][m][r] + s34[r];
                        Y[1][m][r] = fmaf(2.f, d34[r], d12[r]);
                        Y[2][m][r] = fmaf(4.f, s34[r], s12[r]);
                        Y[3][m][r] = fmaf(8.f, d34[r], d12[r]);
                    }
                }
                if (t == 5) Y[3][m] = Y[3][m] + acc[m];
            }
        }

#ifdef SGDFR_WSPLIT_PROBE
        if (p.dbg & 2) { if (!has_next) break; T = Tn; continue; }
#endif
        // ---- epilogue (wsplit_kernel's arithmetic on the transformed outputs Y, per 32-cout MFMA tile m).  C/D layout of 32x32: column (tile) = lane & 31,
        // row (cout) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
        // (the lane coordinates pass through an opaque asm: everything per-lane the epilogue needs -- store addresses, table
        //  pointers -- is then computed HERE and not hoisted above the K loop, where 192 of 256 registers hold accumulators)
        int hi_e = hi, l31_e = l31;
        asm volatile("" : "+v"(hi_e), "+v"(l31_e));
        const int lt = wn * 32 + l31_e;                                   // this lane's tile of the patch: row lt >> 3, column lt & 7
        const int pix = (T.row0 + (lt >> 3)) * p.W + OUTP * (T.col0 + (lt & 7));
        float nz[OUTP];
        {
            const float nw = ((p.noise && p.noise_w) ? p.noise_w[0] : 0.f) * e_gain;
#pragma unroll
            for (int q = 0; q < OUTP; ++q) nz[q] = 0.f;
            if (p.noise) {
                const float4 t4 = *reinterpret_cast<const float4*>(p.noise + (int64_t)img0 * p.noise_bstride + pix);
                nz[0] = nw * t4.x; nz[1] = nw * t4.y; nz[2] = nw * t4.z; nz[3] = nw * t4.w;
            }
        }
        ws_f32x2 rgb2[MI][NP][3];      // per 32-cout MFMA tile: the four 32-cout partial sums of a pixel are added in wsplit_kernel's order
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) rgb2[m][pp][0] = rgb2[m][pp][1] = rgb2[m][pp][2] = (ws_f32x2){0.f, 0.f};
        auto epilogue = [&](auto has_y_t, auto emit_xs_t, auto fuse_rgb_t) {
            constexpr bool HAS_Y = decltype(has_y_t)::value, EMIT_XS = decltype(emit_xs_t)::value, FUSE_RGB = decltype(fuse_rgb_t)::value;
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                const int io = wm * 64 + m * 32 + 4 * hi_e;
                const float4* const d4p = reinterpret_cast<const float4*>(tab + io);
                const float4* const b4p = reinterpret_cast<const float4*>(tab + 128 + io);
                const float4* const s4p = reinterpret_cast<const float4*>(tab + 256 + io);
                float* const yp = p.y + ((int64_t)img0 * p.Cout + n0 + io) * HW + pix;
                unsigned char* const xp = p.xs_out + ((((int64_t)img0 * (p.Cout / 8) + (n0 + wm * 64 + m * 32) / 8) * 2) * HW + pix) * 16 + 8 * hi_e;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 dq = d4p[2 * g], bq = b4p[2 * g];
                    const float dv[4] = {dq.x * d_mul, dq.y * d_mul, dq.z * d_mul, dq.w * d_mul};
                    const float bv[4] = {bq.x * e_gain, bq.y * e_gain, bq.z * e_gain, bq.w * e_gain};
                    ws_f32x2 v2[NP][4];      // [pixel pair of the tile][row of the group]
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g + j;
                        const float y[OUTP] = {Y[0][m][r], Y[1][m][r], Y[2][m][r], Y[3][m][r]};      // A^T M, folded in the K loop
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) {
                            const ws_f32x2 yy = {y[2 * pp], y[2 * pp + 1]}, nn = {nz[2 * pp], nz[2 * pp + 1]};
                            const ws_f32x2 t = yy * dv[j] + (nn + bv[j]);
                            const ws_f32x2 ts = t * e_slope;
                            v2[pp][j] = (ws_f32x2){fmaxf(t[0], ts[0]), fmaxf(t[1], ts[1])};
                        }
                    }
                    if (HAS_Y) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<float4*>(yp + (int64_t)(8 * g + j) * HW) = make_float4(v2[0][j][0], v2[0][j][1], v2[1][j][0], v2[1][j][1]);
                    }
                    if (EMIT_XS) {      // the 4 rows are half of one 8-channel chunk of each pixel of the tile
                        const float4 sq = s4p[2 * g];
                        const float sv[4] = {sq.x * s_mul, sq.y * s_mul, sq.z * s_mul, sq.w * s_mul};
                        unsigned char* dst = xp + (int64_t)g * 2 * HW * 16;
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) {
                            ws_f32x2 pr[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) pr[j] = v2[pp][j] * sv[j];
                            unsigned h01[2], l01[2], h23[2], l23[2];      // [pixel of the pair]
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                ws_pair<ETM>(pr[0][k], pr[1][k], h01[k], l01[k], sat);
                                ws_pair<ETM>(pr[2][k], pr[3][k], h23[k], l23[k], sat);
                            }
                            // one v_permlane32_swap per register: the lower lane half gets the first pixel's whole 8-channel chunk,
                            // the upper half the second's (wsplit_kernel's 16-byte hand-over stores)
                            auto swap32 = [](unsigned& x0, unsigned& x1) {
                                const auto r2 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
                                x0 = r2[0]; x1 = r2[1];
                            };
                            swap32(h01[0], h01[1]); swap32(h23[0], h23[1]); swap32(l01[0], l01[1]); swap32(l23[0], l23[1]);
                            unsigned char* const d16 = dst - 8 * hi_e + 16 * (2 * pp + hi_e);
                            const uint4 vh = make_uint4(h01[0], h23[0], h01[1], h23[1]);
                            uint4 vl = make_uint4(l01[0], l23[0], l01[1], l23[1]);
                            if constexpr (XSF8) ws_f8_lo_chunk(vh, vl, f8_mul_lo, f8_mul_hi, false, sat);      // (the next conv reads fp8 cross-term operands)
                            *reinterpret_cast<uint4*>(d16) = vh;
                            *reinterpret_cast<uint4*>(d16 + (int64_t)HW * 16) = vl;
                        }
                    }
                    if (FUSE_RGB) {
                        const float4 rq = *reinterpret_cast<const float4*>(tab + 3 * 128 + io + 8 * g);
                        const float4 w0 = *reinterpret_cast<const float4*>(tab + 4 * 128 + io + 8 * g);
                        const float4 w1 = *reinterpret_cast<const float4*>(tab + 5 * 128 + io + 8 * g);
                        const float4 w2 = *reinterpret_cast<const float4*>(tab + 6 * 128 + io + 8 * g);
                        const float rr[4] = {rq.x * rgb_mul, rq.y * rgb_mul, rq.z * rgb_mul, rq.w * rgb_mul};
                        const float q0[4] = {w0.x * rr[0], w0.y * rr[1], w0.z * rr[2], w0.w * rr[3]};
                        const float q1[4] = {w1.x * rr[0], w1.y * rr[1], w1.z * rr[2], w1.w * rr[3]};
                        const float q2[4] = {w2.x * rr[0], w2.y * rr[1], w2.z * rr[2], w2.w * rr[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int pp = 0; pp < NP; ++pp) {
                                rgb2[m][pp][0] = v2[pp][j] * q0[j] + rgb2[m][pp][0];
                                rgb2[m][pp][1] = v2[pp][j] * q1[j] + rgb2[m][pp][1];
                                rgb2[m][pp][2] = v2[pp][j] * q2[j] + rgb2[m][pp][2];
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
        if (fuse_rgb) {      // the two lane halves and the two cout waves of a column hold different couts of the same pixels
            // (every wave is past the last group's barrier: its ring slot -- the one no look-ahead has refilled yet -- is free)
            float* const red = reinterpret_cast<float*>(ub0 + ((G - 1) & 3) * WW_USLAB);       // [4 cout groups of 32][512 px][3] = the slot exactly
#pragma unroll
            for (int m = 0; m < MI; ++m)
#pragma unroll
                for (int q = 0; q < OUTP; ++q)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float mine = rgb2[m][q >> 1][j][q & 1];
                        const float vv = mine + __shfl_xor(mine, 32, 64);
                        if (hi_e == 0) red[((wm * 2 + m) * 512 + lt * OUTP + q) * 3 + j] = vv;
                    }
            ws_lds_barrier();
            if (wm == 0 && hi_e == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float tq[OUTP];
#pragma unroll
                    for (int q = 0; q < OUTP; ++q) {
                        float t = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < 4; ++w2) t += red[(w2 * 512 + lt * OUTP + q) * 3 + j];
                        tq[q] = t;
                    }
                    *reinterpret_cast<float4*>(p.rgb_part + (((int64_t)img0 * p.n_cout_tiles + T.ct) * 3 + j) * HW + pix) =
                        make_float4(tq[0], tq[1], tq[2], tq[3]);
                }
            }
        }
        ws_lds_barrier();      // the next tile refills the tables and the free ring slot; every wave has left them
        if (!has_next) break;
        T = Tn;
    }
    if (ETM == SGDFR_SPLIT_FP16 && __builtin_expect(sat != 0, 0)) atomicAdd(p.sat ? p.sat : &g_wswide_saturated, sat);
#endif
}

unsigned int wswide_saturation_count(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wswide_saturated), sizeof(v)) != hipSuccess) return 0;
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wswide_saturated), &z, sizeof(z));
    }
    return v;
}

// The tile geometry of this kernel for the shape in p (wsplit_geometry's block for f = 4), and whether the launch takes it:
// mode 0 never, 1 by tile count, 2 whenever the shape allows.  -1: not this kernel's case.
static int wswide_plan(WsParams& p, int mode_now) {
    if (mode_now == 0) return -1;
    // (the look-ahead of three groups wraps at most once per position: >= 4 channel blocks)
    if (p.Cin % WS_CB != 0 || p.Cin < 4 * WS_CB || p.Cout % 128 != 0 || p.W % 32 != 0 || p.H % WW_TR != 0) return -1;
    const int old_blocks = p.n_pix_tiles * p.n_cout_tiles;
    p.TW = p.W / 4;
    p.TCT = WW_TCT; p.TR = WW_TR; p.tct_shift = 3;
    p.tiles_x = p.TW / WW_TCT;
    p.tiles_y = p.H / WW_TR;
    p.xs = WW_XS;
    p.n_pix_tiles = p.B * p.tiles_x * p.tiles_y;
    p.n_cout_tiles = p.Cout / 128;
    p.total_blocks = p.n_pix_tiles * p.n_cout_tiles;
    // buffer descriptors: one image of the WS tensor and the whole pack are addressed with 32-bit offsets
    if ((int64_t)(p.Cin / 8) * 12 * p.H * p.TW * 16 >= (1ll << 31) || (int64_t)p.Cout * p.Cin * 72 >= (1ll << 31)) return -1;
    if (mode_now == 1) {
        // a wide tile is two of wsplit_kernel's tiles at 0.76x the time of the pair (measured at B = 32 / 64 on 512@32^2, 256@64^2,
        // 128@128^2: 274 -> 209, 310 -> 232, 384 -> 278 us / 574 -> 430, 625 -> 487, 724 -> 552 us): take it when that wins after
        // rounding both to whole rounds of 256 CUs
        const int r_wide = (p.total_blocks + 255) / 256, r_old = (old_blocks + 255) / 256;
        if (p.total_blocks < 192 || r_wide * 2 * 0.76 >= (double)r_old) return -1;
    }
    p.fd_xs = make_fastdiv(p.xs);
    p.fd_tiles_x = make_fastdiv(p.tiles_x);
    p.fd_per_img = make_fastdiv(p.tiles_x * p.tiles_y);
    p.fd_npt = make_fastdiv(p.n_pix_tiles);
    return 0;
}

// does an F(4,3) launch of this shape (p: wsplit_geometry's block) take the wide kernel by its tile count?  (what the host asks before
// it chooses SGDFR_SPLIT_FP16F8 for a layer: only this kernel reads the fp8 chunks)
int wswide_by_tile_count(WsParams p) {
    static const int mode = getenv("SGDFR_WSPLIT_WIDE") ? atoi(getenv("SGDFR_WSPLIT_WIDE")) : 1;
    return mode != 0 && wswide_plan(p, 1) == 0 ? 1 : 0;
}

// p: the parameter block sgdfr_modconv2d_wsplit_f32 filled for wsplit_kernel<., 6> (pointers, shape, activation, dbg); the tile
// geometry is replaced by this kernel's.  -1: not this kernel's case.
int wswide_try_launch(WsParams p, int arith, void* stream) {
    static const int mode = getenv("SGDFR_WSPLIT_WIDE") ? atoi(getenv("SGDFR_WSPLIT_WIDE")) : 1;      // 0 off, 1 by tile count, 2 whenever the shape allows
    int mode_now = getenv("SGDFR_WSPLIT_WIDE_NOW") ? atoi(getenv("SGDFR_WSPLIT_WIDE_NOW")) : mode;     // (read per launch: same-process A/B)
    if (arith == SGDFR_SPLIT_FP16F8) {      // only this kernel reads the fp8 chunks: every shape it can run, or an error (never the 64-tile kernel)
        mode_now = 2;
        if (p.Cin % (2 * WS_CB) != 0) return -1;
    }
    if (!p.d || !p.bias || wswide_plan(p, mode_now) != 0) return -1;
    {
        // First-round start spread: OFF for this kernel.  Same-process A/B at B = 64 (SGDFR_WSWIDE_DESYNC = 0 / 30 / 60 / 100 % of a
        // block time, us per launch): 512@32^2 454 / 468 / 494 / 538, 256@64^2 492 / 505 / 515 / 538, 128@128^2 585 / 580 / 579 / 587
        // -- wsplit_kernel's 60 % costs the first two layers 5-9 % here (2-4 long tiles per block: the spread is a fraction of the
        // whole launch) and buys the third nothing.
        const int pct = getenv("SGDFR_WSWIDE_DESYNC") ? atoi(getenv("SGDFR_WSWIDE_DESYNC")) : 0;
        const double block_clk = (double)(p.Cin / WS_CB) * 108 * 32 * 2 / 0.7 + 20000.0;
        p.desync = pct > 0 ? (int)(block_clk * pct / 100 / 4096) : 0;
    }
    const size_t lds = (size_t)WW_RING * (WW_USLAB + WW_VSLAB) + 7 * 128 * sizeof(float);
    void (*kern)(WsParams) = arith == SGDFR_SPLIT_FP16F8 ? (p.xs_f8 ? wswide_kernel<SGDFR_SPLIT_FP16F8, true> : wswide_kernel<SGDFR_SPLIT_FP16F8>)
                             : arith == SGDFR_SPLIT_FP16 ? (p.xs_f8 ? wswide_kernel<SGDFR_SPLIT_FP16, true> : wswide_kernel<SGDFR_SPLIT_FP16>)
                                                         : wswide_kernel<SGDFR_SPLIT_BF16>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv_wsplit (wide tile): LDS request %zu B refused", lds);
        return 2;
    }
    const int persist = getenv("SGDFR_WSPLIT_PERSIST") ? atoi(getenv("SGDFR_WSPLIT_PERSIST")) : 256;
    // (a persistent grid deals its tiles to the 8 XCDs -- blockIdx & 7 -- so it needs a block on each: fewer than 8 would skip ranges)
    const int grid = (persist >= 8 && p.total_blocks >= 2 * persist) ? persist : p.total_blocks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, as_stream(stream), p);
    return check_launch("modconv2d_wsplit (wide tile)");
}

}  // namespace sgdfr
