// The split-operand conv kernel of split.hip as a header: split.hip instantiates the plain / transposed / adjoint variants,
// upfir.hip the transposed conv with the FIR blur fused into its epilogue (MODE_UPF).  Everything here is described in the
// comment at the top of split.hip.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sgdfr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int frag128 __attribute__((ext_vector_type(4)));   // 8 x 16-bit operand elements of one lane

// ET: element type of the split terms.  SGDFR_SPLIT_BF16: 8+8 mantissa bits, fp32 range.  SGDFR_SPLIT_FP16: 11+11 bits
// (the 22-bit sum is fp32-grade), fp16 range: activations are pre-scaled by 2^-4 and clamped to +-65504 before the
// split, weights by 2^6 in the pack, and the epilogue multiplies by 2^-2 (all exact powers of two).
template <int ET>
__device__ __forceinline__ f32x16 split_mfma(frag128 a, frag128 b, f32x16 c) {
    if (ET == SGDFR_SPLIT_FP16 || ET == SGDFR_SPLIT_FP16F8)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
static __device__ unsigned int g_split_saturated = 0;   // operand pairs clamped to the fp16 range by this translation unit's kernels

#ifdef SGDFR_SPLIT_PROBE
// scripts/tile_trace.py: s_memtime stamps of one wave of one block (SGDFR_SPLIT_DBG bit 6; block = dbg >> 8 & 0xff, wave =
// dbg >> 16 & 7, bit 7 adds one stamp per sub-stage): [0] = count, then (event id << 48) | shader clock
__device__ unsigned long long g_split_trace[4096];
#define SPLIT_TRACE(id)                                                                                              \
    do {                                                                                                             \
        if (trace_on) {                                                                                              \
            if (ntrace < 4094)                                                                                       \
                g_split_trace[1 + ntrace] = ((unsigned long long)(id) << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); \
            ++ntrace;                                                                                                \
        }                                                                                                            \
    } while (0)
#else
#define SPLIT_TRACE(id) do {} while (0)
#endif

// Internal fourth mode (upfir.hip): MODE_UP3's transposed conv over TR x TC patches of the super-pixel grid with the 4x4 FIR blur,
// noise, bias, leaky-ReLU and the hand-over to the next conv's split input form in the epilogue -- the parity planes never leave
// the CU (see the UPF epilogue below).
#define SGDFR_MODE_UPF 3

constexpr float SPLIT_F16_XSCALE = 0.0625f, SPLIT_F16_WSCALE = 64.f, SPLIT_F16_OUT = 0.25f, SPLIT_F16_MAX = 65504.f;

struct SplitParams {
    const float* x;
    int64_t x_bstride;
    const unsigned short* wsp;   // prepacked bf16 hi/lo weights
    const float* s;
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    const float* zeros;          // >= 8 zero floats
    float* y;
    const float* rgb_w;          // fused ToRGB (PLAIN3, ksplit == 1): [3][Cout] 1x1 weights, [B][Cout] styles,
    const float* rgb_s;          //   partial sums out [B][n_cout_tiles*3][H*W] (bias / skip are added by the finish launch)
    float* rgb_part;
    unsigned char* xs_out;       // PLAIN3, ksplit == 1: the activation in the NEXT layer's split input form (x * s_next, see XIN)
    const float* s_next;         //   [B][Cout] modulation of the next layer
    unsigned* sat;               // the caller's saturation word (null: the device-wide counter)
    int B, Cin, Cout, H, W;
    int P, R;                    // padded pitch / rows per image of the flat space (W+1, H+1)
    int n_pix_tiles, n_cout_tiles;
    int xs, xlen;                // staged positions per channel block: padded to a multiple of 64 / real
    int patch;                   // 0: flat runs ; 1: TR x TC patches
    int TC, TR, tiles_x, tiles_y, seglen;
    int simgs;                   // images one tile's staged range can touch
    int64_t total_pix;
    int act;
    float slope, gain;
    int ksplit;                  // K (channel-block) slices; > 1: slice ks writes d * partial sums to y + ks*split_stride
    int64_t split_stride;
    int desync;                  // first-round start spread: estimated block time in 4096-clock units (0 = off)
    int stagger;                 // which waves run MFMAs before staging inside a sub-stage (0 none, 1 waves 4-7, 2 odd waves)
    int dbg;
    int total_blocks;            // tiles x cout tiles x K slices; the grid may be smaller (persistent blocks)
    int rps;                     // UP3: positions per image of the flat space = stride of one parity plane, >= R*P (see plane_stride)
    int plane_il;                // UP3 with plane_stride: the four parity phases of a position are stored together, y [B][Cout][rps][px][py]
    int tstep_r, tstep_c, torg;  // patch tiles: tile (ty, tx) starts at (ty*tstep_r + torg, tx*tstep_c + torg) (plain conv: TR, TC, 0)
    int tpos;                    // UPF: positions per cout of the epilogue's exchange buffer (256 + a margin of TC + 1 either side)
    const float* fir;            // UPF: the 4x4 FIR taps on the device (row-major, as Blur.kernel)
    const float* f8_max;         // SGDFR_SPLIT_FP16F8: the pack's trailer (max |w * scale|, source of the weights' fp8 exponent)
    // divisors of the per-tile index arithmetic (fill_fastdivs() on the host, after the geometry is final)
    FastDiv fd_xs, fd_seglen, fd_P, fd_R, fd_RP, fd_rps, fd_HW, fd_W, fd_TC, fd_tiles_x, fd_per_img, fd_npt, fd_tps, fd_Cin;
    // (new fields go HERE, at the end: a field in the middle moves the kernel's scalar argument loads and has cost 12 % before)
    int lid0;                    // first tile id of this launch (total_blocks counts from it): the tail launch of launch_up_deep_tail()
};

static void fill_fastdivs(SplitParams& p) {
    p.fd_xs = make_fastdiv(p.xs); p.fd_seglen = make_fastdiv(p.seglen); p.fd_P = make_fastdiv(p.P); p.fd_R = make_fastdiv(p.R);
    p.fd_RP = make_fastdiv(p.R * p.P); p.fd_rps = make_fastdiv(p.rps); p.fd_HW = make_fastdiv(p.H * p.W); p.fd_W = make_fastdiv(p.W);
    p.fd_TC = make_fastdiv(p.TC); p.fd_tiles_x = make_fastdiv(p.tiles_x); p.fd_per_img = make_fastdiv(p.tiles_x * p.tiles_y);
    p.fd_npt = make_fastdiv(p.n_pix_tiles); p.fd_tps = make_fastdiv(p.n_pix_tiles * p.n_cout_tiles); p.fd_Cin = make_fastdiv(p.Cin);
}

constexpr int SPLIT_CB = 16;     // input channels per K block (one MFMA K)

template <int N>
__device__ __forceinline__ void split_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in 0..8 (the instruction takes an immediate)
__device__ __forceinline__ void split_wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: split_wait_vmcnt<0>(); break;
        case 1: split_wait_vmcnt<1>(); break;
        case 2: split_wait_vmcnt<2>(); break;
        case 3: split_wait_vmcnt<3>(); break;
        case 4: split_wait_vmcnt<4>(); break;
        case 5: split_wait_vmcnt<5>(); break;
        case 6: split_wait_vmcnt<6>(); break;
        case 7: split_wait_vmcnt<7>(); break;
        default: split_wait_vmcnt<8>(); break;
    }
}

// two floats -> packed hi pair and packed lo pair (lo = round(v - float(hi)), the subtraction is exact)
// `sat` counts the pairs this thread clamped; the caller adds it to g_split_saturated once (split_flush_saturation): a
// branch per pair would fence the scheduler around every conversion.
template <int ET>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo, unsigned& sat) {
    if (ET == SGDFR_SPLIT_FP16) {
        // saturation is never silent: every clamped pair ends up in a device counter (sgdfr_split_saturation_count)
        // (written as !(<=) so that NaN counts too: v_med3 would silently turn it into a finite value)
        sat += (!(fabsf(a) <= SPLIT_F16_MAX) || !(fabsf(b) <= SPLIT_F16_MAX)) ? 1u : 0u;
        a = __builtin_amdgcn_fmed3f(a, -SPLIT_F16_MAX, SPLIT_F16_MAX);
        b = __builtin_amdgcn_fmed3f(b, -SPLIT_F16_MAX, SPLIT_F16_MAX);
        f32x2 v = {a, b};
        f16x2 h = __builtin_convertvector(v, f16x2);
        hi = __builtin_bit_cast(unsigned, h);
        f32x2 hf = __builtin_convertvector(h, f32x2);
        f32x2 r = {a - hf[0], b - hf[1]};
        f16x2 l = __builtin_convertvector(r, f16x2);
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        f32x2 v = {a, b};
        bf16x2 h = __builtin_convertvector(v, bf16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
        f32x2 r = {a - ha, b - hb};
        bf16x2 l = __builtin_convertvector(r, bf16x2);
        lo = __builtin_bit_cast(unsigned, l);
    }
}

// `word`: the caller's saturation word (one per generator / per consumer of the result, sgdfr.h "saturation words"); null = the
// device-wide legacy counter read by sgdfr_split_saturation_count.
__device__ __forceinline__ void split_flush_saturation(unsigned sat, unsigned* word) {
    if (__builtin_expect(sat != 0, 0)) atomicAdd(word ? word : &g_split_saturated, sat);
}

// MODE PLAIN3: y = conv3x3(x*s) * d (+ noise, bias, activation).  MODE UP3: stride-2 transposed conv into the four
// output-parity planes T[B,Cout,4,H+1,W+1] (same contract as modconv.hip): one "pixel" is a super-pixel of the padded
// flat space, each of the 9 taps feeds the accumulator set of its parity phase.
// MODE DOWN3 (XIN only): the adjoint of UP3 = dL/d(x*s) of the transposed conv, a stride-2 3x3 conv over the gradient's
// four parity planes: gx[a,b] = sum_taps W[ky,kx]^T gT[phase(ky,kx)][a + (ky==2), b + (kx==2)].  Output "pixels" are
// positions of the same padded flat space as UP3 (row H / column W are dead outputs).  The K loop walks (16-channel block,
// phase) pairs: each pair stages its own plane slice (the input is the phase-major split form written by
// planes_to_split_kernel) and runs only the taps of that phase (4, 2, 2, 1 of the 9).
// NSS: barrier-delimited sub-stages per 16-channel block: 3 = one kernel row (3 taps) each, 1 = all 9 taps.
// XIN: the input is already in the kernel's own split form ("XS": x * s * range shift as 16-bit hi/lo pairs,
// [B][Cin/8][hi,lo][H*W][8]), written by the producer; staging is then a pure global->LDS DMA (no registers, no VALU).
template <int MODE, int ET_, int WM, int WN, int MI, int NI, int NEX, int NSS, bool XIN = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void split_mfma_kernel(SplitParams p) {
    // ET_ = SGDFR_SPLIT_FP16F8 (transposed conv, all nine taps per stage, pre-split input only): the fp16 arithmetic with BOTH cross
    // terms of two taps in one e4m3 MFMA (common.h); everything else of the kernel is the fp16 arithmetic's
    constexpr bool F8 = (ET_ == SGDFR_SPLIT_FP16F8);
    constexpr int ET = F8 ? SGDFR_SPLIT_FP16 : ET_;
    static_assert(!F8 || (XIN && ((MODE == SGDFR_MODE_UP3 && NSS == 1 && MI == 1) || (MODE == SGDFR_MODE_PLAIN3 && NSS == 3 && NI == 2 && WM * WN == 4))),
                  "fp8 cross terms: the deep transposed plan and the 4-wave plain plan only");
    constexpr int NW = WM * WN;            // 8 waves, one block per CU (4-wave blocks, two per CU, measured slower: more
                                           // halo staging and 1.0 ds_read per MFMA)
    constexpr int NTHR = NW * 64;
    constexpr int NT = WM * MI * 32, PT = WN * NI * 32;
    constexpr bool UPF = (MODE == SGDFR_MODE_UPF);
    constexpr bool UP = (MODE == SGDFR_MODE_UP3) || UPF, DOWN = (MODE == SGDFR_MODE_DOWN3);
    static_assert(!UPF || (XIN && NSS == 1 && MI == 1 && NI == 2 && NW == 8), "the fused blur epilogue is written for the 64 x 256 deep plan");
    constexpr int PH = UP ? 4 : 1;
    static_assert(!DOWN || (XIN && NSS == 1), "DOWN3 stages pre-split planes, one (channel block, phase) pair per stage");
    static_assert(NW == 8 || NW == 4, "4 or 8 waves per block");
    static_assert(NT % 64 == 0 && (NSS == 1 || NSS == 3), "cout tiles are packed 64 wide");
    constexpr int RPS = 3 / NSS;                          // kernel rows per sub-stage
    constexpr int WT = NT / 64;                           // 64-cout pack tiles per block
    constexpr int WROW64 = 64 * 192;                      // one kernel row of one pack tile: [3 kx][2 part][2 k-half][64][8] x 16 bit
    constexpr int WTAP = 4096;                            // one tap of one pack tile: [2 part][2 k-half][64][8] x 16 bit
    constexpr int WSLOT_TAPS = DOWN ? 4 : RPS * 3;        // taps a ring slot holds (DOWN3: the largest phase)
    constexpr int WTILE_BYTES = WSLOT_TAPS * WTAP;
    constexpr int WROW_BYTES = WT * WTILE_BYTES;          // weight bytes of one sub-stage: [pack tile][tap][part][k-half][64][8]
    constexpr int WCHUNKS = WROW_BYTES / 1024;            // 64-lane x 16-byte DMA pieces
    // RING3 (pre-split input, row sub-stages): the weight ring has THREE slots and the slab of sub-stage u+2 is DMA'd while u
    // computes.  Inside a tile that doubles the latency a slab may take; across the tiles of a persistent block it lets the
    // last two sub-stages of a tile stage the first TWO slabs (and the first activation block) of the next tile BEFORE the
    // epilogue's stores join the queue, so the next tile's first sub-stage has nothing to wait for -- loads and stores
    // retire through the same in-order vmcnt, and with two slots the first wait of the next tile drained the stores.
    constexpr bool RING3 = XIN && !DOWN && NSS == 3 && NI <= 2 && NW == 8;      // (the 128 x 512 tiles: two slots, see kPlanPlainXL;
                                                                               //  4-wave blocks run two per CU: 80 KB each, two slots)
    constexpr int NWS = RING3 ? 3 : 2;                    // weight ring slots
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xbuf_bytes = 64 * p.xs;                    // [2 part][2 k-half][xs][8] bf16
    unsigned char* const xb0 = smem;
    unsigned char* const wb0 = smem + 2 * xbuf_bytes;
    float* const ls = reinterpret_cast<float*>(wb0 + NWS * WROW_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // F8: the cross-term MFMA's constant exponent 2^(EW - 7) as an E8M0 scale (common.h)
    const int f8_sa = F8 ? 127 + ws_f8_wexp(*p.f8_max) - WS_F8_XLO : 127;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W;
    const int HWin = DOWN ? p.R * p.P : HW;       // positions per channel of the staged tensor

    // Persistent blocks: the grid is one block per CU (or fewer); block b works on tiles base + map(b) of every round of
    // gridDim.x tiles.  map() keeps the tiles of one XCD (blockIdx & 7) contiguous, so neighbours share halos and weights in L2.
    // (only the plain conv is ever launched persistent, see launch_split; the fp32-input variants have no registers to spare)
    constexpr bool PERSIST = XIN && NW == 8 && (MODE == SGDFR_MODE_PLAIN3 || (UP && RING3));
    auto lid_of = [&](int base) -> int {        // tile of this block in the round starting at `base`, -1: none
        if (base >= p.total_blocks) return -1;
        const int nblk = PERSIST ? min((int)gridDim.x, p.total_blocks - base) : (int)gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        if ((int)blockIdx.x >= nblk) return -1;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        return p.lid0 + base + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    };
    struct TileOrg { int ks, ct, pt, q0, img0, row0, col0; };
    auto tile_org = [&](int lid) -> TileOrg {
        TileOrg t;
        const int tiles_per_slice = p.n_pix_tiles * p.n_cout_tiles;
        t.ks = fdiv(lid, p.fd_tps);                   // K slice of this block (0 when ksplit == 1)
        const int lt = lid - t.ks * tiles_per_slice;
        t.ct = fdiv(lt, p.fd_npt);
        t.pt = lt - t.ct * p.n_pix_tiles;
        t.q0 = 0; t.row0 = 0; t.col0 = 0;
        if (p.patch) {
            const int per_img = p.tiles_x * p.tiles_y;
            t.img0 = fdiv(t.pt, p.fd_per_img);
            const int rem = t.pt - t.img0 * per_img;
            const int ty = fdiv(rem, p.fd_tiles_x), tx = rem - ty * p.tiles_x;
            t.row0 = ty * p.tstep_r + p.torg;
            t.col0 = tx * p.tstep_c + p.torg;
        } else if (UP || DOWN) {
            t.q0 = t.pt * PT;                       // super-pixels ARE positions of the padded flat space
            t.img0 = fdiv(t.q0, UP ? p.fd_rps : p.fd_RP);
        } else {
            const int p0 = t.pt * PT;
            t.img0 = fdiv(p0, p.fd_HW);
            const int rem = p0 - t.img0 * HW;
            const int a = fdiv(rem, p.fd_W), b = rem - a * p.W;
            t.q0 = (t.img0 * p.R + a + 1) * p.P + b + 1 - p.P - 1;
        }
        return t;
    };
    // XIN staging item e of this thread for a tile: byte address of the hi chunk of channel group (cb = 0, h);
    // -1: zero page; -2: no item (skip the DMA)
    auto xin_addr = [&](const TileOrg& t, int e) -> int64_t {
        const int i = tid + e * NTHR;
        const int h = fdiv(i, p.fd_xs);
        const int j = i - h * p.xs;
        if (!(h < 2 && j < p.xs)) return -2;
        bool ok = j < p.xlen;
        int img, pix;
        if (p.patch) {
            const int sg = fdiv(j, p.fd_seglen), cc = j - sg * p.seglen;
            const int row = t.row0 - 1 + sg, col = t.col0 - 1 + cc;
            img = t.img0;
            ok = ok && row >= 0 && row < p.H && col >= 0 && col < p.W && img < p.B;
            pix = row * p.W + col;
        } else if (DOWN) {        // the planes ARE the flat space
            const int q = t.q0 + j;
            img = fdiv(q, p.fd_RP);        // (DOWN: HWin = R*P)
            pix = q - img * HWin;
            ok = ok && img < p.B;
        } else if (UP) {          // position r of image img = grid point (r / P, r % P); r >= R*P is the stride padding
            const int q = t.q0 + j;
            img = fdiv(q, p.fd_rps);
            const int r = q - img * p.rps;
            const int pr = fdiv(r, p.fd_P), pc = r - pr * p.P;
            ok = ok && r < p.R * p.P && pc >= 1 && pr >= 1 && img < p.B;
            pix = (pr - 1) * p.W + (pc - 1);
        } else {
            const int q = t.q0 + j;
            const int qq = q < 0 ? 0 : q;                      // (q < 0 only before the first image: masked by `ok` below)
            const int pir = fdiv(qq, p.fd_P), pc = qq - pir * p.P;
            img = fdiv(pir, p.fd_R);
            const int pr = pir - img * p.R;
            ok = ok && pc >= 1 && pr >= 1 && img < p.B && q >= 0;
            pix = (pr - 1) * p.W + (pc - 1);
        }
        return ok ? ((((int64_t)img * (p.Cin / 8) + h) * 2) * HWin + pix) * 16 : -1;
    };
    // DMA issue is not free: a wave sits ~100-200 clocks in every global_load_lds while the CU's address path takes its 1 KB,
    // and a sub-stage issues 32-40 KB.  With all eight waves issuing at the top of a sub-stage the matrix cores idled for
    // ~1000 of its ~3600 clocks (s_memtime trace, scripts/tile_trace.py).  So the two waves of a SIMD issue at different
    // times: waves 0-3 at the top, waves 4-7 ("late") in the middle of their MFMA stream -- one always has MFMAs to run.
    const bool late = XIN && !DOWN && NW == 8 && wave >= 4 && !(p.stagger & 4);
    // activation DMA instructions this wave issues in sub-stage 0 / 1 of a channel block (RING3; see issue_x: slots beyond 2*xs are skipped)
    int nxw[2] = {0, 0};
    if (RING3) {
#pragma unroll
        for (int e = 0; e < NEX; ++e)
            if (wave * 64 + e * NTHR < 2 * p.xs) nxw[e & 1] += 2;
    }
    unsigned sat = 0;              // fp16 operand pairs this thread clamped
    int base = 0;
    int xsel = 0, wsel = 0;        // x buffer of the current channel block / weight slot of the current sub-stage
    bool prefetched = false;       // this tile's first channel block and weight slab were staged by the previous tile
#ifdef SGDFR_SPLIT_PROBE
    const bool trace_on = (p.dbg & 64) && (int)blockIdx.x == ((p.dbg >> 8) & 0xff) && tid == 64 * ((p.dbg >> 16) & 7);
    int ntrace = 0;
#endif
    do {
    SPLIT_TRACE(1);
    const int lid = lid_of(base);
    if (lid < 0) break;
    const int lid_n = PERSIST ? lid_of(base + gridDim.x) : -1;
    // (tiles over many small images put their epilogue tables into the dead staging buffers: nothing may be staged ahead)
    const bool has_next = PERSIST && lid_n >= 0 && p.simgs <= 2;
    // First-round desynchronisation: equal blocks started together reach their store phase together and share the HBM
    // write bandwidth (one block per CU: nothing else hides it).  Spreading the starts of the first round over about one
    // block time lets every later round store while other CUs compute.  p.desync = block-time estimate in 4096-clock units.
    if (p.desync > 0 && blockIdx.x < 256 && base == 0) {
        const int slot = (int)((blockIdx.x * 2654435761u) >> 24);          // 0..255, scrambled
        const int n_sleep = (slot * p.desync) >> 8;
        for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(64);
    }
    const TileOrg T = tile_org(lid), Tn = has_next ? tile_org(lid_n) : T;
    const int ks = T.ks, ct = T.ct, pt = T.pt, q0 = T.q0, img0 = T.img0, row0 = T.row0, col0 = T.col0;
    const int n0 = ct * NT;

    // ---- epilogue coefficient tables.  Loads and stores share the in-order vmcnt counter: a per-output `d` / bias load
    // between two stores would make the wave wait for the previous store's HBM round trip, so every coefficient goes
    // through LDS (lgkmcnt) and the noise values into registers.  Tiles inside <= 2 images fill their tables HERE, in a
    // dedicated LDS region beside the style table (no barrier pair and no exposed global-load latency between the K loop
    // and the stores); tiles spanning many small images (4x4, 8x8) fill them after the loop in the dead staging buffers.
    const bool whole = p.ksplit == 1;     // K slices only scale by d; noise / bias / activation follow the reduction
    const bool fuse_rgb = !UP && !DOWN && whole && p.rgb_part != nullptr;      // fused ToRGB partial sums (PLAIN3)
    const bool early = p.simgs <= 2;
    // (DOWN3 has no style table: its input arrives modulated)
    constexpr bool SLIM = XIN && NW == 4;       // (split_lds_bytes(slim): pre-split 4-wave plans carry no style table either)
    float* const dl = early ? ls + ((DOWN || SLIM) ? 0 : ((p.simgs * p.Cin + 3) & ~3)) : reinterpret_cast<float*>(smem);   // [simgs][NT]  d * output scale
    float* const bl = dl + p.simgs * NT;                        // [NT]            bias
    float* const cw = bl + NT;                                  // [simgs][NT][4]  ToRGB coefficients
    float* const red = cw + p.simgs * NT * 4;                   // [WM][PT][3]     ToRGB cross-wave reduce
    float* const sn = red + WM * PT * 3;                        // [simgs][NT]     next layer's style * range shift (xs_out)
    const bool emit_xs = (!UP || UPF) && !DOWN && whole && p.xs_out != nullptr;
    // An entry's global loads and its LDS writes are separate steps: tiles inside <= 2 images (one entry per thread at most)
    // load at the top of the tile, so the round trip is hidden behind the descriptor arithmetic, and write in the prologue.
    struct TabVals { float dv, bv, sv, rv, w0, w1, w2; };
    auto tables_load = [&](int e) -> TabVals {
        const int m = e / NT, c = e - m * NT;
        const bool cok = n0 + c < p.Cout;              // Cout = 32 (mod 64): the upper half of the last 64-cout tile is padding
        const bool in = img0 + m < p.B && cok;
        const int64_t bc = (int64_t)(img0 + m) * p.Cout + n0 + c;
        TabVals t;
        t.dv = (p.d && in) ? p.d[bc] : 1.f;
        t.bv = (whole && p.bias && cok) ? p.bias[n0 + c] : 0.f;
        t.sv = (emit_xs && in) ? p.s_next[bc] : 0.f;
        // rgb[j] = sum_co y[co] * w_rgb[j][co] * s_rgb[b][co] / sqrt(Cout) over this block's couts
        t.rv = (fuse_rgb && in) ? p.rgb_s[bc] : 0.f;
        t.w0 = (fuse_rgb && cok) ? p.rgb_w[n0 + c] : 0.f;
        t.w1 = (fuse_rgb && cok) ? p.rgb_w[p.Cout + n0 + c] : 0.f;
        t.w2 = (fuse_rgb && cok) ? p.rgb_w[2 * p.Cout + n0 + c] : 0.f;
        return t;
    };
    auto tables_store = [&](int e, const TabVals& t) {
        const float oscale = (ET == SGDFR_SPLIT_FP16) ? SPLIT_F16_OUT : 1.f;
        const float xsc = (ET == SGDFR_SPLIT_FP16) ? SPLIT_F16_XSCALE : 1.f;
        const float rs = rsqrtf((float)p.Cout);
        const int m = e / NT, c = e - m * NT;
        dl[e] = t.dv * oscale;
        if (m == 0) bl[c] = t.bv;
        if (emit_xs) sn[e] = t.sv * xsc;
        if (fuse_rgb) *reinterpret_cast<float4*>(cw + 4 * e) = make_float4(t.w0 * (t.rv * rs), t.w1 * (t.rv * rs), t.w2 * (t.rv * rs), 0.f);
    };
    auto fill_tables = [&]() {
        // one pass, every global load of an entry issued before the first LDS write (one exposed latency, not four)
        for (int e = tid; e < p.simgs * NT; e += NTHR) tables_store(e, tables_load(e));
    };
    const bool tv_has = early && tid < p.simgs * NT;      // (simgs <= 2, NT <= 128: at most one entry per thread)
    TabVals tv{};
    if (tv_has) tv = tables_load(tid);

    // ---- this lane's two output pixels: position inside the staged range, and where they are stored
    int boff[NI];
    int64_t ybase[NI];    // (img*Cout)*HW + rem (UP3: (img*Cout*4)*RP + rem), or -1 when the pixel does not exist
    int64_t nzoff[NI];
    int dimg[NI];
    const int pitch = p.patch ? p.seglen : p.P;
    const int RP = p.R * p.P;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int l = (wn * NI + n) * 32 + l31;
        if (UPF) {        // super-pixel (row0 + r, col0 + c) of the TR x TC patch; lanes beyond TR*TC repeat the last one (masked later)
            const int lc = min(l, p.TR * p.TC - 1);
            const int r = fdiv(lc, p.fd_TC), c = lc - r * p.TC;
            boff[n] = r * p.seglen + c;              // staged position of input pixel (a-1, b-1)
            ybase[n] = 0; nzoff[n] = 0; dimg[n] = img0;
        } else if (UP) {
            int64_t pix = (int64_t)pt * PT + l;
            const bool ok = pix < p.total_pix;
            if (!ok) pix = p.total_pix - 1;
            const int img = fdiv((int)pix, p.fd_rps);
            const int rem = (int)(pix - (int64_t)img * p.rps);
            boff[n] = l;
            ybase[n] = (ok && rem < RP) ? (int64_t)img * p.Cout * 4 * p.rps + rem : -1;      // (rem >= R*P: stride padding)
            nzoff[n] = 0;
            dimg[n] = img;
        } else if (DOWN) {        // position (a, b) of the (H+1) x (W+1) grid -> output pixel (a, b) when a < H and b < W
            int64_t pix = (int64_t)pt * PT + l;
            bool ok = pix < p.total_pix;
            if (!ok) pix = p.total_pix - 1;
            const int img = fdiv((int)pix, p.fd_RP);
            const int rem = (int)(pix - (int64_t)img * RP);
            const int a = fdiv(rem, p.fd_P), b = rem - a * p.P;
            ok = ok && a < p.H && b < p.W;
            boff[n] = l;
            ybase[n] = ok ? (int64_t)img * p.Cout * HW + a * p.W + b : -1;
            nzoff[n] = 0;
            dimg[n] = img;
        } else if (p.patch) {
            const int r = fdiv(l, p.fd_TC), c = l - r * p.TC;
            boff[n] = (r + 1) * p.seglen + c + 1;
            const int rem = (row0 + r) * p.W + col0 + c;
            const bool ok = img0 < p.B;
            ybase[n] = ok ? (int64_t)img0 * p.Cout * HW + rem : -1;
            nzoff[n] = (int64_t)img0 * p.noise_bstride + rem;
            dimg[n] = img0;
        } else {
            int64_t pix = (int64_t)pt * PT + l;
            const bool ok = pix < p.total_pix;
            if (!ok) pix = p.total_pix - 1;
            const int img = fdiv((int)pix, p.fd_HW);
            const int rem = (int)(pix - (int64_t)img * HW);
            const int a = fdiv(rem, p.fd_W), b = rem - a * p.W;
            boff[n] = (img * p.R + a + 1) * p.P + b + 1 - q0;
            ybase[n] = ok ? (int64_t)img * p.Cout * HW + rem : -1;
            nzoff[n] = (int64_t)img * p.noise_bstride + rem;
            dimg[n] = img;
        }
        boff[n] += hi * p.xs;
    }

    // ---- staging descriptors: item = (position j, k-half h) -> 8 channels -> one 16-byte hi and one 16-byte lo chunk
    const float* xsrc[NEX];
    int64_t xs16[NEX];    // XIN only
    int soff[NEX];        // float offset into the LDS style table, -1: nothing to write (beyond the range)
    int ldst[NEX];        // byte offset inside an x buffer
#pragma unroll
    for (int e = 0; e < NEX; ++e) {
        if (XIN) {
            xs16[e] = xin_addr(T, e);
            xsrc[e] = nullptr; soff[e] = -1; ldst[e] = 0;
            continue;
        }
        const int i = tid + e * NTHR;
        const int h = fdiv(i, p.fd_xs);
        const int j = i - h * p.xs;
        const bool in_range = (h < 2) && (j < p.xlen);
        bool ok;
        int img, pix;
        if (p.patch) {
            const int sg = fdiv(j, p.fd_seglen), cc = j - sg * p.seglen;
            const int row = row0 - 1 + sg, col = col0 - 1 + cc;
            img = img0;
            ok = in_range && row >= 0 && row < p.H && col >= 0 && col < p.W && img < p.B;
            pix = row * p.W + col;
        } else if (UP) {
            const int q = q0 + j;
            img = fdiv(q, p.fd_rps);
            const int r = q - img * p.rps;
            const int pr = fdiv(r, p.fd_P), pc = r - pr * p.P;
            ok = in_range && r < p.R * p.P && pc >= 1 && pr >= 1 && img < p.B;
            pix = (pr - 1) * p.W + (pc - 1);
        } else {
            const int q = q0 + j;
            const int qq = q < 0 ? 0 : q;                      // (q < 0 only before the first image: masked by `ok` below)
            const int pir = fdiv(qq, p.fd_P), pc = qq - pir * p.P;
            img = fdiv(pir, p.fd_R);
            const int pr = pir - img * p.R;
            ok = in_range && pc >= 1 && pr >= 1 && img < p.B && q >= 0;
            pix = (pr - 1) * p.W + (pc - 1);
        }
        xsrc[e] = ok ? p.x + (int64_t)img * p.x_bstride + (int64_t)(8 * h) * HW + pix : nullptr;
        soff[e] = !(h < 2 && j < p.xs) ? -1 : ok ? (img - img0) * p.Cin + 8 * h : 0;
        ldst[e] = (h * p.xs + j) * 16;
        xs16[e] = -2;
    }

    SPLIT_TRACE(2);
    f32x16 acc[PH][MI][NI];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][m][n][r] = 0.f;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    float xr[NEX][8];

    auto load_x = [&](int e, int cb) {
        const float* src = xsrc[e] ? xsrc[e] + (int64_t)cb * SPLIT_CB * HW : p.zeros;
        const int64_t cs = xsrc[e] ? HW : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[e][c] = src[c * cs];
    };
    auto convert_store = [&](int e, int cb, unsigned char* xb) {
        if (soff[e] < 0) return;
        const float* sp = ls + soff[e] + cb * SPLIT_CB;
        const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
        const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        uint4 vh, vl;
        unsigned* ph = reinterpret_cast<unsigned*>(&vh);
        unsigned* pl = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
        for (int c = 0; c < 4; ++c) split_pair<ET>(xr[e][2 * c] * sv[2 * c], xr[e][2 * c + 1] * sv[2 * c + 1], ph[c], pl[c], sat);
        *reinterpret_cast<uint4*>(xb + ldst[e]) = vh;
        *reinterpret_cast<uint4*>(xb + 32 * p.xs + ldst[e]) = vl;
    };
    const int ncb_all = p.Cin / SPLIT_CB;
#ifdef SGDFR_SPLIT_PROBE      // scripts/tile_probe.py: 1 = one channel block only, 4 = no K loop, 2 = no epilogue
    const int cb0 = (int)((int64_t)ncb_all * ks / p.ksplit);
    const int ncb = (p.dbg & 4) ? cb0 : (p.dbg & 1) ? cb0 + 1 : (int)((int64_t)ncb_all * (ks + 1) / p.ksplit);
#else
    const int cb0 = (int)((int64_t)ncb_all * ks / p.ksplit), ncb = (int)((int64_t)ncb_all * (ks + 1) / p.ksplit);
#endif
    // channel blocks of the weight pack per 64-cout tile (DOWN3: the loop index is (block, phase); the pack holds 9 taps per block)
    const int wcb = DOWN ? ncb_all / 4 : ncb_all;
    const unsigned char* const wglb = reinterpret_cast<const unsigned char*>(p.wsp) + (int64_t)ct * WT * wcb * 9 * WTAP;
    const unsigned char* const wglb_n = reinterpret_cast<const unsigned char*>(p.wsp) + (int64_t)Tn.ct * WT * wcb * 9 * WTAP;
    // DOWN3 stage u = (channel block u >> 2, phase u & 3): plane slice ph * C/16 + block of the phase-major input, taps
    // [first, first + count) of the block's 9 (pack order: phase 0's four taps, 1's two, 2's two, 3's one)
    auto chan_block = [&](int u) { return DOWN ? (u & 3) * wcb + (u >> 2) : u; };
    const int cb0_n = (int)((int64_t)ncb_all * Tn.ks / p.ksplit);
    constexpr int WV = (WCHUNKS + NW - 1) / NW; // DMA pieces per wave and sub-stage (every wave issues exactly WV)
    auto issue_w = [&](const unsigned char* wglb, int u, int slot) {      // sub-stage u = cb*NSS + ss of a cout tile -> ring slot
#ifdef SGDFR_SPLIT_PROBE
        if (p.dbg & 16) return;
#endif
        unsigned char* dst = wb0 + slot * WROW_BYTES;
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int chunk = (wave + v * NW) % WCHUNKS;          // wrap: a duplicate piece rewrites identical bytes
            const int tile = chunk / (WSLOT_TAPS * 4), within = chunk - tile * (WSLOT_TAPS * 4);
            const unsigned char* src;
            if (DOWN) {
                const int ph = u & 3;
                const int first = ph == 0 ? 0 : ph == 1 ? 4 : ph == 2 ? 6 : 8, count = ph == 0 ? 4 : ph == 3 ? 1 : 2;
                if (within >= count * 4) continue;                 // wave-uniform: this phase has fewer taps than the slot
                src = wglb + ((int64_t)tile * wcb * 9 + (int64_t)(u >> 2) * 9 + first) * WTAP + within * 1024;
            } else {
                src = wglb + ((int64_t)tile * wcb * 9 + (int64_t)u * (RPS * 3)) * WTAP + within * 1024;
            }
            __builtin_amdgcn_global_load_lds((glb_void*)(src + lane * 16), (lds_void*)(dst + chunk * 1024), 16, 0, 0);
        }
    };

    // XIN staging: both parts of slot e of channel block cb -> x buffer xb, 16 bytes per lane, LDS image = lane order
    auto issue_x = [&](int64_t addr, int e, int cb, unsigned char* xb) {
        const int i0 = __builtin_amdgcn_readfirstlane(tid - lane + e * NTHR);     // first item of this wave's slot
        if (i0 >= 2 * p.xs) return;                                                // wave-uniform (xs % 64 == 0)
#ifdef SGDFR_SPLIT_PROBE
        if (p.dbg & 8) return;
#endif
        const int h0 = fdiv(i0, p.fd_xs), j0 = i0 - h0 * p.xs;
        const unsigned char* xbase = reinterpret_cast<const unsigned char*>(p.x);
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const unsigned char* src = addr >= 0 ? xbase + addr + ((int64_t)chan_block(cb) * 4 + part) * HWin * 16
                                                    : reinterpret_cast<const unsigned char*>(p.zeros);
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(xb + part * 32 * p.xs + (h0 * p.xs + j0) * 16), 16, 0, 0);
        }
    };

    float nz[NI];
    {
        const float nw = (whole && p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
#pragma unroll
        for (int n = 0; n < NI; ++n) nz[n] = (!UP && whole && p.noise && ybase[n] >= 0) ? nw * p.noise[nzoff[n]] : 0.f;
    }

    // ---- prologue: style table, channel block 0 staged, block 1 in registers, weight row 0 in flight
    if (XIN) {
        // (a persistent block passed the barrier that ends the previous tile: buffers and tables are free)
        if (!prefetched) {
#pragma unroll
            for (int e = 0; e < NEX; ++e) issue_x(xs16[e], e, cb0, xb0 + xsel * xbuf_bytes);
            issue_w(wglb, cb0 * NSS, wsel);
            if (RING3 && cb0 * NSS + 1 < ncb * NSS) issue_w(wglb, cb0 * NSS + 1, (wsel + 1) % NWS);
        }
        if (tv_has) tables_store(tid, tv);      // (loaded at the top of the tile)
        split_wait_vmcnt<0>();
    } else {
        for (int e = tid; e < p.simgs * p.Cin; e += NTHR) {
            const int m = fdiv(e, p.fd_Cin);
            ls[e] = (img0 + m < p.B) ? p.s[(int64_t)(img0 + m) * p.Cin + (e - m * p.Cin)] * (ET == SGDFR_SPLIT_FP16 ? SPLIT_F16_XSCALE : 1.f) : 0.f;
        }
        if (tv_has) tables_store(tid, tv);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NEX; ++e) load_x(e, cb0);
#pragma unroll
        for (int e = 0; e < NEX; ++e) convert_store(e, cb0, xb0 + xsel * xbuf_bytes);
        issue_w(wglb, cb0 * NSS, wsel);
        if (ncb > cb0 + 1) {
#pragma unroll
            for (int e = 0; e < NEX; ++e) load_x(e, cb0 + 1);
            split_wait_vmcnt<NEX * 8>();
        } else {
            split_wait_vmcnt<0>();
        }
    }
    SPLIT_TRACE(8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SPLIT_TRACE(3);

    // A fragment (m) of kernel row r (inside the sub-stage), tap kx, part: aoff[m] + r*WROW64 + (kx*2+part)*2048
    int aoff[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) {
        const int col = wm * (MI * 32) + m * 32;
        aoff[m] = (col / 64) * WTILE_BYTES + (hi * 64 + (col % 64) + l31) * 16;
    }
    const bool stagger = !XIN && NW == 8 && ((p.stagger & 3) == 1 ? (wave >= 4) : (p.stagger & 3) == 2 ? (wave & 1) : false);
    for (int cb = cb0; cb < ncb; ++cb) {
        const unsigned char* xcur = xb0 + xsel * xbuf_bytes;
        unsigned char* xnext = xb0 + (xsel ^ 1) * xbuf_bytes;
        const bool conv_next = cb + 1 < ncb, load_next = cb + 2 < ncb;
        // the last channel block of a persistent block's tile stages the first block (and weight slab) of its next tile
        const bool pre_next = XIN && !conv_next && has_next;
        // F8: lo fragments that wait for their partner -- transposed conv: tap (0, 1) for tap (2, 1) of the same phase; plain conv:
        // a kernel row's third tap for the next row's (rows are sub-stages: across a barrier, in registers)
        frag128 f8_ca[MI], f8_cb[NI];
#pragma unroll
        for (int ss = 0; ss < NSS; ++ss) {
            const int u = cb * NSS + ss;
            const bool more_w = u + 1 < ncb * NSS;
#ifdef SGDFR_SPLIT_PROBE
            if (p.dbg & 128) SPLIT_TRACE(16 + (u & 15));
#endif
            auto issue_w_top = [&]() {
                if (RING3) {
                    // (the activation pieces of this sub-stage are issued first, see stage_part below, then the slab of u+2: the
                    // counted wait at the end of the sub-stage leaves exactly this sub-stage's pieces in flight)
                } else if (more_w) issue_w(wglb, u + 1, wsel ^ 1);
                else if (XIN && has_next) issue_w(wglb_n, cb0_n * NSS, wsel ^ 1);
            };
            if (!late) issue_w_top();
            __builtin_amdgcn_sched_barrier(0);
            constexpr int kSlots[3] = {(NEX + NSS - 1) / NSS, NSS == 3 ? (NEX + 1) / 3 : 0, NSS == 3 ? NEX / 3 : 0};
            // 1/NSS of the next channel block's activations: registers -> hi/lo -> LDS, then refill the registers
            bool w_ahead = false;      // RING3: did this sub-stage issue a slab?
            int n_issued = 0;          // RING3: DMA instructions this wave issued in this sub-stage
            auto stage_part = [&]() {
                if (RING3) {
                    // all activation pieces of the next block in the first two sub-stages (a whole sub-stage of slack before
                    // the block is needed), then the weight slab two sub-stages ahead -- of the next tile at the end of this one
                    if (ss < 2) {
                        if (conv_next) {
#pragma unroll
                            for (int e = ss; e < NEX; e += 2) issue_x(xs16[e], e, cb + 1, xnext);
                            n_issued = nxw[ss & 1];
                        } else if (pre_next) {
#pragma unroll
                            for (int e = ss; e < NEX; e += 2) issue_x(xin_addr(Tn, e), e, cb0_n, xnext);
                            n_issued = nxw[ss & 1];
                        }
                    }
                    const int u1 = ncb * NSS;
                    w_ahead = true;
                    if (u + 2 < u1) issue_w(wglb, u + 2, (wsel + 2) % NWS);
                    else if (has_next) issue_w(wglb_n, cb0_n * NSS + (u + 2 - u1), (wsel + 2) % NWS);
                    else w_ahead = false;
                    if (w_ahead) n_issued += WV;
                    return;
                }
                if (conv_next) {
#pragma unroll
                    for (int e = ss; e < NEX; e += NSS) {
                        if (XIN) {
                            issue_x(xs16[e], e, cb + 1, xnext);
                        } else {
                            convert_store(e, cb + 1, xnext);
                            if (load_next) load_x(e, cb + 2);
                        }
                    }
                } else if (pre_next) {
#pragma unroll
                    for (int e = ss; e < NEX; e += NSS) issue_x(xin_addr(Tn, e), e, cb0_n, xnext);
                }
            };
            const unsigned char* wslot = wb0 + wsel * WROW_BYTES;
            // `mid`: called once inside the row, between the MFMAs of tap 1 and tap 2 (the late waves' DMA issue)
            // UP3: activation fragments of the kernel row in flight, [column offset 0/1][part][tile].  Kernel rows 0 and 1 read the
            // SAME input row (offset P; row 2 reads offset 0), so a whole-channel-block stage (RPS = 3) fetches them once for both:
            // 16 instead of 24 activation fragment reads per channel block (34 instead of 42 ds_read_b128 per 54 MFMAs).
            frag128 ub[2][2][NI];
            auto mfma_row = [&](int ky, auto&& mid) {
                const unsigned char* wcur = wslot + (ky - ss * RPS) * WROW64;
                if (UP) {
                    // taps (ky, kx): x[a-(ky==2), b-(kx==2)] -> offsets rowoff + {1, 1, 0}; phase = 2*(ky&1) + (kx&1).
                    // The row's 8 activation fragments are read once; weight fragments of tap kx+1 are requested
                    // while the MFMAs of tap kx run.
                    const int rowoff = (ky == 2) ? 0 : pitch;      // (flat space: P; patches: the staged row length)
                    frag128 (&b)[2][2][NI] = ub;
                    frag128 a[2][2][MI];   // [set][part][tile]
                    auto fetch_a = [&](int set, int kx) {
#pragma unroll
                        for (int part = 0; part < 2; ++part)
#pragma unroll
                            for (int m = 0; m < MI; ++m)
                                a[set][part][m] =
                                    *reinterpret_cast<const frag128*>(wcur + aoff[m] + (kx * 2 + part) * 2048);
                    };
                    fetch_a(0, 0);
                    frag128 f8_a0[MI];      // F8: the lo fragment of tap kx = 0, paired with tap kx = 2 (same phase, same row)
                    if (!(RPS == 3 && ky - ss * RPS == 1)) {     // (the second row of a whole-block stage reuses the first row's)
#pragma unroll
                        for (int part = 0; part < 2; ++part)
#pragma unroll
                            for (int o = 1; o >= 0; --o)
#pragma unroll
                                for (int n = 0; n < NI; ++n)
                                    b[o][part][n] =
                                        *reinterpret_cast<const frag128*>(xcur + part * 32 * p.xs + (boff[n] + rowoff + o) * 16);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ph = 2 * (ky & 1) + (kx & 1);
                        const int o = (kx == 2) ? 0 : 1;
                        const int cur = kx & 1;
                        if (F8 && kx == 0) {
#pragma unroll
                            for (int m = 0; m < MI; ++m) f8_a0[m] = a[cur][1][m];      // (set 0 is refilled with tap 2 below)
                        }
                        if (kx < 2) fetch_a(cur ^ 1, kx + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!F8) {
#pragma unroll
                            for (int t = 0; t < 3; ++t)          // product term outermost: dependent MFMAs are MI*NI apart
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[ph][m][n] = split_mfma<ET>(a[cur][t == 2][m], b[o][t == 1][n], acc[ph][m][n]);
                        } else {
                            // main term per tap; cross terms two taps of ONE phase at a time: (ky, 0) + (ky, 2) inside the row,
                            // (0, 1) + (2, 1) across rows (kernel row 0's fragments wait in f8_ca / f8_cb), (1, 1) alone beside zeros
#pragma unroll
                            for (int m = 0; m < MI; ++m)
#pragma unroll
                                for (int n = 0; n < NI; ++n) acc[ph][m][n] = split_mfma<ET>(a[cur][0][m], b[o][0][n], acc[ph][m][n]);
                            if (kx == 2) {
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[ph][m][n] = ws_mfma_f8(f8_a0[m], a[cur][1][m], b[1][1][n], b[0][1][n], acc[ph][m][n], f8_sa);
                            } else if (kx == 1 && ky == 0) {
#pragma unroll
                                for (int m = 0; m < MI; ++m) f8_ca[m] = a[cur][1][m];
#pragma unroll
                                for (int n = 0; n < NI; ++n) f8_cb[n] = b[1][1][n];
                            } else if (kx == 1 && ky == 1) {
                                const frag128 zero = {0, 0, 0, 0};
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[ph][m][n] = ws_mfma_f8(a[cur][1][m], a[cur][1][m], b[1][1][n], zero, acc[ph][m][n], f8_sa);
                            } else if (kx == 1) {
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[ph][m][n] = ws_mfma_f8(f8_ca[m], a[cur][1][m], f8_cb[n], b[1][1][n], acc[ph][m][n], f8_sa);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (kx == 1) mid();
                    }
                } else {
                    // software pipeline over the 3 taps of the row: after the hi*hi MFMAs of tap kx are issued, the
                    // fragments of tap kx+1 are requested (second register set) and land behind the other 8 MFMAs
                    // (NI = 4: 128 accumulator registers leave room for ONE fragment set; the other wave of the SIMD covers the
                    // fetch latency at the top of a tap)
                    constexpr int NSET = NI > 2 ? 1 : 2;
                    frag128 a[NSET][2][MI], b[NSET][2][NI];     // [set][part][tile]
                    auto fetch = [&](int set, int kx) {
                        const int tapoff = (ky - 1) * pitch + (kx - 1);
#pragma unroll
                        for (int part = 0; part < 2; ++part) {
#pragma unroll
                            for (int m = 0; m < MI; ++m)
                                a[set][part][m] =
                                    *reinterpret_cast<const frag128*>(wcur + aoff[m] + (kx * 2 + part) * 2048);
#pragma unroll
                            for (int n = 0; n < NI; ++n)
                                b[set][part][n] =
                                    *reinterpret_cast<const frag128*>(xcur + part * 32 * p.xs + (boff[n] + tapoff) * 16);
                        }
                    };
                    fetch(0, 0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
#ifdef SGDFR_PROBE_NOFETCH      // ablation (wrong results): one fragment fetch per kernel row instead of three -- is the loop LDS-read bound?
                        const int cur = 0;
#else
                        const int cur = kx & (NSET - 1);
                        if (NSET == 1 && kx > 0) fetch(0, kx);
#endif
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < MI; ++m)
#pragma unroll
                            for (int n = 0; n < NI; ++n)
                                acc[0][m][n] =
                                    split_mfma<ET>(a[cur][0][m], b[cur][0][n], acc[0][m][n]);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (F8) {
                            // cross terms two taps at a time: (ky, 0) + (ky, 1) here; the row's third tap pairs with the NEXT row's
                            // third tap (row 0 -> row 1), row 2's stands beside zeros
                            if (kx == 0) {
                                fetch(1, 1);
                            } else if (kx == 1) {
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[0][m][n] = ws_mfma_f8(a[0][1][m], a[1][1][m], b[0][1][n], b[1][1][n], acc[0][m][n], f8_sa);
                                fetch(0, 2);
                            } else if (ky == 0) {
#pragma unroll
                                for (int m = 0; m < MI; ++m) f8_ca[m] = a[0][1][m];
#pragma unroll
                                for (int n = 0; n < NI; ++n) f8_cb[n] = b[0][1][n];
                            } else if (ky == 1) {
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[0][m][n] = ws_mfma_f8(f8_ca[m], a[0][1][m], f8_cb[n], b[0][1][n], acc[0][m][n], f8_sa);
                            } else {
                                const frag128 zero = {0, 0, 0, 0};
#pragma unroll
                                for (int m = 0; m < MI; ++m)
#pragma unroll
                                    for (int n = 0; n < NI; ++n)
                                        acc[0][m][n] = ws_mfma_f8(a[0][1][m], a[0][1][m], b[0][1][n], zero, acc[0][m][n], f8_sa);
                            }
                        } else {
#ifndef SGDFR_PROBE_NOFETCH
                        if (NSET == 2 && kx < 2) fetch(cur ^ 1, kx + 1);
#endif
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 1; t < 3; ++t)
#pragma unroll
                            for (int m = 0; m < MI; ++m)
#pragma unroll
                                for (int n = 0; n < NI; ++n)
                                    acc[0][m][n] = split_mfma<ET>(a[cur][t == 2][m], b[cur][t == 1][n], acc[0][m][n]);
                        }
                        if (kx == 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            mid();
                        }
                    }
                }
            };
            // DOWN3: the taps of one phase, gx[q] += W[tap]^T plane[q + off]; same tap software pipeline as the plain conv
            auto mfma_down = [&](auto ntaps_t, int o0, int o1, int o2, int o3) {
                constexpr int NTAPS = decltype(ntaps_t)::value;
                const int off[4] = {o0, o1, o2, o3};
                frag128 a[2][2][MI], b[2][2][NI];     // [set][part][tile]
                auto fetch = [&](int set, int t) {
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
#pragma unroll
                        for (int m = 0; m < MI; ++m)
                            a[set][part][m] = *reinterpret_cast<const frag128*>(wslot + aoff[m] + t * WTAP + part * 2048);
#pragma unroll
                        for (int n = 0; n < NI; ++n)
                            b[set][part][n] =
                                *reinterpret_cast<const frag128*>(xcur + part * 32 * p.xs + (boff[n] + off[t]) * 16);
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int cur = t & 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MI; ++m)
#pragma unroll
                        for (int n = 0; n < NI; ++n) acc[0][m][n] = split_mfma<ET>(a[cur][0][m], b[cur][0][n], acc[0][m][n]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < NTAPS) fetch(cur ^ 1, t + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tt = 1; tt < 3; ++tt)
#pragma unroll
                        for (int m = 0; m < MI; ++m)
#pragma unroll
                            for (int n = 0; n < NI; ++n)
                                acc[0][m][n] = split_mfma<ET>(a[cur][tt == 2][m], b[cur][tt == 1][n], acc[0][m][n]);
                }
            };
            auto mfma_part = [&]() {
                if (DOWN) {       // taps (ky,kx) of phase 2*(ky&1) + (kx&1) read the plane at q + (ky==2)*P + (kx==2)
                    using std::integral_constant;
                    switch (u & 3) {          // block-uniform
                        case 0: mfma_down(integral_constant<int, 4>{}, 0, 1, p.P, p.P + 1); break;   // (0,0) (0,2) (2,0) (2,2)
                        case 1: mfma_down(integral_constant<int, 2>{}, 0, p.P, 0, 0); break;         // (0,1) (2,1)
                        case 2: mfma_down(integral_constant<int, 2>{}, 0, 1, 0, 0); break;           // (1,0) (1,2)
                        default: mfma_down(integral_constant<int, 1>{}, 0, 0, 0, 0); break;          // (1,1)
                    }
                    return;
                }
                // late waves (pre-split input): this sub-stage's DMA pieces go out between tap 1 and tap 2 of a one-row sub-stage,
                // after the first kernel row of a whole-channel-block stage
#pragma unroll
                for (int r = 0; r < RPS; ++r) {
                    if (RPS == 1) {
                        mfma_row(ss * RPS + r, [&]() {
                            if (late) {
                                issue_w_top();
                                stage_part();
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    } else {
                        mfma_row(ss * RPS + r, [&]() {});
                        if (r == 0 && late) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_w_top();
                            stage_part();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            // fp32 input: the two waves of a SIMD (w, w+4) run the two parts in opposite order, so one converts while the other
            // keeps the matrix core busy
            if (!stagger && !late) stage_part();
#ifdef SGDFR_SPLIT_PROBE
            if (p.dbg & 32) SPLIT_TRACE(32);
#endif
            __builtin_amdgcn_sched_barrier(0);
            mfma_part();
            __builtin_amdgcn_sched_barrier(0);
#ifdef SGDFR_SPLIT_PROBE
            if (p.dbg & 32) SPLIT_TRACE(33);
#endif
            if (stagger) stage_part();
            if (RING3) {
                if (more_w) {
                    // slab u+1 was issued one sub-stage ago: wait for everything older than this sub-stage's slab.  The FIRST
                    // sub-stage of a tile whose operands were staged by the previous tile waits for nothing: slab u+1 landed
                    // before that tile's epilogue, and what is in flight now are its stores.
                    if (!(prefetched && u == cb0 * NSS)) {
#ifdef SGDFR_SPLIT_PROBE
                        if (!(p.dbg & 256))      // ablation (wrong results): never wait for a DMA inside the K loop
#endif
                        split_wait_vmcnt_dyn(n_issued);
                    }
#ifdef SGDFR_SPLIT_PROBE
                    if (p.dbg & 32) SPLIT_TRACE(34);
#endif
                    __builtin_amdgcn_s_barrier();
#ifdef SGDFR_SPLIT_PROBE
                    if (p.dbg & 32) SPLIT_TRACE(35);
#endif
                } else if (has_next) {
                    split_wait_vmcnt<0>();      // the next tile's first stages have landed before this tile's stores join the queue
                }
                wsel = (wsel + 1) % NWS;
                continue;
            }
            if (more_w) {
                if (XIN) {
                    split_wait_vmcnt<0>();
                } else if (conv_next && load_next) {
                    if (ss == 0) split_wait_vmcnt<8 * kSlots[0]>();
                    else if (ss == 1) split_wait_vmcnt<8 * kSlots[1]>();
                    else split_wait_vmcnt<8 * kSlots[2]>();
                } else {
                    split_wait_vmcnt<0>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else if (XIN && has_next) {
                split_wait_vmcnt<0>();      // the next tile's first stage has landed before this tile's stores join the queue
            }
            wsel ^= 1;
        }
        xsel ^= 1;
    }
    prefetched = XIN && has_next && ncb > cb0;
    SPLIT_TRACE(4);

#ifdef SGDFR_SPLIT_PROBE
    if (p.dbg & 2) { if (PERSIST) __syncthreads(); continue; }
#endif
    if constexpr (UPF) {
        // ---- fused FIR epilogue: T = d * acc (the parity planes of MODE_UP3) goes through LDS 16 couts at a time as
        // [cout][position][phase 2*py+px]; every lane then gathers the 3 x 3 super-pixel neighbourhood of its own position =
        // the 5 x 5 window of T rows 2a-1..2a+3 / columns 2b-1..2b+3 around its 2 x 2 output quad, applies the 16 taps in the
        // order of upfirdn2d.hip's blur (bit-identical sums), adds noise + bias, leaky-ReLU, multiplies by the next layer's
        // modulation and splits.  The two lane halves of a position hold channels 4*hi .. 4*hi+3 of one 8-channel chunk for
        // both pixels of a quad row: one v_permlane32_swap per register hands the lower half pixel 2b's whole chunk and the
        // upper half pixel 2b+1's, so every store is one whole 16-byte chunk (full 128-byte lines per instruction).
        // Patch rows / columns 0 and TR-1 / TC-1 are the recomputed halo ring: they only feed their neighbours.
        __syncthreads();                       // every wave is done with the staging buffers: they are the exchange buffer now
        float* const tb = reinterpret_cast<float*>(smem);
        const int TPOS = p.tpos, TCs = p.TC, OW = 2 * p.W;
        const int64_t OHW = (int64_t)4 * p.H * p.W;
        // flipped taps, as the blur kernel; pinned in SGPRs (the compiler otherwise re-loads them from memory inside every
        // output's FMA chain: 16 live VGPRs are more than this epilogue has to spare)
        float kf[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            kf[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.fir[15 - i])));
        const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
        const float f_slope = p.act ? p.slope : 1.f, f_gain = p.act ? p.gain : 1.f;
        int lpos[NI], opix[NI];
        bool lok[NI];
        float lokf[NI];                        // 1 for lanes that own an output quad, 0 for the halo ring / out-of-image lanes
        float nzq[NI][2][2];
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int l = (wn * NI + n) * 32 + l31;
            const int lc = min(l, p.TR * p.TC - 1);
            const int r = fdiv(lc, p.fd_TC), c = lc - r * p.TC;
            const int a = row0 + r, b = col0 + c;
            lok[n] = l < p.TR * p.TC && r >= 1 && r <= p.TR - 2 && c >= 1 && c <= p.TC - 2 && a < p.H && b < p.W && img0 < p.B;
            lokf[n] = lok[n] ? 1.f : 0.f;
            lpos[n] = l + p.TC + 1;
            opix[n] = lok[n] ? (2 * a) * OW + 2 * b : 0;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                nzq[n][rr][0] = nzq[n][rr][1] = 0.f;
                if (p.noise && lok[n]) {
                    const float2 t2 = *reinterpret_cast<const float2*>(p.noise + (int64_t)img0 * p.noise_bstride + opix[n] + rr * OW);
                    nzq[n][rr][0] = t2.x; nzq[n][rr][1] = t2.y;
                }
            }
        }
        // The margins of the exchange buffer (TC + 1 positions either side of the 256) are only ever read by lanes that produce
        // no output; they are zeroed once so those lanes compute on finite values and can be masked by a multiplication
        // (a select on the result makes the compiler wrap every output's FMA chain in its own divergent branch).
        for (int i = tid; i < 16 * 2 * (p.TC + 1); i += NTHR) {
            const int c16 = i / (2 * (p.TC + 1)), m = i - c16 * 2 * (p.TC + 1);
            const int pos = m < p.TC + 1 ? m : PT + m;
            *reinterpret_cast<float4*>(tb + (c16 * TPOS + pos) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int cc = wm * 8 + hi * 4;        // this lane's first cout inside a 16-cout chunk (chunk g = rows 4g .. 4g+3 of every wave)
        const float4* const d4p = reinterpret_cast<const float4*>(dl + wm * 32 + 4 * hi);
        const float4* const b4p = reinterpret_cast<const float4*>(bl + wm * 32 + 4 * hi);
        const float4* const s4p = reinterpret_cast<const float4*>(sn + wm * 32 + 4 * hi);
        unsigned char* const xbase = p.xs_out + ((int64_t)img0 * (p.Cout / 8) + (n0 + wm * 32) / 8) * 2 * OHW * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 dq = d4p[2 * g], bq = b4p[2 * g], sq = s4p[2 * g];
            const float dv[4] = {dq.x, dq.y, dq.z, dq.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, sv[4] = {sq.x, sq.y, sq.z, sq.w};
            if (g > 0) __syncthreads();        // the previous chunk has been read by everybody
#pragma unroll
            for (int n = 0; n < NI; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(tb + ((cc + j) * TPOS + lpos[n]) * 4) =
                        make_float4(acc[0][0][n][4 * g + j] * dv[j], acc[1][0][n][4 * g + j] * dv[j],
                                    acc[2][0][n][4 * g + j] * dv[j], acc[3][0][n][4 * g + j] * dv[j]);
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                unsigned hh[2][2][2], ll[2][2][2];      // [cout pair][quad row][quad column]: split terms of (y * next modulation)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    float v[2][2][2];          // [cout of the pair][quad row][quad column]
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int j = 2 * jp + j2;
                        const float* const base = tb + ((cc + j) * TPOS + lpos[n]) * 4;
                        // win[u][w] = T[2a-1+u][2b-1+w] lives in super-pixel ((u+1)>>1, (w+1)>>1), phase 2*((u+1)&1) + ((w+1)&1).
                        // One super row (three 16-byte reads) at a time; per output the taps still arrive ky-major, kx-minor.
                        float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                        for (int sr = 0; sr < 3; ++sr) {
                            float nb[3][4];
#pragma unroll
                            for (int dc = 0; dc < 3; ++dc) {
                                const float4 t4 = *reinterpret_cast<const float4*>(base + ((sr - 1) * TCs + (dc - 1)) * 4);
                                nb[dc][0] = t4.x; nb[dc][1] = t4.y; nb[dc][2] = t4.z; nb[dc][3] = t4.w;
                            }
#pragma unroll
                            for (int py = (sr == 0 ? 1 : 0); py < 2; ++py) {
                                const int u = 2 * sr - 1 + py;
#pragma unroll
                                for (int rr = 0; rr < 2; ++rr) {
                                    const int ky = u - rr;
                                    if (ky < 0 || ky > 3) continue;
#pragma unroll
                                    for (int kx = 0; kx < 4; ++kx) {
                                        o[rr][0] = fmaf(nb[(kx + 1) >> 1][2 * py + ((kx + 1) & 1)], kf[ky * 4 + kx], o[rr][0]);
                                        o[rr][1] = fmaf(nb[(kx + 2) >> 1][2 * py + ((kx + 2) & 1)], kf[ky * 4 + kx], o[rr][1]);
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                            for (int xo = 0; xo < 2; ++xo) {
                                const float x = lrelu_gain(fmaf(nw, nzq[n][rr][xo], o[rr][xo] + bv[j]), f_slope, f_gain);
                                v[j2][rr][xo] = x * (sv[j] * lokf[n]);      // (x * (s * 1) == x * s; lanes without a quad: finite * 0)
                            }
                    }
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int xo = 0; xo < 2; ++xo) split_pair<ET>(v[0][rr][xo], v[1][rr][xo], hh[jp][rr][xo], ll[jp][rr][xo], sat);
                }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    // lanes 32..63 of the pixel-2b registers <-> lanes 0..31 of the pixel-2b+1 registers
                    auto swap32 = [](unsigned& x0, unsigned& x1) {
                        const auto r2 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
                        x0 = r2[0]; x1 = r2[1];
                    };
                    swap32(hh[0][rr][0], hh[0][rr][1]); swap32(hh[1][rr][0], hh[1][rr][1]);
                    swap32(ll[0][rr][0], ll[0][rr][1]); swap32(ll[1][rr][0], ll[1][rr][1]);
                    if (lok[n]) {
                        unsigned char* const dst = xbase + ((int64_t)g * 2 * OHW + opix[n] + rr * OW + hi) * 16;
                        *reinterpret_cast<uint4*>(dst) = make_uint4(hh[0][rr][0], hh[1][rr][0], hh[0][rr][1], hh[1][rr][1]);
                        *reinterpret_cast<uint4*>(dst + OHW * 16) = make_uint4(ll[0][rr][0], ll[1][rr][0], ll[0][rr][1], ll[1][rr][1]);
                    }
                }
            }
        }
        break;
    }
    // ---- epilogue.  C/D layout of 32x32: column (pixel) = lane&31, row (cout) = (r&3) + 8*(r>>2) + 4*(lane>>5).
    if (!early) {
        __syncthreads();                  // every wave is done with the staging buffers the tables are about to overwrite
        fill_tables();
        __syncthreads();
    }
    float* const yout = p.y + (int64_t)ks * p.split_stride;
    float rgb[NI][3];
#pragma unroll
    for (int n = 0; n < NI; ++n) rgb[n][0] = rgb[n][1] = rgb[n][2] = 0.f;
    // no activation = slope 1, gain 1 (exact), so the element loop has no runtime switch at all: which outputs exist is a
    // compile-time property of the variant dispatched below (a uniform branch per element would fence every LDS read,
    // conversion and store of the 64 elements a lane owns)
    const float e_slope = (whole && p.act) ? p.slope : 1.f, e_gain = (whole && p.act) ? p.gain : 1.f;
    auto epilogue = [&](auto has_y_t, auto emit_xs_t, auto fuse_rgb_t) {
        constexpr bool HAS_Y = decltype(has_y_t)::value, EMIT_XS = decltype(emit_xs_t)::value, FUSE_RGB = decltype(fuse_rgb_t)::value;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            if (ybase[n] < 0) continue;
            const int io = (dimg[n] - img0) * NT + wm * (MI * 32) + 4 * hi;      // this lane's first row in the per-image tables
            if (UP) {
                // (rows 4g .. 4g+3 = couts 8g + 4*hi + j: one 16-byte read of d; the lane-dependent part of the address is in
                // the base pointer, the per-row / per-plane increments are wave-uniform)
                const float4* const d4p = reinterpret_cast<const float4*>(dl + io);
                float* const yp = yout + ybase[n] + (int64_t)(n0 + wm * (MI * 32) + 4 * hi) * 4 * p.rps;
                if (p.plane_il) {      // (block-uniform) one 16-byte store per cout and position: [position][px][py], 512-byte runs per half-wave
                    float* const yq = yp + 3 * (ybase[n] - (int64_t)dimg[n] * p.Cout * 4 * p.rps);
#pragma unroll
                    for (int m = 0; m < MI; ++m) {
                        if (n0 + wm * (MI * 32) + m * 32 >= p.Cout) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 d4 = d4p[m * 8 + 2 * g];
                            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                *reinterpret_cast<float4*>(yq + (int64_t)(m * 32 + 8 * g + j) * 4 * p.rps) =
                                    make_float4(acc[0][m][n][4 * g + j] * dv[j], acc[PH > 2 ? 2 : 0][m][n][4 * g + j] * dv[j],
                                                acc[PH > 1 ? 1 : 0][m][n][4 * g + j] * dv[j], acc[PH > 3 ? 3 : 0][m][n][4 * g + j] * dv[j]);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int m = 0; m < MI; ++m) {
                    if (n0 + wm * (MI * 32) + m * 32 >= p.Cout) continue;     // padding rows of a half-filled cout tile (wave-uniform)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 d4 = d4p[m * 8 + 2 * g];
                        const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float* dst = yp + (int64_t)(m * 32 + 8 * g + j) * 4 * p.rps;
#pragma unroll
                            for (int ph = 0; ph < PH; ++ph) dst[(int64_t)ph * p.rps] = acc[ph][m][n][4 * g + j] * dv[j];
                        }
                    }
                }
                continue;
            }
            // Rows r = 4g .. 4g+3 of an accumulator are 4 consecutive couts (8g + 4*hi + j): their coefficients are ONE 16-byte
            // LDS read per table, and the ToRGB quads of group g+1 are requested before group g is computed -- the element
            // loop used to wait for an LDS round trip every one or two elements (ISA: ds_read2_b32 ... s_waitcnt lgkmcnt(0)).
            // The lane-dependent part of every address lives in a base pointer; what is added per row is wave-uniform.
            const float4* const d4p = reinterpret_cast<const float4*>(dl + io);
            const float4* const b4p = reinterpret_cast<const float4*>(bl + wm * (MI * 32) + 4 * hi);
            const float4* const s4p = reinterpret_cast<const float4*>(sn + io);
            const float4* const cwp = reinterpret_cast<const float4*>(cw) + io;
            const int rem = (int)(ybase[n] - (int64_t)dimg[n] * p.Cout * HW);
            float* const yp = yout + ybase[n] + (int64_t)(n0 + wm * (MI * 32) + 4 * hi) * HW;
            unsigned char* const xp =
                p.xs_out + ((((int64_t)dimg[n] * (p.Cout / 8) + (n0 + wm * (MI * 32)) / 8) * 2) * HW + rem) * 16 + 8 * hi;
#pragma unroll
            for (int m = 0; m < MI; ++m) {
                if (n0 + wm * (MI * 32) + m * 32 >= p.Cout) continue;     // padding rows of a half-filled cout tile (wave-uniform)
                float4 d4[2], b4[2], s4[2], q[2][4];      // [row group parity]: group g+1 is requested before g is computed
                d4[0] = d4p[m * 8];
                b4[0] = b4p[m * 8];
                if (EMIT_XS) s4[0] = s4p[m * 8];
                if (FUSE_RGB) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[0][j] = cwp[m * 32 + j];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g + 1 < 4) {
                        d4[(g + 1) & 1] = d4p[m * 8 + 2 * (g + 1)];
                        b4[(g + 1) & 1] = b4p[m * 8 + 2 * (g + 1)];
                        if (EMIT_XS) s4[(g + 1) & 1] = s4p[m * 8 + 2 * (g + 1)];
                        if (FUSE_RGB) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) q[(g + 1) & 1][j] = cwp[m * 32 + (g + 1) * 8 + j];
                        }
                    }
                    const float4 dq = d4[g & 1], bq = b4[g & 1], sq = s4[g & 1];
                    const float dv[4] = {dq.x, dq.y, dq.z, dq.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = lrelu_gain(acc[0][m][n][4 * g + j] * dv[j] + nz[n] + bv[j], e_slope, e_gain);
                    if (HAS_Y) {      // (no y: only the fused ToRGB / xs_out consume this layer)
#pragma unroll
                        for (int j = 0; j < 4; ++j) yp[(int64_t)(m * 32 + 8 * g + j) * HW] = v[j];
                    }
                    if (EMIT_XS) {    // the 4 rows are half of one 8-channel chunk of this pixel
                        unsigned h01, l01, h23, l23;
                        split_pair<ET>(v[0] * sq.x, v[1] * sq.y, h01, l01, sat);
                        split_pair<ET>(v[2] * sq.z, v[3] * sq.w, h23, l23, sat);
                        unsigned char* dst = xp + (int64_t)(m * 4 + g) * 2 * HW * 16;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(dst + (int64_t)HW * 16) = make_uint2(l01, l23);
                    }
                    if (FUSE_RGB) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            rgb[n][0] = fmaf(v[j], q[g & 1][j].x, rgb[n][0]);
                            rgb[n][1] = fmaf(v[j], q[g & 1][j].y, rgb[n][1]);
                            rgb[n][2] = fmaf(v[j], q[g & 1][j].z, rgb[n][2]);
                        }
                    }
                }
            }
        }
    };
    {
        using yes = std::true_type;
        using no = std::false_type;
        if (UP || DOWN) {
            epilogue(yes{}, no{}, no{});
        } else {
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
    }
    SPLIT_TRACE(5);
    if (fuse_rgb) {      // block-uniform
        // the two lane halves hold different couts of the same pixels; the WM cout-waves of a pixel column meet in LDS
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                rgb[n][j] += __shfl_xor(rgb[n][j], 32, 64);
                if (hi == 0) red[(wm * PT + (wn * NI + n) * 32 + l31) * 3 + j] = rgb[n][j];
            }
        __syncthreads();
        if (wm == 0 && hi == 0) {
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                if (ybase[n] < 0) continue;
                const int rem = (int)(ybase[n] - (int64_t)dimg[n] * p.Cout * HW);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < WM; ++w2) t += red[(w2 * PT + (wn * NI + n) * 32 + l31) * 3 + j];
                    p.rgb_part[(((int64_t)dimg[n] * p.n_cout_tiles + ct) * 3 + j) * HW + rem] = t;
                }
            }
        }
    }
    SPLIT_TRACE(6);
    if (PERSIST) __syncthreads();      // the next tile refills the tables and the staging buffers
    SPLIT_TRACE(7);
    } while (PERSIST && (base += gridDim.x) < p.total_blocks);
    if (ET == SGDFR_SPLIT_FP16) split_flush_saturation(sat, p.sat);
#ifdef SGDFR_SPLIT_PROBE
    if (trace_on) g_split_trace[0] = (unsigned long long)ntrace;
#endif
}

}  // namespace sgdfr
