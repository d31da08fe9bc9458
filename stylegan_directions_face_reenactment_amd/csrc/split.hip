// 3x3 modulated convolution on the 16-bit matrix cores with every fp32 operand split in two 16-bit terms:
//
//     a = a_hi + a_lo,  b = b_hi + b_lo   (hi = round16(v), lo = round16(v - hi); v - hi is exact in fp32)
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi      accumulated in fp32 by v_mfma_f32_32x32x16_{f16,bf16}
//
// Three 16-bit MFMAs replace eight fp32 ones (v_mfma_f32_32x32x2_f32 covers K=2 in 64 cycles, the 16-bit form K=16 in
// 32), so the contraction runs ~5x above the fp32-MFMA roof and the kernel is bound by LDS reads + MFMA issue, stores
// and staging instead.
//   ET = SGDFR_SPLIT_FP16 (default arithmetic of the inference path): 11+11 mantissa bits = fp32-grade products; operands
//        are range-shifted by exact powers of two (x*2^-4, W*2^6, result*2^-2) and |x*s| saturates at 1.04e6.  Measured on
//        the 256x256 generator: 7.7e-6 max-abs vs the fp64 oracle (fp32 MFMA kernels: 9.5e-6).
//   ET = SGDFR_SPLIT_BF16: 8+8 bits, full fp32 range (1.2e-4 on images; contract 1e-3).  Used for dL/dx (gradients have no
//        natural scale).
//
// Same algebra as modconv.hip (y = d * conv(x*s, Wc), batch folded into the pixel dimension), different data path:
//   * GEMM view M = cout, N = pixels, K = (tap, cin); one MFMA = 16 input channels of one tap.  A lane holds 8
//     consecutive channels (16 B): LDS keeps activations as [part hi/lo][k-half][position][8 x 16 bit] and weights as
//     [cout tile of 64][row][tap][part][k-half][cout][8 x 16 bit], so every fragment is one conflict-free ds_read_b128.
//   * block = 8 waves, one block per CU: NT couts x PT pixels from a tiling plan (plain 128x256 / 64x512, wave tile 64x64 =
//     2x2 MFMA tiles x 3 products; transposed conv: 4 parity-phase accumulator sets, wave tile 32x64, 64x256 or 128x128).
//   * K loop over 16-channel blocks in NSS barrier-delimited sub-stages (3 = one kernel row each, 1 = all 9 taps).  The
//     weight slab of a sub-stage (prepacked in LDS order) is DMA'd global->LDS one sub-stage ahead into a 2-slot ring.
//     Activations of the NEXT channel block go to the other x buffer either (fp32 input) converted x*s -> hi/lo from
//     registers loaded a whole channel block earlier, the two waves of a SIMD running {convert, MFMA} in opposite order,
//     or (XIN: the producer already wrote the split form, see to_split_kernel / xs_out / upfirdn2d.hip's blur) by DMA.
//     Fragments of tap k+1 are requested while the MFMAs of tap k run.
//   * pixel tiles: contiguous runs of the padded flat space for W <= 64 (a tile may span images), TR x 128 patches
//     for wider images; layers with too few tiles slice the channel blocks over more blocks (deterministic reduce).
//   * epilogue: coefficients (d, bias, next layer's style, ToRGB weights) come from LDS tables so the wave issues its
//     stores back to back; optional outputs: fp32 activation, the activation in the next conv's split input form, and the
//     partial sums of the 1x1 ToRGB conv that follows the layer -- which of them exist selects a branch-free instantiation
//     of the element loop.  Transposed conv: `plane_stride` pads each parity plane (and the flat position space per image)
//     to whole 128-byte lines, because a store run that straddles two lines costs twice.
//   * layers of many short tiles (plain conv, pre-split input, >= 12 tiles per CU) run as persistent blocks: one block
//     per CU walks its tiles and stages the next tile's first channel block and weight slab while the current tile's last
//     one is in the matrix cores.
//   * mode DOWN3 (backward of the transposed conv) reuses the transposed conv's position space and gather pattern with
//     (channel block, phase) stages over phase-major pre-split planes (see the kernel's own comment).
#include "split_kernel.h"

namespace sgdfr {

// x [B,Cin,HW] fp32 (NCHW) and s [B,Cin] -> XS [B][Cin/8][hi,lo][HW][8]: the split form of x*s (with the fp16 range shift)
// that split_mfma_kernel<..., XIN> stages by DMA.  One thread = one pixel of one 8-channel group.
template <int ET>
__global__ __launch_bounds__(256) void to_split_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                      unsigned char* __restrict__ xs, int B, int Cin, int HW,
                                                      unsigned* __restrict__ sat_word) {
    const int G = Cin / 8;
    const int64_t n = (int64_t)B * G * HW;
    unsigned sat = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int pix = (int)(idx % HW);
        const int64_t bg = idx / HW;
        const int g = (int)(bg % G);
        const int b = (int)(bg / G);
        const float* xp = x + ((int64_t)b * Cin + g * 8) * HW + pix;
        const float* sp = s + (int64_t)b * Cin + g * 8;
        uint4 vh, vl;
        unsigned* ph = reinterpret_cast<unsigned*>(&vh);
        unsigned* pl = reinterpret_cast<unsigned*>(&vl);
        constexpr int ETM = ws_main_et<ET>::value;
        const float sc = (ETM == SGDFR_SPLIT_FP16) ? SPLIT_F16_XSCALE : 1.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            split_pair<ETM>(xp[(int64_t)(2 * c) * HW] * (sp[2 * c] * sc), xp[(int64_t)(2 * c + 1) * HW] * (sp[2 * c + 1] * sc), ph[c], pl[c], sat);
        if (ET == SGDFR_SPLIT_FP16F8) ws_f8_lo_chunk(vh, vl, exp2f((float)WS_F8_XLO), exp2f((float)WS_F8_XHI), false, sat);
        unsigned char* dst = xs + ((bg * 2) * HW + pix) * 16;
        *reinterpret_cast<uint4*>(dst) = vh;
        *reinterpret_cast<uint4*>(dst + (int64_t)HW * 16) = vl;
    }
    if (ET != SGDFR_SPLIT_BF16) split_flush_saturation(sat, sat_word);
}

// gT [B,C,4,RP] fp32 parity planes and d [B,C] (or null) -> the phase-major split form of gT*d that the DOWN3 kernel stages:
// [B][(ph*C + c)/8][hi,lo][RP][8].  One thread = one position of one 8-channel group of one phase.
template <int ET>
__global__ __launch_bounds__(256) void planes_to_split_kernel(const float* __restrict__ gt, const float* __restrict__ d,
                                                             unsigned char* __restrict__ xs, int B, int C, int RP,
                                                             unsigned* __restrict__ sat_word) {
    const int G = C / 8;
    const int64_t n = (int64_t)B * 4 * G * RP;
    unsigned sat = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int pos = (int)(idx % RP);
        int64_t r = idx / RP;
        const int g = (int)(r % G);
        r /= G;
        const int ph = (int)(r % 4);
        const int b = (int)(r / 4);
        const float* xp = gt + (((int64_t)b * C + g * 8) * 4 + ph) * RP + pos;
        const float sc = (ET == SGDFR_SPLIT_FP16) ? SPLIT_F16_XSCALE : 1.f;
        float sv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) sv[c] = (d ? d[(int64_t)b * C + g * 8 + c] : 1.f) * sc;
        uint4 vh, vl;
        unsigned* ph_ = reinterpret_cast<unsigned*>(&vh);
        unsigned* pl_ = reinterpret_cast<unsigned*>(&vl);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            split_pair<ET>(xp[(int64_t)(2 * c) * 4 * RP] * sv[2 * c], xp[(int64_t)(2 * c + 1) * 4 * RP] * sv[2 * c + 1], ph_[c], pl_[c], sat);
        unsigned char* dst = xs + ((((int64_t)b * 4 * G + ph * G + g) * 2) * RP + pos) * 16;
        *reinterpret_cast<uint4*>(dst) = vh;
        *reinterpret_cast<uint4*>(dst + (int64_t)RP * 16) = vl;
    }
    if (ET == SGDFR_SPLIT_FP16) split_flush_saturation(sat, sat_word);
}

// weight [Cout,Cin,3,3] fp32 -> bf16 hi/lo of Wc = weight/sqrt(9 Cin) in the kernel's LDS order:
//   [cout tile][cin block][ky][kx][part][k-half][cout in tile (NT)][8 cin]
// SGDFR_SPLIT_FP16F8: max |w * scale| (as float bits: non-negative floats order like unsigned ints) into the pack's trailer word
__global__ __launch_bounds__(256) void split_absmax_kernel(const float* __restrict__ w, int64_t n, float scale, unsigned* __restrict__ trailer) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i] * scale));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(trailer, __builtin_bit_cast(unsigned, m));
}

__global__ __launch_bounds__(256) void prepack_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                           int Cout, int Cin, int NT, float scale, int et, int transpose_flip,
                                                           unsigned* __restrict__ sat_word) {
    // packed conv: n_out x n_in channels.  transpose_flip 1: the adjoint conv (dL/dx of the plain conv): channels swapped, taps
    // rotated by 180 degrees; 2: the adjoint of the transposed conv (DOWN3): channels swapped, taps as they are but stored
    // phase by phase ((0,0) (0,2) (2,0) (2,2) | (0,1) (2,1) | (1,0) (1,2) | (1,1)); weight stays indexed [Cout][Cin][3][3]
    const int n_out = transpose_flip ? Cin : Cout, n_in = transpose_flip ? Cout : Cin;
    const int64_t n = (int64_t)n_out * n_in * 9;
    const int ncb = n_in / SPLIT_CB;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(idx % 9);
        const int ci = (int)((idx / 9) % n_in);
        const int co = (int)(idx / (9 * (int64_t)n_in));
        const float v = (transpose_flip == 1 ? w[((int64_t)ci * Cin + co) * 9 + (8 - tap)]
                         : transpose_flip == 2 ? w[((int64_t)ci * Cin + co) * 9 + tap] : w[idx]) * scale;   // fp16: scale carries 2^6
        unsigned hp, lp, sat = 0;
        if (et != SGDFR_SPLIT_BF16) split_pair<SGDFR_SPLIT_FP16>(v, 0.f, hp, lp, sat);
        else split_pair<SGDFR_SPLIT_BF16>(v, 0.f, hp, lp, sat);
        split_flush_saturation(sat, sat_word);
        const unsigned hbits = hp & 0xffffu, lbits = lp & 0xffffu;
        const int ctile = co / NT, col = co - ctile * NT, cb = ci / SPLIT_CB, h = (ci % SPLIT_CB) / 8, c8 = ci % 8;
        const int down_pos[9] = {0, 4, 1, 6, 8, 7, 2, 5, 3};      // tap ky*3+kx -> slot in the phase-by-phase order
        const int slot = transpose_flip == 2 ? down_pos[tap] : tap;
        const int64_t base = (((int64_t)ctile * ncb + cb) * 9 + slot) * 2;   // -> [part]
        out[(((base + 0) * 2 + h) * NT + col) * 8 + c8] = (unsigned short)hbits;
        if (et == SGDFR_SPLIT_FP16F8) {      // lo chunk of (cout, 8 channels) as fp8 bytes, ws_f8_half's order: (4 x hi | 4 x lo) per channel half
            const int ew = ws_f8_wexp(*reinterpret_cast<const float*>(out + (int64_t)((n_out + 63) / 64 * 64) * n_in * 9 * 2));
            const float fh = (float)__builtin_bit_cast(_Float16, (unsigned short)hbits), fl = (float)__builtin_bit_cast(_Float16, (unsigned short)lbits);
            const int two = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(fh * exp2f((float)-ew), -448.f, 448.f),
                                                            __builtin_amdgcn_fmed3f(fl * exp2f((float)(11 - ew)), -448.f, 448.f), 0, false);
            unsigned char* const chunk = reinterpret_cast<unsigned char*>(out) + (((base + 1) * 2 + h) * NT + col) * 16 + (c8 >> 2) * 8 + (c8 & 3);
            chunk[0] = (unsigned char)(two & 0xff);
            chunk[4] = (unsigned char)((two >> 8) & 0xff);
        } else {
            out[(((base + 1) * 2 + h) * NT + col) * 8 + c8] = (unsigned short)lbits;
        }
    }
}

}  // namespace sgdfr

using namespace sgdfr;


// ---- tiling plans.  A plan = (NT couts, PT pixels, NSS sub-stages per channel block); `cfg` indexes the kernel
// instantiations in launch_plan().
struct SplitPlan {
    int cfg, nt, pt, nss, nw;
};
static const SplitPlan kPlanPlainWide = {0, 128, 256, 3, 8};    // plain: 128 couts x 256 pixels, row sub-stages
static const SplitPlan kPlanPlainNarrow = {1, 64, 512, 3, 8};   // plain, Cout % 128 != 0: 64 x 512
static const SplitPlan kPlanUpWide = {2, 128, 128, 3, 8};       // transposed: 128 couts x 128 super-pixels, row sub-stages
static const SplitPlan kPlanUpNarrow = {3, 64, 256, 3, 8};      // transposed, Cout % 128 != 0 (or as a fallback): 64 x 256
static const SplitPlan kPlanUpDeep = {4, 64, 256, 1, 8};        // transposed: 64 x 256, all 9 taps between barriers
static const SplitPlan kPlanDown = {5, 128, 256, 1, 8};         // adjoint of the transposed conv: 128 x 256 positions, (block, phase) stages
// plain, pre-split input, >= 2 tiles per CU: 128 couts x 512 pixels, wave tile 64 x 128.  The chip is power-bound inside the K loop
// (DESIGN 4.7): per MFMA this tile DMAs 0.58x the bytes of 128 x 256 (the weight slab is streamed once per 512 pixels) and
// reads 0.75x the LDS fragments (12 ds_read_b128 per 24 MFMAs), and a layer has half as many tile prologues / epilogues.
static const SplitPlan kPlanPlainXL = {6, 128, 512, 3, 8};
// 4-wave blocks, TWO per CU (<= 80 KB of LDS each, pre-split input only), for the plain layers whose K loop is short: 64 couts x
// 256 pixels, the wave tile of the 64 x 512 plan (64 x 64).  A tile of the 64 -> 64 @ 256^2 layer spends 31 % of its time outside
// the K loop (element loop, ToRGB reduce, descriptors, tables); with one 8-wave block per CU every wave leaves the matrix cores
// idle together, with two independent blocks one is in its K loop while the other runs its epilogue.  Same-process A/B at B=64
// (scripts/layer_ab.py): 996 -> 858 us, bit-identical.  (The same arrangement for the transposed conv -- 64 couts x 128
// super-pixels, row sub-stages because all nine taps do not fit 80 KB -- lost 4-11 %: twice the weight and halo DMA per MFMA.)
static const SplitPlan kPlanPlain4 = {8, 64, 256, 3, 4};

// nws: weight ring slots (3 for a pre-split input with row sub-stages, see RING3 in the kernel; 2 otherwise)
static size_t split_lds_bytes(const SplitParams& p, int NT, int nss, bool down = false, int nws = 2, int PT = 256, bool slim = false) {
    const size_t wslot = down ? (size_t)NT * 256 : (size_t)NT * 192 * (3 / nss);       // DOWN3: 4 taps x 64 bytes per cout
    // (slim: the pre-split 4-wave plan -- no style table, the input arrives modulated)
    const size_t loop = 2 * (size_t)64 * p.xs + nws * wslot + ((down || slim) ? 0 : (size_t)((p.simgs * p.Cin + 3) & ~3) * sizeof(float));
    // d, bias, ToRGB coefficient / reduce ([WM][PT][3] = 1536 floats in every plan but the 128 x 512 one), next-style tables
    const size_t epi = ((size_t)p.simgs * NT * 6 + NT + (NT * PT > 128 * 256 ? 1024 : 512) * 3) * sizeof(float);
    if (p.simgs <= 2) return loop + epi;      // tables live beside the style table for the whole kernel
    return loop > epi ? loop : epi;           // tables overwrite the dead staging buffers after the K loop
}

// geometry of the pixel tiling for one plan; returns 0 when the shape cannot use it
static int split_geometry(int B, int Cin, int Cout, int H, int W, int mode, const SplitPlan& plan, SplitParams* out) {
    // (NT = 64 plans also take Cout = 32 mod 64 -- the 512^2 / 1024^2 layers of the ffhq-1024 generator: the pack pads the last
    // cout tile with zero rows, the epilogue skips them; forward modes only)
    const bool half_tile = plan.nt == 64 && Cout % 64 == 32 && mode != SGDFR_MODE_DOWN3;
    if (Cin % SPLIT_CB != 0 || (Cout % plan.nt != 0 && !half_tile) || B < 1) return 0;
    const bool down = mode == SGDFR_MODE_DOWN3;
    if (down) Cin *= 4;          // K channels = (plane channel, phase)
    SplitParams p{};
    const int PT = plan.pt;
    const int HW = H * W;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    p.total_pix = (int64_t)B * HW;
    if ((int64_t)B * p.R * p.P + 4ll * p.P + 8 >= (1ll << 31) || p.total_pix >= (1ll << 31)) return 0;
    if (mode == SGDFR_MODE_UP3 || down) {
        p.patch = 0;
        p.total_pix = (int64_t)B * p.R * p.P;            // super-pixels
        p.xlen = PT + p.P + 2;
        p.simgs = (p.xlen - 1) / (p.R * p.P) + 2;
        p.n_pix_tiles = (int)((p.total_pix + PT - 1) / PT);
        p.rps = p.R * p.P;
    } else if (W > 64) {
        p.patch = 1;
        // patch width: 64 columns (TR = PT / 64 rows) stage 0.85x the positions of 128-wide patches for the same pixels; the
        // 128 x 512 tiles take 32 x 16 patches (three staging slots instead of four).  Same-box A/B at B=64: 128->128@128^2
        // 899 -> 891 us (-> 874 with the 128 x 512 tiles), 64->64@256^2 961 -> 950.
        static const int tc_env = getenv("SGDFR_SPLIT_TC") ? atoi(getenv("SGDFR_SPLIT_TC")) : 0;
        p.TC = tc_env > 0 ? tc_env : ((plan.nt == 128 && plan.pt == 512) || plan.cfg == 8) ? 32 : 64;
        p.TR = PT / p.TC;
        if (W % p.TC != 0 || H % p.TR != 0) return 0;
        p.tiles_x = W / p.TC; p.tiles_y = H / p.TR;
        p.tstep_r = p.TR; p.tstep_c = p.TC; p.torg = 0;
        p.seglen = p.TC + 2;
        p.xlen = (p.TR + 2) * p.seglen;
        p.simgs = 1;
        p.n_pix_tiles = B * p.tiles_x * p.tiles_y;
    } else {
        p.patch = 0;
        int span;   // q distance between the first and the last pixel of a tile
        if (PT % W == 0 && HW % PT == 0) {
            span = (PT / W - 1) * p.P + (W - 1);
            p.simgs = 1;
        } else if (PT % HW == 0) {
            span = (PT / HW - 1) * p.R * p.P + (H - 1) * p.P + (W - 1);
            p.simgs = PT / HW;
        } else {
            return 0;
        }
        p.xlen = span + 2 * p.P + 3;
        p.n_pix_tiles = (int)((p.total_pix + PT - 1) / PT);
    }
    const bool slim = plan.cfg == 8;                      // 4-wave plan of the pre-split input (DMA staging: whole 64-lane pieces)
    p.xs = (plan.nw == 8 || slim) ? (p.xlen + 63) & ~63 : (p.xlen + 7) & ~7;
    p.n_cout_tiles = (Cout + plan.nt - 1) / plan.nt;
    const int nthr = plan.nw * 64;
    if ((2 * p.xs + nthr - 1) / nthr > (mode == SGDFR_MODE_DOWN3 ? 3 : 4)) return 0;                  // staging slots (512-wide transposed conv: 4)
    const size_t lds = split_lds_bytes(p, plan.nt, plan.nss, down, 2, plan.pt, slim);
    if (lds > (plan.nw == 8 ? 160 : 80) * 1024) return 0;
    if (slim && p.simgs > 2) return 0;                    // (their tables live beside the staging buffers)
    if (out) *out = p;
    return 1;
}

// the plan used for a shape (first that fits), or nullptr
static const SplitPlan* split_plan(int B, int Cin, int Cout, int H, int W, int mode, SplitParams* out, bool xin_whole = false) {
    static const int deep = getenv("SGDFR_SPLIT_DEEP") ? atoi(getenv("SGDFR_SPLIT_DEEP")) : 1;
    static const int narrow_first = getenv("SGDFR_SPLIT_UP_NARROW") ? atoi(getenv("SGDFR_SPLIT_UP_NARROW")) : 0;
    const SplitPlan* order[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    if (mode == SGDFR_MODE_UP3) {
        // deep stages need enough blocks per 64-cout tile to fill the chip; small layers keep the row sub-stages + K slices
        const bool big = (int64_t)B * (H + 1) * (W + 1) * (Cout / 64) >= 256 * 256;
        if (deep && big) order[n++] = &kPlanUpDeep;
        if (narrow_first && big) order[n++] = &kPlanUpNarrow;     // row sub-stages + 3-slot ring + persistent blocks (pre-split input)
        if (Cout % 128 == 0) order[n++] = &kPlanUpWide;
        order[n++] = &kPlanUpNarrow;
    } else if (mode == SGDFR_MODE_PLAIN3) {
        static const int xl = getenv("SGDFR_SPLIT_XL") ? atoi(getenv("SGDFR_SPLIT_XL")) : 1;
        // (xin_whole: pre-split input, no K slices -- the only instantiation of the 128 x 512 tile; it keeps the cout tiling
        // of the 128 x 256 plan, so buffers sized from a query without the flag stay right)
        // (same-box A/B at B=64: 512->512@32^2 842 -> 808 us, 256->256@64^2 822 -> 798; with 128-wide patches 128^2 lost 1 % --
        // four staging slots spill -- with 32 x 16 patches it gains 2.8 %)
        if (xl && xin_whole && Cout % 128 == 0 && (int64_t)B * H * W * (Cout / 128) >= 2ll * 256 * 512) order[n++] = &kPlanPlainXL;
        // (SGDFR_SPLIT_P4: 0 = off, n = layers with Cin <= n; read per call so scripts/layer_ab.py can flip it in one process.
        //  Cout % 128 == 0 layers take it only when asked for (n >= 128): their cout tiling changes from 128 to 64 wide, which
        //  sgdfr_modconv2d_split_cout_tiles_xin reports)
        const int p4 = getenv("SGDFR_SPLIT_P4") ? atoi(getenv("SGDFR_SPLIT_P4")) : 64;
        const bool p4_ok = p4 > 0 && xin_whole && Cin <= p4 && (int64_t)B * H * W * ((Cout + 63) / 64) >= 4ll * 512 * 256;
        if (p4_ok && Cout % 128 == 0 && p4 >= 128) order[n++] = &kPlanPlain4;
        if (Cout % 128 == 0) order[n++] = &kPlanPlainWide;
        if (p4_ok) order[n++] = &kPlanPlain4;
        order[n++] = &kPlanPlainNarrow;
    } else if (mode == SGDFR_MODE_DOWN3) {
        order[n++] = &kPlanDown;
    }
    for (int i = 0; i < n; ++i)
        if (split_geometry(B, Cin, Cout, H, W, mode, *order[i], out)) return order[i];
    return nullptr;
}


extern "C" int sgdfr_modconv2d_split_supported(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    return split_plan(B, Cin, Cout, H, W, mode, &p) ? 1 : 0;
}

// SGDFR_SPLIT_FP16F8 (fp8 cross terms) exists for the transposed conv's deep plan (all nine taps of a channel block per stage) and
// the 4-wave plain plan (64 -> 64 @ 256^2 of the bench generator), both with a pre-split input and without K slices
extern "C" int sgdfr_modconv2d_split_f8_ok(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    if (mode != SGDFR_MODE_UP3 && mode != SGDFR_MODE_PLAIN3) return 0;
    const SplitPlan* plan = split_plan(B, Cin, Cout, H, W, mode, &p, true);
    return plan && plan->cfg == (mode == SGDFR_MODE_UP3 ? 4 : 8) ? 1 : 0;
}

extern "C" int sgdfr_modconv2d_split_ksplit_hint(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    if (!split_plan(B, Cin, Cout, H, W, mode, &p)) return 1;
    const int blocks = p.n_pix_tiles * p.n_cout_tiles;
    if (blocks >= 192) return 1;
    // one resident wave of blocks: every extra slice also writes and re-reads a full partial tensor (measured at B=64:
    // target 256 beats 512 by 30 us on each of the 4x4 / 8x8 layers, B=8 5.3k -> 5.6k frames/s)
    static const int target = getenv("SGDFR_SPLIT_KTARGET") ? atoi(getenv("SGDFR_SPLIT_KTARGET")) : 256;
    int s = target / blocks;
    const int max_by_k = (Cin / SPLIT_CB) / 2;  // at least 2 channel blocks per slice
    if (s > max_by_k) s = max_by_k;
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

extern "C" int sgdfr_to_split_f32(const float* x, const float* s, unsigned short* xs, int B, int Cin, int H, int W, int arith,
                                  unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cin % 8 == 0 && H > 0 && W > 0, "to_split: bad shape B=%d Cin=%d H=%d W=%d (Cin %% 8)", B, Cin, H, W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || arith == SGDFR_SPLIT_FP16F8, "to_split: arith must be SGDFR_SPLIT_BF16/FP16/FP16F8");
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && s && xs && (reinterpret_cast<uintptr_t>(xs) & 15) == 0, "to_split: null or misaligned pointer");
    int64_t g = ((int64_t)B * (Cin / 8) * H * W + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (arith == SGDFR_SPLIT_FP16F8)
        hipLaunchKernelGGL(to_split_kernel<SGDFR_SPLIT_FP16F8>, dim3((int)g), dim3(256), 0, as_stream(stream), x, s,
                           reinterpret_cast<unsigned char*>(xs), B, Cin, H * W, sat);
    else if (arith == SGDFR_SPLIT_FP16)
        hipLaunchKernelGGL(to_split_kernel<SGDFR_SPLIT_FP16>, dim3((int)g), dim3(256), 0, as_stream(stream), x, s,
                           reinterpret_cast<unsigned char*>(xs), B, Cin, H * W, sat);
    else
        hipLaunchKernelGGL(to_split_kernel<SGDFR_SPLIT_BF16>, dim3((int)g), dim3(256), 0, as_stream(stream), x, s,
                           reinterpret_cast<unsigned char*>(xs), B, Cin, H * W, sat);
    return check_launch("to_split");
}

extern "C" int sgdfr_planes_to_split_f32(const float* gt, const float* d, unsigned short* xs, int B, int C, int H, int W,
                                        int arith, unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "planes_to_split: bad shape B=%d C=%d H=%d W=%d (C %% 8)", B, C, H, W);
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "planes_to_split: arith must be SGDFR_SPLIT_BF16/FP16");
    if (B == 0) return 0;
    SGDFR_REQUIRE(gt && xs && (reinterpret_cast<uintptr_t>(xs) & 15) == 0, "planes_to_split: null or misaligned pointer");
    const int RP = (H + 1) * (W + 1);
    int64_t g = ((int64_t)B * 4 * (C / 8) * RP + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (arith == SGDFR_SPLIT_FP16)
        hipLaunchKernelGGL(planes_to_split_kernel<SGDFR_SPLIT_FP16>, dim3((int)g), dim3(256), 0, as_stream(stream), gt, d,
                           reinterpret_cast<unsigned char*>(xs), B, C, RP, sat);
    else
        hipLaunchKernelGGL(planes_to_split_kernel<SGDFR_SPLIT_BF16>, dim3((int)g), dim3(256), 0, as_stream(stream), gt, d,
                           reinterpret_cast<unsigned char*>(xs), B, C, RP, sat);
    return check_launch("planes_to_split");
}

unsigned int blur_split_saturation_count(int reset);     // upfirdn2d.hip's counter
namespace sgdfr { unsigned int wsplit_saturation_count(int reset); }     // wsplit.hip's

// Number of fp16-split operand pairs that hit the +-65504 clamp (|x*s| > 1.04e6) since the last reset, over all split
// kernels on the current device; synchronises the device.  Negative: HIP error.
extern "C" long long sgdfr_split_saturation_count(int reset) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_saturated), sizeof(v)) != hipSuccess) return -1;
    if (reset) {
        const unsigned int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_saturated), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return (long long)v + (long long)blur_split_saturation_count(reset) + (long long)sgdfr::wsplit_saturation_count(reset);
}

#ifdef SGDFR_SPLIT_PROBE
extern "C" int sgdfr_split_trace_read(unsigned long long* out, int n) {     // probe builds only (scripts/tile_trace.py)
    if (n > 4096) n = 4096;
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int sgdfr_modconv2d_split_xin_supported(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    const SplitPlan* plan = split_plan(B, Cin, Cout, H, W, mode, &p);
    if (!plan || !(plan->cfg == 0 || plan->cfg == 1 || plan->cfg == 3 || plan->cfg == 4 || plan->cfg == 5)) return 0;     // the instantiations of launch_plan(xin)
    if (plan->nss == 3 && plan->cfg != 5 && split_lds_bytes(p, plan->nt, plan->nss, false, 3) > 160 * 1024) return 0;   // the 3-slot weight ring must fit
    return 1;
}

extern "C" int sgdfr_modconv2d_split_cout_tiles(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    return split_plan(B, Cin, Cout, H, W, mode, &p) ? p.n_cout_tiles : 0;
}

// the same for a launch with a pre-split input and no K slices (x_is_split = 1, ksplit <= 1): the plan such a launch really takes
extern "C" int sgdfr_modconv2d_split_cout_tiles_xin(int B, int Cin, int Cout, int H, int W, int mode) {
    SplitParams p;
    return split_plan(B, Cin, Cout, H, W, mode, &p, true) ? p.n_cout_tiles : 0;
}

// cout tiles of 64, + a 16-byte trailer (SGDFR_SPLIT_FP16F8 packs keep max |w * scale| there: the source of their fp8 exponent)
static int64_t split_pack_body_elems(int n_out, int n_in) { return (int64_t)((n_out + 63) / 64 * 64) * n_in * 9 * 2; }
extern "C" int64_t sgdfr_modconv_prepack_split_elems(int Cout, int Cin) { return split_pack_body_elems(Cout, Cin) + 8; }

extern "C" int sgdfr_modconv_prepack_split_f32(const float* weight, unsigned short* wsp, int Cout, int Cin, int arith,
                                               int transpose_flip, unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 || (arith == SGDFR_SPLIT_FP16F8 && transpose_flip == 0),
                  "prepack_split: arith must be SGDFR_SPLIT_BF16/FP16 (or FP16F8 for a forward pack)");
    SGDFR_REQUIRE(transpose_flip >= 0 && transpose_flip <= 2, "prepack_split: transpose_flip is 0 (forward), 1 (adjoint of "
                  "the plain conv) or 2 (adjoint of the transposed conv)");
    const int n_out = transpose_flip ? Cin : Cout, n_in = transpose_flip ? Cout : Cin;
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && n_in % SPLIT_CB == 0 && (n_out % 64 == 0 || (transpose_flip == 0 && n_out % 64 == 32)),
                  "prepack_split: needs in-channels %% 16 == 0 and out-channels %% 64 == 0 (forward pack: %% 32), got in=%d out=%d",
                  n_in, n_out);
    SGDFR_REQUIRE(weight && wsp, "prepack_split: null pointer");
    if (n_out % 64 != 0) {       // zero rows for the padding half of the last 64-cout tile
        if (hipMemsetAsync(wsp, 0, sizeof(unsigned short) * (size_t)split_pack_body_elems(Cout, Cin), as_stream(stream)) !=
            hipSuccess) {
            set_error("prepack_split: hipMemsetAsync failed");
            return 2;
        }
    }
    const int64_t n = (int64_t)Cout * Cin * 9;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    const float scale = (arith != SGDFR_SPLIT_BF16 ? SPLIT_F16_WSCALE : 1.f) / sqrtf((float)Cin * 9);
    {
        unsigned* const trailer = reinterpret_cast<unsigned*>(wsp + split_pack_body_elems(n_out, n_in));
        if (hipMemsetAsync(trailer, 0, 16, as_stream(stream)) != hipSuccess) { (void)hipGetLastError(); set_error("prepack_split: hipMemsetAsync failed"); return 2; }
        if (arith == SGDFR_SPLIT_FP16F8) hipLaunchKernelGGL(split_absmax_kernel, dim3(256), dim3(256), 0, as_stream(stream), weight, n, scale, trailer);
    }
    hipLaunchKernelGGL(prepack_split_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wsp, Cout, Cin,
                       64, scale, arith, transpose_flip, sat);
    return check_launch("modconv_prepack_split");
}

template <int MODE, int ET, int WM, int WN, int MI, int NI, int NSS = 3, bool XIN = false>
static int launch_split(const SplitParams& p, hipStream_t st, int lid0 = 0, int count = -1) {
    constexpr int NTHR = WM * WN * 64;
    const int nex = (2 * p.xs + NTHR - 1) / NTHR;
    constexpr int NEX_MAX = (MODE == SGDFR_MODE_DOWN3) ? 3 : 4;    // staged positions <= 4 x 256 (UP3: PT + P + 2, 4 slots from P = 513 on)
    SGDFR_REQUIRE(nex <= NEX_MAX, "modconv_split: staged range %d too long for mode %d", p.xlen, MODE);
    void (*kern)(SplitParams) = nex <= 2   ? split_mfma_kernel<MODE, ET, WM, WN, MI, NI, 2, NSS, XIN>
                                : nex == 3 ? split_mfma_kernel<MODE, ET, WM, WN, MI, NI, 3, NSS, XIN>
                                           : split_mfma_kernel<MODE, ET, WM, WN, MI, NI, NEX_MAX, NSS, XIN>;
    const size_t lds = split_lds_bytes(p, WM * MI * 32, NSS, MODE == SGDFR_MODE_DOWN3,
                                       (XIN && MODE != SGDFR_MODE_DOWN3 && NSS == 3 && NI <= 2 && WM * WN == 8) ? 3 : 2, WN * NI * 32,
                                       XIN && WM * WN == 4);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv_split: LDS request %zu B refused (B=%d Cin=%d Cout=%d H=%d W=%d xs=%d simgs=%d)", lds, p.B, p.Cin, p.Cout,
                  p.H, p.W, p.xs, p.simgs);
        return 2;
    }
    SplitParams q = p;
    q.total_blocks = count >= 0 ? count : p.n_pix_tiles * p.n_cout_tiles * p.ksplit;      // (lid0 / count: a sub-range of the tiles)
    q.lid0 = lid0;
    // Persistent blocks (one per CU, each walking its share of the tiles and staging the next tile's first channel block
    // and first two weight slabs while the current one finishes; see RING3 in the kernel).  Round 1 (two weight slots: the
    // next tile's first wait drained the tile's own stores) only gained on the layers of >= 12 short tiles per CU; with the
    // three-slot ring the 4- and 8-tiles-per-CU plain layers gain too (same-box A/B at B=64: 807 -> 795, 895 -> 868,
    // 1010 -> 995, 1161 -> 1132 us on the 32^2 .. 256^2 layers).  The transposed conv stays one block per tile: its row
    // sub-stage plan (the only one whose ring fits three slots) is 5-12 % slower than all nine taps between two barriers.
    static const int persist = getenv("SGDFR_SPLIT_PERSIST") ? atoi(getenv("SGDFR_SPLIT_PERSIST")) : 256;
    static const int persist_min = getenv("SGDFR_SPLIT_PERSIST_MIN") ? atoi(getenv("SGDFR_SPLIT_PERSIST_MIN")) : 4;    // tiles per block
    static const int persist_min_up = getenv("SGDFR_SPLIT_PERSIST_MIN_UP") ? atoi(getenv("SGDFR_SPLIT_PERSIST_MIN_UP")) : 4;    // (only reachable with SGDFR_SPLIT_UP_NARROW=1)
    constexpr bool kCanPersist = XIN && NSS == 3 && WM * WN == 8 && (MODE == SGDFR_MODE_PLAIN3 || MODE == SGDFR_MODE_UP3);
    const bool persistent = kCanPersist && persist > 0 &&
                            q.total_blocks >= (MODE == SGDFR_MODE_UP3 ? persist_min_up : persist_min) * persist;
    const int grid = persistent ? persist : q.total_blocks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, st, q);
    return check_launch("modconv2d_split");
}

// The deep-plan transposed conv with its TAIL ROUND on half tiles.  One block holds a CU, so a launch of T tiles costs
// ceil(T / 256) tile times: 1092 / 2114 / 4161 tiles (the 32^2 / 64^2 / 128^2 levels at B=64) pay for 5 / 9 / 17 rounds, the last
// one 27 % / 26 % / 25 % full -- 15 % / 8 % / 4 % of the launch; at B=32 and B=16 (configs[2], configs[4]) twice and four times
// that.  When the last round has <= 128 tiles, the whole rounds run as before (tiles [0, n_full)) and a second launch covers
// the same remaining (cout tile, position) range with 64 x 128 tiles (NI = 1: 2 x as many blocks of half the work, still one
// round): 4.5 / 8.5 / 16.5 rounds.  Per output element nothing changes -- same channel-block, tap and product order -- so the
// planes are bit-identical (tests/test_gpu_split.py).  SGDFR_SPLIT_UP_TAIL=0 switches it off, n > 1 = allow it up to n tiles of whole
// rounds (read per call: same-process A/B).
template <int MODE, int ET, int MI, bool XIN>
static int launch_deep_tail(const SplitParams& p, hipStream_t st) {
    const char* env = getenv("SGDFR_SPLIT_UP_TAIL");
    const int total = p.n_pix_tiles * p.n_cout_tiles;
    const int n_full = total / 256 * 256, tail = total - n_full;
    // Measured (scripts/up_tail_ab.py, profiles/r06_up_tail_ab.txt): it pays while the launch is SHORT -- up to two whole rounds
    // (512@16^2 at B=64: 258 -> 239 us; B=32: 164 -> 141 and 255 -> 234 us) -- and loses 1-7 % from four rounds on: blocks of a long
    // launch finish at different times, so its last round is already ragged, and the second launch adds a drain + launch gap.
    // (the adjoint -- 128 x 256 tiles, no plane stores -- still gains at four whole rounds: 179 -> 175, 301 -> 286, 616 -> 544 us; from
    //  eight on it is +- 2 %)
    const int max_full = (env && atoi(env) > 1) ? atoi(env) : (MODE == SGDFR_MODE_DOWN3 ? 1024 : 512);
    if ((env && atoi(env) == 0) || p.ksplit != 1 || n_full == 0 || n_full > max_full || tail == 0 || tail > 128)
        return launch_split<MODE, ET, 2, 4, MI, 2, 1, XIN>(p, st);
    if (int rc = launch_split<MODE, ET, 2, 4, MI, 2, 1, XIN>(p, st, 0, n_full)) return rc;
    SplitParams t = p;           // the same positions in 128-wide tiles
    constexpr int PT2 = 128;
    t.xlen = PT2 + p.P + 2;
    t.xs = (t.xlen + 63) & ~63;                  // (8 waves: staged positions padded to whole 64-lane pieces, as split_geometry does)
    t.simgs = (t.xlen - 1) / p.rps + 2;
    t.n_pix_tiles = (int)((p.total_pix + PT2 - 1) / PT2);
    t.desync = 0;
    fill_fastdivs(t);
    const int c0 = n_full / p.n_pix_tiles, p0 = n_full - c0 * p.n_pix_tiles;      // cout-major tile order: (c0, p0) is the first tile left
    const int lid0 = c0 * t.n_pix_tiles + 2 * p0;
    const int count = t.n_cout_tiles * t.n_pix_tiles - lid0;
    if (count <= 0) return 0;
    return launch_split<MODE, ET, 2, 4, MI, 1, 1, XIN>(t, st, lid0, count);
}

template <int ET>
static int launch_plan(int cfg, const SplitParams& p, hipStream_t st, bool xin) {
    if (xin) {
        switch (cfg) {
            case 0: return launch_split<SGDFR_MODE_PLAIN3, ET, 2, 4, 2, 2, 3, true>(p, st);
            case 1: return launch_split<SGDFR_MODE_PLAIN3, ET, 1, 8, 2, 2, 3, true>(p, st);
            case 3: return launch_split<SGDFR_MODE_UP3, ET, 2, 4, 1, 2, 3, true>(p, st);
            case 4: return launch_deep_tail<SGDFR_MODE_UP3, ET, 1, true>(p, st);
            case 5: return launch_deep_tail<SGDFR_MODE_DOWN3, ET, 2, true>(p, st);      // (the adjoint: 128 x 256 tiles, tail on 128 x 128)
            case 6: return launch_split<SGDFR_MODE_PLAIN3, ET, 2, 4, 2, 4, 3, true>(p, st);
            case 8: return launch_split<SGDFR_MODE_PLAIN3, ET, 1, 4, 2, 2, 3, true>(p, st);
            default: set_error("modconv_split: pre-split input is not built for tiling plan %d", cfg); return 1;
        }
    }
    if (cfg == 5) {
        set_error("modconv_split: DOWN3 reads the phase-major split planes of sgdfr_planes_to_split_f32 (x_is_split = 1)");
        return 1;
    }
    switch (cfg) {
        case 0: return launch_split<SGDFR_MODE_PLAIN3, ET, 2, 4, 2, 2, 3>(p, st);
        case 1: return launch_split<SGDFR_MODE_PLAIN3, ET, 1, 8, 2, 2, 3>(p, st);
        case 2: return launch_split<SGDFR_MODE_UP3, ET, 4, 2, 1, 2, 3>(p, st);
        case 3: return launch_split<SGDFR_MODE_UP3, ET, 2, 4, 1, 2, 3>(p, st);
        default: return launch_deep_tail<SGDFR_MODE_UP3, ET, 1, false>(p, st);      // (the autograd forward's fp32-input form of the same plan)
    }
}

extern "C" int sgdfr_modconv2d_split_f32(const float* x, int64_t x_bstride, const unsigned short* wsp, const float* s,
                                         const float* d, const float* noise, int64_t noise_bstride, const float* noise_w,
                                         const float* bias, const float* zeros, float* y, float* partials, int ksplit,
                                         const float* rgb_w, const float* rgb_s, float* rgb_part, int x_is_split,
                                         unsigned short* xs_out, const float* s_next, int B, int Cin, int Cout, int H, int W,
                                         int mode, int64_t plane_stride, int arith, int act, float slope, float gain,
                                         unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16 ||
                      (arith == SGDFR_SPLIT_FP16F8 && sgdfr_modconv2d_split_f8_ok(B, Cin, Cout, H, W, mode) && x_is_split && ksplit <= 1),
                  "modconv_split: arith must be SGDFR_SPLIT_BF16/FP16 (FP16F8: only where sgdfr_modconv2d_split_f8_ok() says so, with a "
                  "pre-split input and no K slices)");
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv_split: bad shape B=%d Cin=%d Cout=%d H=%d W=%d",
                  B, Cin, Cout, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(sgdfr_modconv2d_split_supported(B, Cin, Cout, H, W, mode),
                  "modconv_split: shape B=%d Cin=%d Cout=%d H=%d W=%d mode=%d not supported; use sgdfr_modconv2d_fwd_f32", B,
                  Cin, Cout, H, W, mode);
    SGDFR_REQUIRE(mode == SGDFR_MODE_PLAIN3 || (!noise && !bias && !act), "modconv_split: UP3 writes raw parity planes "
                  "(noise / bias / activation belong to sgdfr_blur_bias_act_f32), DOWN3 a raw gradient");
    SGDFR_REQUIRE(mode != SGDFR_MODE_DOWN3 || (x_is_split && y && !rgb_part && !xs_out), "modconv_split: DOWN3 takes "
                  "pre-split planes and writes y only");
    SGDFR_REQUIRE(x && wsp && zeros && (y || rgb_part || xs_out) && (s || x_is_split), "modconv_split: null pointer");
    SGDFR_REQUIRE(!xs_out || (s_next && mode == SGDFR_MODE_PLAIN3 && ksplit <= 1 && Cout % 8 == 0 &&
                              (reinterpret_cast<uintptr_t>(xs_out) & 15) == 0),
                  "modconv_split: xs_out needs s_next, mode PLAIN3, ksplit == 1 and a 16-byte aligned buffer");
    SGDFR_REQUIRE(!x_is_split || (x_bstride != 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0),
                  "modconv_split: a pre-split input is per image and 16-byte aligned");
    SGDFR_REQUIRE(!noise || noise_w, "modconv_split: noise without noise_w");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(wsp) | reinterpret_cast<uintptr_t>(s)) & 15) == 0,
                  "modconv_split: wsp and s must be 16-byte aligned");
    SplitParams p;
    const SplitPlan* plan = split_plan(B, Cin, Cout, H, W, mode, &p, x_is_split != 0 && ksplit <= 1);
    if (plane_stride != 0) {
        // UP3 with padded parity planes y [B,Cout,4,plane_stride]: a multiple of 32 floats makes every 32-position store run of
        // the epilogue one whole 128-byte line (the dense (H+1)(W+1) planes are odd-sized: every run straddles two lines and
        // plane stores run at 2.4 instead of 4.7 TB/s, scripts/store_probe.hip).  The flat position space gets the same
        // per-image stride, so tiles stay 256 consecutive positions; the padding positions are computed and dropped.
        SGDFR_REQUIRE(mode == SGDFR_MODE_UP3 && plane_stride >= (int64_t)p.R * p.P && plane_stride < (1 << 30) && ksplit <= 1,
                      "modconv_split: plane_stride is for UP3 launches without K slices, >= (H+1)*(W+1)");
        p.rps = (int)plane_stride;
        static const int il_env = getenv("SGDFR_PLANE_IL") ? atoi(getenv("SGDFR_PLANE_IL")) : 1;      // (0: padded PLANAR planes, A/B only)
        p.plane_il = il_env;
        p.total_pix = (int64_t)B * p.rps;
        SGDFR_REQUIRE(p.total_pix + 4ll * p.P + 8 < (1ll << 31), "modconv_split: padded position space too large");
        p.n_pix_tiles = (int)((p.total_pix + plan->pt - 1) / plan->pt);
        p.simgs = (p.xlen - 1) / p.rps + 2;
    }
    p.f8_max = nullptr;
    p.x = x; p.x_bstride = x_bstride; p.wsp = wsp; p.s = s; p.d = d; p.noise = noise; p.noise_bstride = noise_bstride;
    p.noise_w = noise_w; p.bias = bias; p.zeros = zeros; p.y = y;
    p.act = act; p.slope = slope; p.gain = gain; p.sat = sat;
    if (ksplit < 1) ksplit = 1;
    SGDFR_REQUIRE(!rgb_part || (rgb_w && rgb_s && mode == SGDFR_MODE_PLAIN3 && ksplit == 1),
                  "modconv_split: the fused ToRGB needs rgb_w, rgb_s, mode PLAIN3 and ksplit == 1");
    p.rgb_w = rgb_w; p.rgb_s = rgb_s; p.rgb_part = rgb_part;
    p.xs_out = reinterpret_cast<unsigned char*>(xs_out); p.s_next = s_next;
    SGDFR_REQUIRE(ksplit == 1 || (partials && ksplit <= Cin / SPLIT_CB), "modconv_split: ksplit %d needs a partials buffer "
                  "and at most %d slices", ksplit, Cin / SPLIT_CB);
    const int64_t n_out = (mode == SGDFR_MODE_UP3) ? (int64_t)B * Cout * 4 * p.rps : (int64_t)B * Cout * H * W;
    p.ksplit = ksplit;
    p.split_stride = n_out;
    if (ksplit > 1) p.y = partials;
    static const int stagger = getenv("SGDFR_SPLIT_STAGGER") ? atoi(getenv("SGDFR_SPLIT_STAGGER")) : 1;
    p.stagger = stagger;
    p.dbg = getenv("SGDFR_SPLIT_DBG") ? atoi(getenv("SGDFR_SPLIT_DBG")) : 0;
    {
        // block time ~ K loop (MFMAs of the two waves of a SIMD, ~55 % busy) + epilogue stores at the per-CU HBM share
        // Measured: +4..6 % on the 32x32..128x128 transposed layers at 50-100 % of the estimate, nothing (or a loss) on the
        // plain layers, so only the transposed conv uses it.
        static const int pct = getenv("SGDFR_SPLIT_DESYNC") ? atoi(getenv("SGDFR_SPLIT_DESYNC")) : 75;
        const bool up = mode == SGDFR_MODE_UP3;
        const double mfma_clk = (double)(Cin / SPLIT_CB / ksplit) * (up ? 54 : 108) * 32 * 2 / 0.55;
        const double store_clk = (double)plan->nt * plan->pt * (up ? 4 : 1) * 4 / 8.3;
        const int blocks = p.n_pix_tiles * p.n_cout_tiles * ksplit;
        p.desync = (up && pct > 0 && blocks >= 1024) ? (int)((mfma_clk + store_clk) * pct / 100 / 4096) : 0;   // >= 4 rounds
    }
    fill_fastdivs(p);
    hipStream_t st = as_stream(stream);
    if (arith == SGDFR_SPLIT_FP16F8) {      // (the deep transposed plan with a pre-split input: checked above)
        p.f8_max = reinterpret_cast<const float*>(wsp + split_pack_body_elems(Cout, Cin));
        return mode == SGDFR_MODE_UP3 ? launch_split<SGDFR_MODE_UP3, SGDFR_SPLIT_FP16F8, 2, 4, 1, 2, 1, true>(p, st)
                                      : launch_split<SGDFR_MODE_PLAIN3, SGDFR_SPLIT_FP16F8, 1, 4, 2, 2, 3, true>(p, st);
    }
    const int rc = arith == SGDFR_SPLIT_FP16 ? launch_plan<SGDFR_SPLIT_FP16>(plan->cfg, p, st, x_is_split != 0)
                                             : launch_plan<SGDFR_SPLIT_BF16>(plan->cfg, p, st, x_is_split != 0);
    if (rc || ksplit == 1) return rc;
    const bool plain = mode == SGDFR_MODE_PLAIN3;
    return launch_splitk_reduce(partials, ksplit, n_out, plain ? noise : nullptr, noise_bstride, noise_w, plain ? bias : nullptr, y,
                                Cout, plain ? H * W : 1, plain ? act : 0, slope, gain, st);
}
