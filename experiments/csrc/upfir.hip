// Upsampling StyledConv in ONE launch: the stride-2 transposed 3x3 modulated conv of split.hip (MODE_UP3, split-operand
// 16-bit MFMA) with the 4x4 FIR blur, noise, bias, leaky-ReLU and the hand-over to the next conv's split input form applied in
// its epilogue (reference: model.py:246-257 conv_transpose2d + Blur, model.py:303-337 StyledConv, op/upfirdn2d.py:168-209).
//
// The two-pass form (split.hip UP3 -> fp32 parity planes T[B,Cout,4,(H+1)(W+1)] -> upfirdn2d.hip blur_split_kernel) writes and
// re-reads 8 bytes of planes per output element; at B=64 the six blur launches were 1.26 ms of a 6.1 ms forward, at their HBM
// roof.  Here a block computes T for a TR x TC patch of the super-pixel grid (TR*TC <= 256 positions = the 64 cout x 256
// position tile of the "deep" transposed plan, all nine taps of a channel block between two barriers) and finishes the 2 x 2
// output quads of the patch's interior from T exchanged through LDS; the outermost ring of the patch is recomputed by the
// neighbouring tiles (halo), so T never leaves the CU.  The price is the halo: (TR-2)(TC-2) of TR*TC positions produce
// output, e.g. 14 x 18 patches for 128-wide inputs = 88 tiles per image instead of 65 (1.35 x the MFMA work of the layer).
// The kernel body is split_kernel.h's split_mfma_kernel<SGDFR_MODE_UPF, ...>; this file holds its geometry, launch and C ABI.
#include "split_kernel.h"

namespace sgdfr {

unsigned int upfir_saturation_count(int reset) {      // this translation unit's copy of the device-wide legacy counter
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_saturated), sizeof(v)) != hipSuccess) return 0;
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_split_saturated), &z, sizeof(z));
    }
    return v;
}

}  // namespace sgdfr

using namespace sgdfr;

namespace {

constexpr int kNT = 64, kPT = 256;       // couts x positions of a block (WM = 2, WN = 4, MI = 1, NI = 2)

// TR x TC patch with the fewest tiles per image (ties: the wider one -- longer store runs)
void upfir_patch(int H, int W, int* TR, int* TC) {
    static const int tc_env = getenv("SGDFR_UPFIR_TC") ? atoi(getenv("SGDFR_UPFIR_TC")) : 0;
    long best = -1;
    for (int tc = 4; tc <= 64; ++tc) {
        if (tc_env > 0 && tc != tc_env) continue;
        const int tr = kPT / tc;
        if (tr < 4) break;
        const long tiles = (long)((H + tr - 3) / (tr - 2)) * ((W + tc - 3) / (tc - 2));
        if (best < 0 || tiles <= best) { best = tiles; *TR = tr; *TC = tc; }
    }
}

int upfir_geometry(int B, int Cin, int Cout, int H, int W, SplitParams* out) {
    if (B < 1 || Cin % SPLIT_CB != 0 || Cout % kNT != 0 || H < 2 || W < 2) return 0;
    SplitParams p{};
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    if ((int64_t)4 * H * W >= (1ll << 30)) return 0;      // (pixel offsets inside an image are ints)
    p.patch = 1;
    upfir_patch(H, W, &p.TR, &p.TC);
    p.tstep_r = p.TR - 2; p.tstep_c = p.TC - 2; p.torg = -1;
    p.tiles_y = (H + p.tstep_r - 1) / p.tstep_r;
    p.tiles_x = (W + p.tstep_c - 1) / p.tstep_c;
    p.seglen = p.TC + 1;                         // staged input rows a-1 .. a+TR-1 x columns b-1 .. b+TC-1
    p.xlen = (p.TR + 1) * p.seglen;
    p.xs = (p.xlen + 63) & ~63;
    p.simgs = 1;
    p.n_pix_tiles = B * p.tiles_x * p.tiles_y;
    p.n_cout_tiles = Cout / kNT;
    p.total_pix = (int64_t)B * p.R * p.P;
    p.rps = p.R * p.P;
    p.tpos = kPT + 2 * (p.TC + 1);
    if ((int64_t)p.n_pix_tiles * p.n_cout_tiles >= (1ll << 30)) return 0;
    if ((2 * p.xs + 511) / 512 > 2) return 0;
    if (out) *out = p;
    return 1;
}

size_t upfir_lds_bytes(const SplitParams& p) {
    const size_t loop = 2 * (size_t)64 * p.xs + 2 * (size_t)kNT * 576;            // x double buffer + two weight slots of 9 taps
    const size_t exch = (size_t)16 * p.tpos * 16;                                 // [16 couts][tpos][4 phases] fp32
    const size_t style = (size_t)((p.simgs * p.Cin + 3) & ~3) * sizeof(float);    // (reserved by the kernel's table layout)
    const size_t epi = ((size_t)p.simgs * kNT * 6 + kNT + 512 * 3) * sizeof(float);
    return (loop > exch ? loop : exch) + style + epi;
}

template <int ET>
int launch_upfir(const SplitParams& p, hipStream_t st) {
    void (*kern)(SplitParams) = split_mfma_kernel<SGDFR_MODE_UPF, ET, 2, 4, 1, 2, 2, 1, true>;
    const size_t lds = upfir_lds_bytes(p);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv2d_upfir: LDS request %zu B refused (B=%d Cin=%d Cout=%d H=%d W=%d)", lds, p.B, p.Cin, p.Cout, p.H, p.W);
        return 2;
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(512), lds, st, p);
    return check_launch("modconv2d_upfir");
}

}  // namespace

extern "C" int sgdfr_modconv2d_upfir_supported(int B, int Cin, int Cout, int H, int W) {
    SplitParams p;
    if (!upfir_geometry(B, Cin, Cout, H, W, &p)) return 0;
    const size_t loop = 2 * (size_t)64 * p.xs + 2 * (size_t)kNT * 576;
    return upfir_lds_bytes(p) <= 160 * 1024 && (size_t)16 * p.tpos * 16 <= loop ? 1 : 0;
}

// tiles per image (tiles_y * tiles_x) and the patch, for the host's cost model / tests: returns tiles, writes TR / TC
extern "C" int sgdfr_modconv2d_upfir_tiles(int B, int Cin, int Cout, int H, int W, int* TR, int* TC) {
    SplitParams p;
    if (!upfir_geometry(B, Cin, Cout, H, W, &p)) return 0;
    if (TR) *TR = p.TR;
    if (TC) *TC = p.TC;
    return p.tiles_x * p.tiles_y;
}

extern "C" int sgdfr_modconv2d_upfir_split_f32(const unsigned short* xs_in, const unsigned short* wsp, const float* d, const float* fir,
                                               const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                                               const float* s_next, const float* zeros, unsigned short* xs_out, int B, int Cin,
                                               int Cout, int H, int W, int arith, int act, float slope, float gain,
                                               unsigned int* sat, void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "modconv2d_upfir: arith must be SGDFR_SPLIT_BF16/FP16");
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv2d_upfir: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(sgdfr_modconv2d_upfir_supported(B, Cin, Cout, H, W), "modconv2d_upfir: shape B=%d Cin=%d Cout=%d H=%d W=%d not "
                  "supported (Cin %% 16, Cout %% 64); use sgdfr_modconv2d_split_f32(mode UP3) + sgdfr_blur_bias_act_split_f32", B, Cin, Cout, H, W);
    SGDFR_REQUIRE(xs_in && wsp && d && fir && s_next && zeros && xs_out, "modconv2d_upfir: null pointer");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(xs_in) | reinterpret_cast<uintptr_t>(wsp) | reinterpret_cast<uintptr_t>(xs_out)) & 15) == 0,
                  "modconv2d_upfir: xs_in, wsp and xs_out must be 16-byte aligned");
    SGDFR_REQUIRE(!noise || (noise_w && (reinterpret_cast<uintptr_t>(noise) & 7) == 0 && noise_bstride % 2 == 0),
                  "modconv2d_upfir: noise needs noise_w, 8-byte alignment and an even batch stride");
    SplitParams p;
    upfir_geometry(B, Cin, Cout, H, W, &p);
    p.x = reinterpret_cast<const float*>(xs_in); p.x_bstride = (int64_t)Cin * H * W; p.wsp = wsp; p.s = nullptr; p.d = d;
    p.noise = noise; p.noise_bstride = noise_bstride; p.noise_w = noise_w; p.bias = bias; p.zeros = zeros; p.y = nullptr;
    p.rgb_w = nullptr; p.rgb_s = nullptr; p.rgb_part = nullptr;
    p.xs_out = reinterpret_cast<unsigned char*>(xs_out); p.s_next = s_next; p.sat = sat; p.fir = fir;
    p.act = act; p.slope = slope; p.gain = gain;
    p.ksplit = 1; p.split_stride = 0;
    static const int stagger = getenv("SGDFR_SPLIT_STAGGER") ? atoi(getenv("SGDFR_SPLIT_STAGGER")) : 1;
    p.stagger = stagger;
    p.dbg = getenv("SGDFR_SPLIT_DBG") ? atoi(getenv("SGDFR_SPLIT_DBG")) : 0;
    p.total_blocks = p.n_pix_tiles * p.n_cout_tiles;
    {
        // first-round start spread of the transposed conv (split.hip): block time ~ K loop + the epilogue's stores
        static const int pct = getenv("SGDFR_UPFIR_DESYNC") ? atoi(getenv("SGDFR_UPFIR_DESYNC")) : 75;
        const double mfma_clk = (double)(Cin / SPLIT_CB) * 54 * 32 * 2 / 0.55;
        const double store_clk = (double)kNT * (p.TR - 2) * (p.TC - 2) * 4 * 4 / 8.3;
        p.desync = (pct > 0 && p.total_blocks >= 1024) ? (int)((mfma_clk + store_clk) * pct / 100 / 4096) : 0;
    }
    fill_fastdivs(p);
    hipStream_t st = as_stream(stream);
    return arith == SGDFR_SPLIT_FP16 ? launch_upfir<SGDFR_SPLIT_FP16>(p, st) : launch_upfir<SGDFR_SPLIT_BF16>(p, st);
}
