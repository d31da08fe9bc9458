// Transposed 3x3 modulated conv (MODE_UP3 of split.hip, pre-split input, interleaved parity planes out) with the plane stores
// taken OFF the matrix cores' critical path: two wave groups per block that swap roles every tile.
//
// split.hip's transposed conv spends 45-50 % of a tile in its epilogue: 262 KB of plane stores per 64 cout x 256 position tile
// leave a CU at ~10 B/clk whatever the other CUs do, one 8-wave block per CU, so the matrix cores idle meanwhile (PMC: MFMA-busy
// 48-54 % on a kernel whose K loop alone runs the pipe at > 90 %).  Loads and stores retire through one in-order vmcnt per wave,
// so a wave with stores in flight cannot run the next tile's DMA-fed K loop -- but ANOTHER wave can.  Here a block's eight waves
// are two groups of four (one wave per SIMD each), each owning 64 couts x 128 positions x 4 parity phases of accumulators:
//   * the K group runs the MFMAs of its tile, one 16-channel block (all nine taps) per slot, reading operands from LDS only;
//   * the other group is loader + storer: in every slot it DMAs the K group's NEXT channel block (activations + the weight slab)
//     into the other halves of the LDS double buffers, stores 1/S of its own finished tile's planes, and waits with a COUNTED
//     vmcnt that leaves exactly this slot's stores in flight -- the DMA (issued first) has landed, the stores get a whole slot;
//   * one s_barrier per slot publishes the DMA'd operands; after S slots (S = Cin / 16) the groups swap roles.
// Same MFMA order per accumulator as split.hip's deep plan (bit-identical planes); tiles are half as large, so a layer has twice as
// many, finer tiles (the 16x16 level's 2.5 rounds of tiles become 5.0) at twice the operand DMA per MFMA.
#include "split_kernel.h"

namespace sgdfr {

constexpr int PP_NT = 64, PP_PT = 128, PP_NI = 2;
constexpr int PP_WBYTES = 9 * 4096;          // weight slab of one channel block and cout tile: [tap 9][part 2][k-half 2][64][8] x 16 bit

template <int ET>
__global__ __launch_bounds__(512, 1) void up_pp_kernel(SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xbuf_bytes = 64 * p.xs;                       // [part 2][k-half 2][xs][8] x 16 bit
    unsigned char* const xb0 = smem;
    unsigned char* const wb0 = smem + 2 * xbuf_bytes;
    float* const dl = reinterpret_cast<float*>(wb0 + 2 * PP_WBYTES);      // [group 2][64] d * output scale

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3, wm = wg >> 1, wn = wg & 1;
    const int gtid = tid & 255;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W, RP = p.R * p.P;
    const int S = p.Cin / SPLIT_CB;

    // this block's tiles: the XCDs take eight contiguous ranges of the (cout tile, position tile) list, a block every (blocks on its
    // XCD)-th tile of its XCD's range (wsplit.hip's order: neighbouring tiles of one cout tile meet in one L2)
    const int nblk = (int)gridDim.x, xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int bk = (nblk >> 3) + (xcd < (nblk & 7) ? 1 : 0);
    const int tq = p.total_blocks >> 3, trm = p.total_blocks & 7;
    const int nk = tq + (xcd < trm ? 1 : 0), sk = xcd * tq + min(xcd, trm);
    const int nt = bidx < nk ? (nk - bidx + bk - 1) / bk : 0;          // tiles of this block
    auto lid_of = [&](int j) { return sk + bidx + j * bk; };
    struct Tile { int ct, q0; };
    auto tile_of = [&](int lid) -> Tile {
        Tile t;
        t.ct = fdiv(lid, p.fd_npt);
        t.q0 = (lid - t.ct * p.n_pix_tiles) * PP_PT;
        return t;
    };
    // loader item (k-half h, staged position j) of a tile: byte offset of the hi chunk of channel block 0, -1: zero page, -2: none
    auto xaddr_of = [&](const Tile& t, int e) -> int64_t {
        const int i = gtid + e * 256;
        const int h = fdiv(i, p.fd_xs), j = i - h * p.xs;
        if (!(h < 2)) return -2;
        const int q = t.q0 + j;
        const int img = fdiv(q, p.fd_rps);
        const int r = q - img * p.rps;
        const int pr = fdiv(r, p.fd_P), pc = r - pr * p.P;
        const bool ok = j < p.xlen && r < RP && pc >= 1 && pr >= 1 && img < p.B;
        return ok ? ((((int64_t)img * (p.Cin / 8) + h) * 2) * HW + (pr - 1) * p.W + (pc - 1)) * 16 : -1;
    };
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    constexpr int NEXL = 3;                                  // loader items per thread (2 * xs <= 768)
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(p.x);
    // stage channel block cb of tile t into x buffer / weight slot `sel`; returns the DMA instructions this wave issued
    auto stage = [&](const Tile& t, const int64_t (&xa)[NEXL], int cb, int sel) -> int {
        int n = 0;
        unsigned char* const xb = xb0 + sel * xbuf_bytes;
#pragma unroll
        for (int e = 0; e < NEXL; ++e) {
            const int i0 = __builtin_amdgcn_readfirstlane(gtid - lane + e * 256);
            if (i0 >= 2 * p.xs) continue;                    // wave-uniform (xs % 64 == 0)
            const int h0 = fdiv(i0, p.fd_xs), j0 = i0 - h0 * p.xs;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const unsigned char* src = xa[e] >= 0 ? xbase + xa[e] + ((int64_t)cb * 4 + part) * HW * 16
                                                      : reinterpret_cast<const unsigned char*>(p.zeros);
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(xb + part * 32 * p.xs + (h0 * p.xs + j0) * 16), 16, 0, 0);
            }
            n += 2;
        }
        const unsigned char* const wsrc = reinterpret_cast<const unsigned char*>(p.wsp) + ((int64_t)t.ct * S + cb) * PP_WBYTES;
        unsigned char* const wdst = wb0 + sel * PP_WBYTES;
#pragma unroll
        for (int v = 0; v < 9; ++v) {
            const int c = wg + 4 * v;
            __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + c * 1024 + lane * 16), (lds_void*)(wdst + c * 1024), 16, 0, 0);
        }
        return n + 9;
    };

    f32x16 acc[4][PP_NI];
    int boff[PP_NI], dm[PP_NI];      // dm: image of this lane's position relative to the tile's first one (0 | 1)
    int64_t ybase[PP_NI];       // float offset of this lane's position in the interleaved planes (cout 0 of the tile), -1: none
    const int aoff = (hi * 64 + wm * 32 + l31) * 16;
    const float oscale = (ET == SGDFR_SPLIT_FP16) ? SPLIT_F16_OUT : 1.f;

    // prologue: the loader group (group 1) stages slot 0 of the first tile
    int u = 0;                  // running slot index: operands of slot u live in buffers u & 1
    if (nt > 0 && grp == 1) {
        const Tile t0 = tile_of(lid_of(0));
        int64_t xa[NEXL];
#pragma unroll
        for (int e = 0; e < NEXL; ++e) xa[e] = xaddr_of(t0, e);
        stage(t0, xa, 0, 0);
        split_wait_vmcnt<0>();
    }
    __syncthreads();

    for (int phi = 0; phi <= nt; ++phi) {
        const int kg = phi & 1;
        const bool k_valid = phi < nt;
        if (grp == kg) {
            // ---------------- K role: MFMAs of tile L[phi]
            float dval = 1.f;
            if (k_valid) {
                const Tile t = tile_of(lid_of(phi));
#pragma unroll
                for (int n = 0; n < PP_NI; ++n) {
                    const int l = (wn * PP_NI + n) * 32 + l31;
                    int64_t pix = (int64_t)t.q0 + l;
                    const bool ok = pix < p.total_pix;
                    if (!ok) pix = p.total_pix - 1;
                    const int img = fdiv((int)pix, p.fd_rps);
                    const int rem = (int)(pix - (int64_t)img * p.rps);
                    boff[n] = l + hi * p.xs;
                    dm[n] = img - fdiv(t.q0, p.fd_rps);
                    ybase[n] = (ok && rem < RP) ? (((int64_t)img * p.Cout + t.ct * PP_NT) * p.rps + rem) * 4 : -1;
                }
                // d of this tile's couts for the (at most two) images its positions belong to: [image 2][cout 64] through LDS
                if (gtid < 2 * PP_NT) {
                    const int m = gtid / PP_NT, c = gtid - m * PP_NT;
                    const int img = fdiv(t.q0, p.fd_rps) + m;
                    dval = (p.d && img < p.B) ? p.d[(int64_t)img * p.Cout + t.ct * PP_NT + c] : 1.f;
                }
#pragma unroll
                for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                    for (int n = 0; n < PP_NI; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ph][n][r] = 0.f;
            }
            for (int s = 0; s < S; ++s, ++u) {
                if (k_valid) {
                    const unsigned char* const xcur = xb0 + (u & 1) * xbuf_bytes;
                    const unsigned char* const wslot = wb0 + (u & 1) * PP_WBYTES;
                    frag128 ub[2][2][PP_NI];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const unsigned char* const wcur = wslot + ky * 3 * 4096;
                        const int rowoff = (ky == 2) ? 0 : p.P;
                        frag128 a[2][2];
                        auto fetch_a = [&](int set, int kx) {
#pragma unroll
                            for (int part = 0; part < 2; ++part)
                                a[set][part] = *reinterpret_cast<const frag128*>(wcur + aoff + (kx * 2 + part) * 2048);
                        };
                        fetch_a(0, 0);
                        if (ky != 1) {      // (kernel rows 0 and 1 read the same input row)
#pragma unroll
                            for (int part = 0; part < 2; ++part)
#pragma unroll
                                for (int o = 1; o >= 0; --o)
#pragma unroll
                                    for (int n = 0; n < PP_NI; ++n)
                                        ub[o][part][n] =
                                            *reinterpret_cast<const frag128*>(xcur + part * 32 * p.xs + (boff[n] + rowoff + o) * 16);
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ph = 2 * (ky & 1) + (kx & 1);
                            const int o = (kx == 2) ? 0 : 1;
                            const int cur = kx & 1;
                            if (kx < 2) fetch_a(cur ^ 1, kx + 1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                                for (int n = 0; n < PP_NI; ++n)
                                    acc[ph][n] = split_mfma<ET>(a[cur][t3 == 2], ub[o][t3 == 1][n], acc[ph][n]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (s == 0 && gtid < 2 * PP_NT) dl[grp * 2 * PP_NT + gtid] = dval * oscale;      // (loaded before the first slot)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            // ---------------- loader + storer role: DMA for the K group's next slot, 1/S of the own finished tile's planes per slot
            const bool e_valid = phi >= 1;
            Tile tk{0, 0}, tn{0, 0};
            int64_t xa[NEXL] = {-2, -2, -2}, xan[NEXL] = {-2, -2, -2};
            if (k_valid) {
                tk = tile_of(lid_of(phi));
#pragma unroll
                for (int e = 0; e < NEXL; ++e) xa[e] = xaddr_of(tk, e);
            }
            const bool n_valid = phi + 1 < nt;
            if (n_valid) {
                tn = tile_of(lid_of(phi + 1));
#pragma unroll
                for (int e = 0; e < NEXL; ++e) xan[e] = xaddr_of(tn, e);
            }
            const float* const dg = dl + grp * 2 * PP_NT + wm * 32 + 4 * hi;
            for (int s = 0; s < S; ++s, ++u) {
                if (p.dbg & 4) {
                } else if (k_valid && s + 1 < S) stage(tk, xa, s + 1, (u + 1) & 1);
                else if (s + 1 == S && n_valid) stage(tn, xan, 0, (u + 1) & 1);
                // (the counted wait below relies on program order: DMA first, then this slot's stores)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
                int n_st = 0;
                if (e_valid && !(p.dbg & 2)) {
                    const int lo = s * 32 / S, hi_k = (s + 1) * 32 / S;          // this slot's share of the 32 quad stores per lane
#pragma unroll
                    for (int k = 0; k < 32; ++k) {
                        if (k < lo || k >= hi_k) continue;                       // wave-uniform
                        const int n = k >> 4, g = (k >> 2) & 3, j = k & 3;
                        ++n_st;
                        if (ybase[n] < 0) continue;
                        const float dv = dg[dm[n] * PP_NT + 8 * g + j];
                        float* const dst = p.y + ybase[n] + (int64_t)(wm * 32 + 8 * g + 4 * hi + j) * 4 * p.rps;
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[0][n][4 * g + j] * dv, acc[2][n][4 * g + j] * dv,
                                                                      acc[1][n][4 * g + j] * dv, acc[3][n][4 * g + j] * dv);
                    }
                }
                if (!(p.dbg & 1)) split_wait_vmcnt_dyn(n_st);      // (SGDFR_UPPP_DBG=1: never wait for the DMA -- wrong results, timing only)
                __builtin_amdgcn_s_barrier();
            }
        }
    }
}

}  // namespace sgdfr

using namespace sgdfr;

namespace {

int pp_geometry(int B, int Cin, int Cout, int H, int W, int64_t plane_stride, SplitParams* out) {
    if (B < 1 || Cin % SPLIT_CB != 0 || Cout % PP_NT != 0 || H < 1 || W < 1) return 0;
    SplitParams p{};
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 1; p.R = H + 1;
    if (plane_stride < (int64_t)p.R * p.P || plane_stride >= (1 << 30)) return 0;
    p.rps = (int)plane_stride;
    p.total_pix = (int64_t)B * p.rps;
    if (p.total_pix + 4ll * p.P + 8 >= (1ll << 31)) return 0;
    p.xlen = PP_PT + p.P + 2;
    p.xs = (p.xlen + 63) & ~63;
    if (2 * p.xs > 3 * 256) return 0;                      // three loader items per thread
    if (p.xlen - 1 >= p.rps) return 0;                     // a tile's staged range touches at most two images
    p.n_pix_tiles = (int)((p.total_pix + PP_PT - 1) / PP_PT);
    p.n_cout_tiles = Cout / PP_NT;
    if ((int64_t)p.n_pix_tiles * p.n_cout_tiles >= (1ll << 30)) return 0;
    p.total_blocks = p.n_pix_tiles * p.n_cout_tiles;
    if (out) *out = p;
    return 1;
}

size_t pp_lds_bytes(const SplitParams& p) { return 2 * (size_t)64 * p.xs + 2 * (size_t)PP_WBYTES + 4 * PP_NT * sizeof(float); }

}  // namespace

extern "C" int sgdfr_modconv2d_up_pp_supported(int B, int Cin, int Cout, int H, int W, int64_t plane_stride) {
    SplitParams p;
    return pp_geometry(B, Cin, Cout, H, W, plane_stride, &p) && pp_lds_bytes(p) <= 160 * 1024 ? 1 : 0;
}

extern "C" int sgdfr_modconv2d_up_pp_f32(const unsigned short* xs_in, const unsigned short* wsp, const float* d, const float* zeros,
                                         float* y, int B, int Cin, int Cout, int H, int W, int64_t plane_stride, int arith,
                                         void* stream) {
    SGDFR_REQUIRE(arith == SGDFR_SPLIT_BF16 || arith == SGDFR_SPLIT_FP16, "modconv2d_up_pp: arith must be SGDFR_SPLIT_BF16/FP16");
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv2d_up_pp: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(sgdfr_modconv2d_up_pp_supported(B, Cin, Cout, H, W, plane_stride), "modconv2d_up_pp: shape B=%d Cin=%d Cout=%d H=%d "
                  "W=%d plane_stride=%lld not supported; use sgdfr_modconv2d_split_f32(mode UP3)", B, Cin, Cout, H, W, (long long)plane_stride);
    SGDFR_REQUIRE(xs_in && wsp && zeros && y, "modconv2d_up_pp: null pointer");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(xs_in) | reinterpret_cast<uintptr_t>(wsp) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  "modconv2d_up_pp: xs_in, wsp and y must be 16-byte aligned");
    SplitParams p;
    pp_geometry(B, Cin, Cout, H, W, plane_stride, &p);
    p.x = reinterpret_cast<const float*>(xs_in); p.wsp = wsp; p.d = d; p.zeros = zeros; p.y = y; p.plane_il = 1;
    p.dbg = getenv("SGDFR_UPPP_DBG") ? atoi(getenv("SGDFR_UPPP_DBG")) : 0;
    fill_fastdivs(p);
    static const int persist = getenv("SGDFR_UPPP_GRID") ? atoi(getenv("SGDFR_UPPP_GRID")) : 256;
    const int grid = p.total_blocks < persist ? p.total_blocks : persist;
    const size_t lds = pp_lds_bytes(p);
    void (*kern)(SplitParams) = arith == SGDFR_SPLIT_FP16 ? up_pp_kernel<SGDFR_SPLIT_FP16> : up_pp_kernel<SGDFR_SPLIT_BF16>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error("modconv2d_up_pp: LDS request %zu B refused", lds);
        return 2;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, as_stream(stream), p);
    return check_launch("modconv2d_up_pp");
}
