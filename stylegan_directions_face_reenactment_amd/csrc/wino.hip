// Winograd F(2x2, 3x3) form of the shared-weight modulated 3x3 convolution on fp32 MFMA.
//
// The plain 3x3 layers are MFMA-bound at the fp32 matrix rate (157 TFLOP/s), so the only way past that roof is
// fewer multiplies: Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input patch with 16
// multiplies per (cin, cout) instead of 36 -- 2.25x less MFMA work, in exact-coefficient fp32 arithmetic
// (transform matrices hold only 0, +-1, +-1/2).
//
//     U = G g G^T          [Cin][16][Cout]   precomputed per weight version (sgdfr_modconv_prepack_wino_f32)
//     V = B^T (x*s) B      16 values per (cin, tile), computed IN REGISTERS from the staged input patch
//     M_xi[o][t] += U_xi[o][i] * V_xi[i][t]    sixteen independent GEMMs -> 16 MFMA accumulators per wave
//     Y = A^T M A          2x2 outputs per tile, per lane, straight from the accumulators; then the usual
//                          d * Y + noise + bias -> leaky-ReLU epilogue, float2 stores.
//
// Staging is the padded-flat q-space trick of csrc/modconv.hip with an EVEN row pitch P = W+2 (a zero column on both
// sides of every row, one shared zero row between images): the 4x4 patch of tile (img, ty, tx) starts at the even index
// q = (img*(H+1) + 2ty)*P + 2tx and its rows are P apart, so a block's 64 consecutive tiles need one contiguous q-range
// per channel, and every patch row is two 8-byte aligned ds_read_b64 (lanes = consecutive tiles = consecutive 8-byte
// words: conflict-free).  The weight fragments are stored [cin][cout][16] with the four 16-byte quads of a cout row
// XOR-swizzled by (cout & 3), so a lane fetches its 16 Winograd-domain weights with four conflict-free ds_read_b128.  One barrier per K stage (8 input channels = 64 MFMAs per wave), LDS stages
// double-buffered, next stage prefetched into registers under the MFMAs.
#include <stdlib.h>

#include "common.h"

namespace sgdfr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoParams {
    const float* x;
    int64_t x_bstride;
    const float* u;      // [Cin][16][Cout]
    const float* s;
    const float* d;
    const float* noise;
    int64_t noise_bstride;
    const float* noise_w;
    const float* bias;
    const float* zeros;  // >= 16 B of zeros (source of the padding positions)
    float* y;
    int B, Cin, Cout, H, W, P, R;
    int TW, TH;          // tiles per row / per column (W/2, H/2)
    int n_t_tiles, n_o_tiles;
    int total_tiles;
    int xs, xlen;
    int simgs;           // images a block of 64 tiles can touch
    int act;
    float slope, gain;
};

constexpr int WCK = 8;    // input channels per LDS stage (4 when three 8-channel stages do not fit in LDS)
constexpr int WNT = 64;   // couts per block
constexpr int WTT = 64;   // tiles per block
constexpr int WINO_SPLIT = 4;   // MFMAs issued before the first use of the prefetched fragments
constexpr int WEX = 4;    // staged q elements per thread per channel (xlen <= 1024)

__device__ __forceinline__ int wino_qbase(const WinoParams& p, int t) {
    const int per_img = p.TW * p.TH;
    const int img = t / per_img;
    const int rem = t - img * per_img;
    const int ty = rem / p.TW, tx = rem - ty * p.TW;
    return (img * p.R + 2 * ty) * p.P + 2 * tx;
}

// One k-pair's operands from LDS: the lane's 4x4 input patch (8 x ds_read_b64) and its 16 Winograd-domain weights
// (4 x ds_read_b128, quads un-swizzled by usw = cout & 3).
__device__ __forceinline__ void wino_fetch(const float* px, int P, const float* pu, int usw, float (&uu)[16],
                                           float (&dd)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(pu + ((j ^ usw) << 2));
        uu[4 * j + 0] = q.x; uu[4 * j + 1] = q.y; uu[4 * j + 2] = q.z; uu[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 a = *reinterpret_cast<const float2*>(px + r * P);
        const float2 b = *reinterpret_cast<const float2*>(px + r * P + 2);
        dd[r][0] = a.x; dd[r][1] = a.y; dd[r][2] = b.x; dd[r][3] = b.y;
    }
}

// V = sc * B^T d B for one 4x4 patch (32 add/sub + 16 mul per patch)
__device__ __forceinline__ void wino_input_transform(const float (&dd)[4][4], float sc, float (&vv)[16]) {
    float tmp[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // B^T d  (rows)
        tmp[0][c] = dd[0][c] - dd[2][c];
        tmp[1][c] = dd[1][c] + dd[2][c];
        tmp[2][c] = dd[2][c] - dd[1][c];
        tmp[3][c] = dd[1][c] - dd[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (.) B  (columns), then the style scale
        vv[a * 4 + 0] = (tmp[a][0] - tmp[a][2]) * sc;
        vv[a * 4 + 1] = (tmp[a][1] + tmp[a][2]) * sc;
        vv[a * 4 + 2] = (tmp[a][2] - tmp[a][1]) * sc;
        vv[a * 4 + 3] = (tmp[a][1] - tmp[a][3]) * sc;
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `per_stage * stages_in_flight` DMA instructions of this wave are outstanding
template <int CKK>
__device__ __forceinline__ void wait_dma(bool one_stage_in_flight) {
    // weight-slab DMA instructions per stage and wave: CKK (16-byte pieces of 1 KiB per wave)
    if (one_stage_in_flight) wait_vmcnt<CKK>();
    else wait_vmcnt<0>();
}

// CKK input channels per K stage, two LDS stages.  The k-pair pipeline runs CONTINUOUSLY across stages: the fragments
// of k-pair n+1 are fetched from LDS before the 16 MFMAs of k-pair n are issued, also when n+1 is the first k-pair of
// the next stage.  The single rendezvous per stage sits right before that cross-stage fetch; at that point every
// wave already holds the last fragments of stage st in registers, so the buffer of stage st is immediately free to
// receive stage st+2 (inputs: dword loads issued a whole stage earlier, then ds_write; weights: 16-byte LDS DMA,
// retired by a counted vmcnt at the next rendezvous).  With one wave per SIMD (16 accumulators = 256 registers)
// this keeps the matrix pipe from draining at stage boundaries.
template <int CKK, int NEX>
__global__ __launch_bounds__(256, 1) void wino_mfma_kernel(WinoParams p) {
    constexpr int NBUF = 2;
    constexpr int NKP = CKK / 2;   // MFMA k-pairs per stage
    constexpr int UV = CKK;   // weight-slab DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stage_floats = CKK * p.xs + CKK * 16 * WNT;
    float* ls = smem + NBUF * stage_floats;            // styles of the images this block touches: [simgs][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave >> 1, wt = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W;

    int lid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ct = lid / p.n_t_tiles, tt = lid - ct * p.n_t_tiles;
    const int n0 = ct * WNT;
    const int t0 = tt * WTT;
    const int q0 = wino_qbase(p, t0);
    const int per_img = p.TW * p.TH;
    const int img0 = t0 / per_img;

    // this lane's tile (column of the B operand) and its patch origin inside the staged range
    int tile = t0 + wt * 32 + l31;
    const bool tile_ok = tile < p.total_tiles;
    if (!tile_ok) tile = p.total_tiles - 1;
    const int boff = wino_qbase(p, tile) - q0;
    const float* lsp = ls + (tile / per_img - img0) * p.Cin + hi;   // this lane's style row (+ channel parity)
    const int uoff = (wo * 32 + l31) * 16, usw = l31 & 3;           // this lane's cout row in a [64][16] weight tile

    // staging descriptors: padding positions of the q-range read a zero word instead of x
    const float* xsrc[NEX];
#pragma unroll
    for (int e = 0; e < NEX; ++e) {
        const int j = tid + e * 256;
        const int q = q0 + j;
        const int pir = q / p.P;
        const int pc = q - pir * p.P;
        const int img = pir / p.R;
        const int pr = pir - img * p.R;
        const bool ok = (j < p.xlen) && pc >= 1 && pc <= p.W && pr >= 1 && img < p.B;
        xsrc[e] = ok ? p.x + (int64_t)img * p.x_bstride + (pr - 1) * p.W + (pc - 1) : nullptr;
    }

    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;

    // Global -> LDS by DMA (no staging registers, no arithmetic: the style scale is applied after the input
    // transform, which is linear).  dwordx4 for the weight slab, dword for the q-range (zero word for padding).
    // Staging.  Weight slab: global -> LDS DMA in 16-byte pieces (1 KiB per wave instruction, no registers).
    // q-range of the inputs: ordinary dword loads into registers two stages ahead, written to LDS after the MFMAs
    // (a 4-byte DMA piece costs as much issue time as a 16-byte one, so the narrow gather goes through VGPRs).
    float xr[NEX][CKK];
    auto load_x = [&](int st) {
        const int c0 = st * CKK;
#pragma unroll
        for (int e = 0; e < NEX; ++e) {
            const float* src = xsrc[e] ? xsrc[e] + (int64_t)c0 * HW : p.zeros;
            const int64_t cs = xsrc[e] ? HW : 0;
#pragma unroll
            for (int c = 0; c < CKK; ++c) xr[e][c] = src[c * cs];
        }
    };
    auto store_x = [&](int st) {
        float* lx = smem + (st % NBUF) * stage_floats;
#pragma unroll
        for (int e = 0; e < NEX; ++e)
#pragma unroll
            for (int c = 0; c < CKK; ++c) lx[c * p.xs + tid + e * 256] = xr[e][c];
    };
    auto issue_u = [&](int st) {
        const int c0 = st * CKK;
        float* lu = smem + (st % NBUF) * stage_floats + CKK * p.xs;
#pragma unroll
        for (int v = 0; v < UV; ++v) {
            const int f = tid + v * 256;                  // float4 index inside the stage slab: 256 per channel
            const int row = f >> 8, col = (f & 255) * 4;
            const float* src = p.u + (((int64_t)(c0 + row) * p.Cout + n0) << 4) + col;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lu + (size_t)((tid & ~63) + v * 256) * 4), 16, 0, 0);
        }
    };

    const int nstage = p.Cin / CKK;
    const int P = p.P;
    // styles of the block's images -> LDS (plain loads: retired by the first wait below)
    for (int e = tid; e < p.simgs * p.Cin; e += 256) {
        const int m = e / p.Cin;
        ls[e] = (img0 + m < p.B) ? p.s[(int64_t)(img0 + m) * p.Cin + (e - m * p.Cin)] : 0.f;
    }
    load_x(0);
    store_x(0);
    issue_u(0);
    if (nstage > 1) {
        load_x(1);
        store_x(1);
        issue_u(1);
        wait_vmcnt<CKK>();     // stage 0's weight DMA landed; stage 1's may still fly
    } else {
        wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // style rows + staged inputs written above
    __builtin_amdgcn_s_barrier();

    // Input transform of the upcoming k-pair:  V = s * B^T x B   (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
    float uc[16], vc[16], un[16], dn[4][4];
    {
        wino_fetch(smem + hi * p.xs + boff, P, smem + CKK * p.xs + hi * (16 * WNT) + uoff, usw, uc, dn);
        wino_input_transform(dn, lsp[0], vc);
    }
    for (int st = 0; st < nstage; ++st) {
        const bool more = st + 2 < nstage, next_stage = st + 1 < nstage;
        if (more) load_x(st + 2);                      // global -> registers, consumed after the rendezvous below
#pragma unroll
        for (int cp = 0; cp < NKP; ++cp) {
            const bool cross = (cp == NKP - 1);
            const bool fetch = !cross || next_stage;
            float sn = 0.f;
            if (cross && next_stage) {
                // rendezvous: stage st+1 complete for every wave (weights: DMA issued one stage ago; inputs: ds_write)
                if (more) wait_vmcnt<NEX * CKK>();      // all but the input loads issued at the top of this iteration
                else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (more) {                             // buffer of stage st is free: every wave has its last fragments
                    store_x(st + 2);
                    issue_u(st + 2);
                }
            }
            if (fetch) {
                const int fst = cross ? st + 1 : st, fcp = cross ? 0 : cp + 1;
                const float* lx = smem + (fst % NBUF) * stage_floats;
                wino_fetch(lx + (fcp * 2 + hi) * p.xs + boff, P, lx + CKK * p.xs + (fcp * 2 + hi) * (16 * WNT) + uoff, usw, un,
                           dn);
                sn = lsp[fst * CKK + fcp * 2];
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the fetches above the MFMAs
#pragma unroll
            for (int k = 0; k < WINO_SPLIT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[k], vc[k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);           // first use of the fetched fragments comes after these MFMAs
            float vn[16];
            if (fetch) wino_input_transform(dn, sn, vn);
#pragma unroll
            for (int k = WINO_SPLIT; k < 16; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[k], vc[k], acc[k], 0, 0, 0);
            if (fetch) {
#pragma unroll
                for (int k = 0; k < 16; ++k) { uc[k] = un[k]; vc[k] = vn[k]; }
            }
        }
    }

    // ---- output transform Y = A^T M A per lane, then the StyledConv epilogue
    if (!tile_ok) return;
    const int64_t img = tile / per_img;
    const int rem = tile - (int)img * per_img;
    const int ty = rem / p.TW, tx = rem - ty * p.TW;
    const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
    float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (p.noise) {
        const float* np = p.noise + img * p.noise_bstride + (2 * ty) * p.W + 2 * tx;
        nz[0][0] = nw * np[0]; nz[0][1] = nw * np[1];
        nz[1][0] = nw * np[p.W]; nz[1][1] = nw * np[p.W + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = n0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= p.Cout) continue;
        float m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = acc[k][r];
        float ra[2][4];   // A^T M  (rows): [0] = m0+m1+m2, [1] = m1-m2-m3, per column
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ra[0][c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
            ra[1][c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
        }
        float yv[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            yv[a][0] = ra[a][0] + ra[a][1] + ra[a][2];
            yv[a][1] = ra[a][1] - ra[a][2] - ra[a][3];
        }
        const float dv = p.d ? p.d[img * p.Cout + co] : 1.f;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float* dst = p.y + (img * p.Cout + co) * HW + (2 * ty) * p.W + 2 * tx;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v0 = yv[a][0] * dv + nz[a][0] + bv;
            float v1 = yv[a][1] * dv + nz[a][1] + bv;
            if (p.act) {
                v0 = lrelu_gain(v0, p.slope, p.gain);
                v1 = lrelu_gain(v1, p.slope, p.gain);
            }
            *reinterpret_cast<float2*>(dst + a * p.W) = make_float2(v0, v1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-xi variant: 8 waves per block, TWO waves per SIMD.  A pair of waves shares one (32 cout x 32 tile) quadrant
// and splits the 16 Winograd-domain positions: half h owns rows a = 2h, 2h+1 of the 4x4 domain, i.e. 8 accumulators
// (128 registers) instead of 16.  An in-order wave pays for its own LDS reads / transform VALU / staging with matrix
// issue slots, but a SIMD partner's work overlaps perfectly (scripts/mfma_probe.hip), so two 8-accumulator waves per
// SIMD keep the fp32 matrix pipe busier than one 16-accumulator wave.  Each half needs only two rows of B^T x B
// (24 VALU instead of 80 per k-pair) and two of the four weight quads.  The output transform is linear in the domain
// rows, so each half reduces its rows to a partial 2x2 result; half 1 hands it to half 0 through LDS once, after the K
// loop.
__device__ __forceinline__ void wino_fetch_half(const float* px, int P, const float* pu, int usw, int half, float (&uu)[8],
                                                float (&dd)[4][4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(pu + (((2 * half + j) ^ usw) << 2));
        uu[4 * j + 0] = q.x; uu[4 * j + 1] = q.y; uu[4 * j + 2] = q.z; uu[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 a = *reinterpret_cast<const float2*>(px + r * P);
        const float2 b = *reinterpret_cast<const float2*>(px + r * P + 2);
        dd[r][0] = a.x; dd[r][1] = a.y; dd[r][2] = b.x; dd[r][3] = b.y;
    }
}

// rows a = 2*half, 2*half+1 of  sc * B^T d B
__device__ __forceinline__ void wino_input_transform_half(const float (&dd)[4][4], float sc, int half, float (&vv)[8]) {
    float t0[4], t1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        t0[c] = half ? dd[2][c] - dd[1][c] : dd[0][c] - dd[2][c];
        t1[c] = half ? dd[1][c] - dd[3][c] : dd[1][c] + dd[2][c];
    }
    vv[0] = (t0[0] - t0[2]) * sc; vv[1] = (t0[1] + t0[2]) * sc; vv[2] = (t0[2] - t0[1]) * sc; vv[3] = (t0[1] - t0[3]) * sc;
    vv[4] = (t1[0] - t1[2]) * sc; vv[5] = (t1[1] + t1[2]) * sc; vv[6] = (t1[2] - t1[1]) * sc; vv[7] = (t1[1] - t1[3]) * sc;
}

template <int CKK, int NEX>   // NEX: 512-element rows of the staged q-range per channel
__global__ __launch_bounds__(512, 2) void wino2_mfma_kernel(WinoParams p) {
    constexpr int NBUF = 2;
    constexpr int NKP = CKK / 2;
    constexpr int UV = CKK / 2;   // weight-slab DMA pieces per wave and stage (CKK*16*64 floats / 4 / 512 threads)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stage_floats = CKK * p.xs + CKK * 16 * WNT;
    float* ls = smem + NBUF * stage_floats;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = wave & 3, half = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int wo = quad >> 1, wt = quad & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int HW = p.H * p.W;

    int lid;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int ct = lid / p.n_t_tiles, tt = lid - ct * p.n_t_tiles;
    const int n0 = ct * WNT;
    const int t0 = tt * WTT;
    const int q0 = wino_qbase(p, t0);
    const int per_img = p.TW * p.TH;
    const int img0 = t0 / per_img;

    int tile = t0 + wt * 32 + l31;
    const bool tile_ok = tile < p.total_tiles;
    if (!tile_ok) tile = p.total_tiles - 1;
    const int boff = wino_qbase(p, tile) - q0;
    const float* lsp = ls + (tile / per_img - img0) * p.Cin + hi;
    const int uoff = (wo * 32 + l31) * 16, usw = l31 & 3;

    const float* xsrc[NEX];
#pragma unroll
    for (int e = 0; e < NEX; ++e) {
        const int j = tid + e * 512;
        const int q = q0 + j;
        const int pir = q / p.P;
        const int pc = q - pir * p.P;
        const int img = pir / p.R;
        const int pr = pir - img * p.R;
        const bool ok = (j < p.xlen) && pc >= 1 && pc <= p.W && pr >= 1 && img < p.B;
        xsrc[e] = ok ? p.x + (int64_t)img * p.x_bstride + (pr - 1) * p.W + (pc - 1) : nullptr;
    }

    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    float xr[NEX][CKK];
    auto load_x = [&](int st) {
        const int c0 = st * CKK;
#pragma unroll
        for (int e = 0; e < NEX; ++e) {
            const float* src = xsrc[e] ? xsrc[e] + (int64_t)c0 * HW : p.zeros;
            const int64_t cs = xsrc[e] ? HW : 0;
#pragma unroll
            for (int c = 0; c < CKK; ++c) xr[e][c] = src[c * cs];
        }
    };
    auto store_x = [&](int st) {
        float* lx = smem + (st % NBUF) * stage_floats;
#pragma unroll
        for (int e = 0; e < NEX; ++e)
#pragma unroll
            for (int c = 0; c < CKK; ++c) lx[c * p.xs + tid + e * 512] = xr[e][c];
    };
    auto issue_u = [&](int st) {
        const int c0 = st * CKK;
        float* lu = smem + (st % NBUF) * stage_floats + CKK * p.xs;
#pragma unroll
        for (int v = 0; v < UV; ++v) {
            const int f = tid + v * 512;                  // float4 index inside the stage slab: 256 per channel
            const int row = f >> 8, col = (f & 255) * 4;
            const float* src = p.u + (((int64_t)(c0 + row) * p.Cout + n0) << 4) + col;
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(lu + (size_t)((tid & ~63) + v * 512) * 4), 16, 0, 0);
        }
    };

    const int nstage = p.Cin / CKK;
    const int P = p.P;
    for (int e = tid; e < p.simgs * p.Cin; e += 512) {
        const int m = e / p.Cin;
        ls[e] = (img0 + m < p.B) ? p.s[(int64_t)(img0 + m) * p.Cin + (e - m * p.Cin)] : 0.f;
    }
    load_x(0);
    store_x(0);
    issue_u(0);
    if (nstage > 1) {
        load_x(1);
        store_x(1);
        issue_u(1);
        wait_vmcnt<UV>();
    } else {
        wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float uc[8], vc[8], un[8], dn[4][4];
    wino_fetch_half(smem + hi * p.xs + boff, P, smem + CKK * p.xs + hi * (16 * WNT) + uoff, usw, half, uc, dn);
    wino_input_transform_half(dn, lsp[0], half, vc);
    for (int st = 0; st < nstage; ++st) {
        const bool more = st + 2 < nstage, next_stage = st + 1 < nstage;
        if (more) load_x(st + 2);
#pragma unroll
        for (int cp = 0; cp < NKP; ++cp) {
            const bool cross = (cp == NKP - 1);
            const bool fetch = !cross || next_stage;
            float sn = 0.f;
            if (cross && next_stage) {
                if (more) wait_vmcnt<NEX * CKK>();
                else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (more) {
                    store_x(st + 2);
                    issue_u(st + 2);
                }
            }
            if (fetch) {
                const int fst = cross ? st + 1 : st, fcp = cross ? 0 : cp + 1;
                const float* lx = smem + (fst % NBUF) * stage_floats;
                wino_fetch_half(lx + (fcp * 2 + hi) * p.xs + boff, P, lx + CKK * p.xs + (fcp * 2 + hi) * (16 * WNT) + uoff, usw,
                                half, un, dn);
                sn = lsp[fst * CKK + fcp * 2];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[k], vc[k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float vn[8];
            if (fetch) wino_input_transform_half(dn, sn, half, vn);
#pragma unroll
            for (int k = 2; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[k], vc[k], acc[k], 0, 0, 0);
            if (fetch) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { uc[k] = un[k]; vc[k] = vn[k]; }
            }
        }
    }

    // ---- partial output transform of this half's two domain rows:  Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
    //      half 0 (rows 0,1): ra0 = m0 + m1, ra1 = m1 ;  half 1 (rows 2,3): ra0 = m2, ra1 = -m2 - m3
    float py[16][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float ra0[4], ra1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ma = acc[c][r], mb = acc[4 + c][r];
            ra0[c] = half ? ma : ma + mb;
            ra1[c] = half ? -ma - mb : mb;
        }
        py[r][0] = ra0[0] + ra0[1] + ra0[2];
        py[r][1] = ra0[1] - ra0[2] - ra0[3];
        py[r][2] = ra1[0] + ra1[1] + ra1[2];
        py[r][3] = ra1[1] - ra1[2] - ra1[3];
    }
    __syncthreads();                       // every wave is done with the staging buffers
    float* ex = smem + quad * (64 * 64);   // [r*4+i][lane] per quadrant: 16 KB
    if (half) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[(r * 4 + i) * 64 + lane] = py[r][i];
    }
    // coefficients of the epilogue go through LDS too: a global `d` / bias load between two stores would make the
    // wave wait for the previous store's HBM round trip (loads and stores share the in-order vmcnt counter)
    float* const dl = smem + 4 * 64 * 64;      // [simgs][WNT]
    float* const bl = dl + p.simgs * WNT;      // [WNT]
    for (int e = tid; e < p.simgs * WNT; e += 512) {
        const int m = e / WNT, c = e - m * WNT;
        dl[e] = (p.d && img0 + m < p.B && n0 + c < p.Cout) ? p.d[(int64_t)(img0 + m) * p.Cout + n0 + c] : 1.f;
    }
    for (int e = tid; e < WNT; e += 512) bl[e] = (p.bias && n0 + e < p.Cout) ? p.bias[n0 + e] : 0.f;
    const int64_t img = tile / per_img;
    const int rem = tile - (int)img * per_img;
    const int ty = rem / p.TW, tx = rem - ty * p.TW;
    const float nw = (p.noise && p.noise_w) ? p.noise_w[0] : 0.f;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.noise && !half) {
        const float* np = p.noise + img * p.noise_bstride + (2 * ty) * p.W + 2 * tx;
        nz[0] = nw * np[0]; nz[1] = nw * np[1]; nz[2] = nw * np[p.W]; nz[3] = nw * np[p.W + 1];
    }
    __syncthreads();
    if (half || !tile_ok) return;
    const float* dln = dl + ((int)img - img0) * WNT + wo * 32 + 4 * hi;
    const float* bln = bl + wo * 32 + 4 * hi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = (r & 3) + 8 * (r >> 2);
        const int co = n0 + wo * 32 + cl + 4 * hi;
        if (co >= p.Cout) continue;
        const float dv = dln[cl];
        const float bv = bln[cl];
        float* dst = p.y + (img * p.Cout + co) * HW + (2 * ty) * p.W + 2 * tx;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (py[r][i] + ex[(r * 4 + i) * 64 + lane]) * dv + nz[i] + bv;
            if (p.act) v[i] = lrelu_gain(v[i], p.slope, p.gain);
        }
        *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(dst + p.W) = make_float2(v[2], v[3]);
    }
}

// U[i][o][xi] = (G g G^T)[xi] * scale (quads swizzled, see file header),  g = weight[o][i][3][3]  (optionally rotated 180 degrees and read
// transposed, which turns the pack into the one of the adjoint conv: weight is then indexed [i][o])
__global__ __launch_bounds__(256) void prepack_wino_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout,
                                                          int Cin, int transpose_flip, float scale) {
    // output dims: rows = "in" channels of this conv, cols = "out" channels
    const int n_in = transpose_flip ? Cout : Cin, n_out = transpose_flip ? Cin : Cout;
    const int64_t n = (int64_t)n_in * n_out;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % n_out);
        const int i = (int)(idx / n_out);
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                g[ky][kx] = (transpose_flip ? w[((int64_t)i * Cin + o) * 9 + (2 - ky) * 3 + (2 - kx)]
                                            : w[((int64_t)o * Cin + i) * 9 + ky * 3 + kx]) * scale;
        float t[4][3];   // G g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t[0][c] = g[0][c];
            t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
            t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
            t[3][c] = g[2][c];
        }
        // row of 16 Winograd-domain weights for (i, o); quad a is stored at slot a ^ (o & 3)
        float* dst = u + (((int64_t)i * n_out + o) << 4);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 q;
            q.x = t[a][0];
            q.y = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
            q.z = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
            q.w = t[a][2];
            *reinterpret_cast<float4*>(dst + ((a ^ (o & 3)) << 2)) = q;
        }
    }
}

}  // namespace sgdfr

using namespace sgdfr;

// Largest q-distance between the first and last patch origin of 64 consecutive tiles starting at a multiple of 64.
static int wino_span(int H, int W) {
    const int TW = W / 2, TH = H / 2, P = W + 2;
    const int per_img = TW * TH;
    if (TW % WTT == 0) return 2 * (WTT - 1);                                            // inside one tile row
    if (WTT % TW == 0 && per_img % WTT == 0) return (WTT / TW - 1) * 2 * P + 2 * (TW - 1);  // whole tile rows of one image
    const int rows_crossed = (WTT - 1) / TW + 1, imgs_crossed = (WTT - 1) / per_img + 1;
    return rows_crossed * 2 * P + imgs_crossed * P + 2 * (TW - 1);   // a row step is 2P, an image step 3P
}

extern "C" int sgdfr_modconv_prepack_wino_f32(const float* weight, float* u, int Cout, int Cin, int transpose_flip,
                                              void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0, "prepack_wino: bad shape %d %d", Cout, Cin);
    SGDFR_REQUIRE(weight && u, "prepack_wino: null pointer");
    const int64_t n = (int64_t)Cout * Cin;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prepack_wino_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, u, Cout, Cin,
                       transpose_flip, 1.0f / sqrtf((float)Cin * 9));
    return check_launch("modconv_prepack_wino");
}

extern "C" int sgdfr_modconv2d_wino_supported(int B, int Cin, int Cout, int H, int W) {
    if (Cin % WCK != 0 || Cout % WNT != 0 || (H & 1) || (W & 1) || H < 2 || W < 2) return 0;
    // staged q-range of 64 consecutive tiles must fit WEX*256 elements
    return (wino_span(H, W) + 3 * (W + 2) + 4) <= WEX * 256 ? 1 : 0;
}

extern "C" int sgdfr_modconv2d_wino_f32(const float* x, int64_t x_bstride, const float* u, const float* s, const float* d,
                                        const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                                        const float* zeros, float* y, int B, int Cin, int Cout, int H, int W, int act,
                                        float slope, float gain, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "modconv_wino: bad shape B=%d Cin=%d Cout=%d H=%d W=%d",
                  B, Cin, Cout, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(sgdfr_modconv2d_wino_supported(B, Cin, Cout, H, W), "modconv_wino: shape not supported "
                  "(needs Cin %% 8 == 0, Cout %% 64 == 0, even H and W); use sgdfr_modconv2d_fwd_f32");
    SGDFR_REQUIRE(x && u && s && y && zeros, "modconv_wino: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "modconv_wino: noise without noise_w");
    SGDFR_REQUIRE(((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  "modconv_wino: s, u and y must be 16-byte aligned");
    WinoParams p{};
    p.x = x; p.x_bstride = x_bstride; p.u = u; p.s = s; p.d = d; p.noise = noise; p.noise_bstride = noise_bstride;
    p.noise_w = noise_w; p.bias = bias; p.zeros = zeros; p.y = y;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.P = W + 2; p.R = H + 1;
    p.TW = W / 2; p.TH = H / 2;
    p.total_tiles = B * p.TW * p.TH;
    SGDFR_REQUIRE((int64_t)B * p.R * p.P + 4ll * p.P + 8 < (1ll << 31), "modconv_wino: batch too large for 32-bit indices");
    p.n_t_tiles = (p.total_tiles + WTT - 1) / WTT;
    p.n_o_tiles = Cout / WNT;
    p.xlen = wino_span(H, W) + 3 * p.P + 4;
    p.xs = (p.xlen + 255) & ~255;   // whole 256-element DMA rows: stray lanes of the last row stay inside the channel
    p.act = act; p.slope = slope; p.gain = gain;
    const int per_img = p.TW * p.TH;
    p.simgs = (per_img % WTT == 0) ? 1 : (WTT - 1) / per_img + 2;
    const size_t s_bytes = (size_t)((p.simgs * Cin + 3) & ~3) * sizeof(float);
    static const int variant = getenv("SGDFR_WINO") ? atoi(getenv("SGDFR_WINO")) : 2;   // 1: one 16-accumulator wave per SIMD
    void (*kern)(WinoParams);
    size_t lds;
    int threads;
    if (variant == 1) {
        const int nex = (p.xlen + 255) / 256;
        p.xs = nex * 256;                               // the q-range is staged in whole 256-element rows
        const size_t lds8 = 2 * (size_t)(8 * p.xs + 8 * 16 * WNT) * sizeof(float) + s_bytes;
        const size_t lds4 = 2 * (size_t)(4 * p.xs + 4 * 16 * WNT) * sizeof(float) + s_bytes;
        const bool use8 = lds8 <= 160 * 1024;
        lds = use8 ? lds8 : lds4;
        if (use8) kern = nex <= 1 ? wino_mfma_kernel<8, 1> : nex == 2 ? wino_mfma_kernel<8, 2> : nex == 3 ? wino_mfma_kernel<8, 3> : wino_mfma_kernel<8, 4>;
        else kern = nex <= 2 ? wino_mfma_kernel<4, 2> : nex == 3 ? wino_mfma_kernel<4, 3> : wino_mfma_kernel<4, 4>;
        threads = 256;
    } else {
        const int nex = (p.xlen + 511) / 512;           // 512 staging threads
        p.xs = nex * 512;
        lds = 2 * (size_t)(8 * p.xs + 8 * 16 * WNT) * sizeof(float) + s_bytes;
        const size_t epi = (4 * 64 * 64 + (size_t)(p.simgs + 1) * WNT) * sizeof(float);   // half-exchange area + d / bias rows of the epilogue
        if (lds < epi) lds = epi;
        kern = nex <= 1 ? wino2_mfma_kernel<8, 1> : wino2_mfma_kernel<8, 2>;
        threads = 512;
    }
    SGDFR_REQUIRE(lds <= 160 * 1024, "modconv_wino: LDS request %zu too large", lds);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return check_launch("modconv_wino(lds attribute)");
    hipLaunchKernelGGL(kern, dim3(p.n_t_tiles * p.n_o_tiles), dim3(threads), lds, as_stream(stream), p);
    return check_launch("modconv2d_wino");
}
