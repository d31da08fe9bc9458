// Backward-pass helpers of the generator path.  The heavy part of every gradient --
// dL/dx of a modulated conv -- is the SAME fp32-MFMA kernel as the forward (csrc/modconv.hip) run with a
// transposed weight pack and the roles of s and d exchanged; this file holds the HBM-bound pieces around it:
//   * activation gradient + the per-(b,c) reductions that feed d(bias), d(noise strength), d(demod)
//   * adjoint of the 4x4 blur, emitted directly as parity planes for MODE_DOWN3
//   * dx = s * gu with the reduction sum_q x*gu that is d(style) of the modulation
//   * ToRGB backward (dx and the 3 x Cin reductions that give d(style) and d(w_rgb))
//   * transposed weight pack
// Reductions use one wave per (b, c) plane chunk + fp32 atomics into zero-initialised buffers.
#include <stdlib.h>

#include "common.h"

namespace sgdfr {

constexpr int kChunk = 2048;  // elements of one plane handled by one wave

// g_pre = g_out * (out > 0 ? 1 : slope) * gain ;  sums[b,c,0] += sum g_pre ; [1] += sum g_pre*noise ;
// [2] += sum g_pre * y, with y = act^-1(out) - noise_w*noise - bias  (the demodulated conv output d*v)
__global__ __launch_bounds__(256) void act_grad_reduce_kernel(const float* __restrict__ g_out,
                                                             const float* __restrict__ out,
                                                             const float* __restrict__ noise, int64_t noise_bstride,
                                                             const float* __restrict__ noise_w,
                                                             const float* __restrict__ bias, float* __restrict__ g_pre,
                                                             float* __restrict__ sums, int B, int C, int HW, int chunks,
                                                             float slope, float gain, int want_y,
                                                             unsigned* __restrict__ g_absmax) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nw_total = (int64_t)B * C * chunks;
    if (wid >= nw_total) return;
    const int ch = (int)(wid % chunks);
    const int64_t pl = wid / chunks;
    const int c = (int)(pl % C), b = (int)(pl / C);
    const float nwv = (noise && noise_w) ? noise_w[0] : 0.f;
    const float bv = bias ? bias[c] : 0.f;
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * slope);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    unsigned gm = 0u;            // bit pattern of max |g_pre| over this wave's elements (the range plan of the fp16-split dL/dx conv)
    const int lo = ch * kChunk, hi = min(HW, lo + kChunk);
    const int64_t base = pl * HW;
    for (int p = lo + lane; p < hi; p += 64) {
        const float o = out[base + p], g = g_out[base + p];
        const float gp = g * (o > 0.f ? 1.f : slope) * gain;
        g_pre[base + p] = gp;
        gm = max(gm, __float_as_uint(fabsf(gp)));
        const float nz = noise ? noise[(int64_t)b * noise_bstride + p] : 0.f;
        s0 += gp;
        s1 = fmaf(gp, nz, s1);
        if (want_y) {
            const float pre = o * (o > 0.f ? inv_pos : inv_neg);
            s2 = fmaf(gp, pre - nwv * nz - bv, s2);
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        atomicAdd(&sums[pl * 3 + 0], s0);
        atomicAdd(&sums[pl * 3 + 1], s1);
        if (want_y) atomicAdd(&sums[pl * 3 + 2], s2);
    }
    if (g_absmax) {
        // one word per (image, channel) plane, like the sums: the waves of a plane (<= 32) share it; a word per IMAGE would take
        // thousands of same-address atomics, which serialise in L2 (see absmax_kernel, linear.hip)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));
        if (lane == 0 && gm != 0u) atomicMax(g_absmax + pl, gm);
    }
}

// The backward of the FROZEN generator (whole-synthesis Function, autograd.SynthesisFrozenFn) walks a saved activation `out`
// ONCE: its gradient arrives as up to three terms that the per-layer Functions materialised separately --
//     g = gu * s_next                                  (scale_reduce of the conv that read `out`: dL/d(x*s) times its style)
//       + s_rgb * sum_j (w_rgb[j,c] * rgb_scale) g_rgb[b,j]    (torgb_bwd of the ToRGB that read it)
//       + g_add                                        (anything else, e.g. a caller-side gradient)
// and the same pass yields both style reductions that need `out` as x (r_next = sum out*gu, r_rgb[j] = sum out*g_rgb[j]) and
// everything act_grad_reduce computes for the layer that PRODUCED `out` (g_pre, the three sums, max |g_pre| per plane).
// Per element 12-16 bytes instead of the 36 of scale_reduce + torgb_bwd + autograd's add + act_grad_reduce; the
// expressions per term are those kernels' (a two-term sum is commutative: same bits as the per-layer path).
// VEC = 4: float4 per lane (HW % 4 == 0, the generator's planes are powers of two), VEC = 1: any HW.
template <int VEC>
__global__ __launch_bounds__(256) void grad_join_kernel(const float* __restrict__ out, const float* __restrict__ gu,
                                                       const float* __restrict__ s_next, const float* __restrict__ g_rgb,
                                                       const float* __restrict__ w_rgb, const float* __restrict__ s_rgb,
                                                       float rgb_scale, const float* __restrict__ g_add,
                                                       const float* __restrict__ noise, int64_t noise_bstride,
                                                       const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                       float* __restrict__ g_pre, float* __restrict__ sums,
                                                       unsigned* __restrict__ g_absmax, float* __restrict__ r_next,
                                                       float* __restrict__ r_rgb, int B, int C, int HW, int chunks, float slope,
                                                       float gain, int want_y) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * C * chunks) return;
    const int ch = (int)(wid % chunks);
    const int64_t pl = wid / chunks;
    const int c = (int)(pl % C), b = (int)(pl / C);
    const float nwv = (noise && noise_w) ? noise_w[0] : 0.f;
    const float bv = bias ? bias[c] : 0.f;
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * slope);
    const float sn = gu ? s_next[pl] : 0.f;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, sr = 0.f;
    if (g_rgb) {
        c0 = w_rgb[c] * rgb_scale; c1 = w_rgb[C + c] * rgb_scale; c2 = w_rgb[2 * C + c] * rgb_scale;
        sr = s_rgb[pl];
    }
    const float* gb = g_rgb ? g_rgb + (int64_t)b * 3 * HW : nullptr;
    const float* nzb = noise ? noise + (int64_t)b * noise_bstride : nullptr;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, rn = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
    unsigned gm = 0u;
    const int lo = ch * kChunk, hi = min(HW, lo + kChunk);
    const int64_t base = pl * HW;
    for (int p = lo + lane * VEC; p < hi; p += 64 * VEC) {
        float o[VEC], g[VEC], nz[VEC];
        if (VEC == 4) {
            *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(out + base + p);
            if (nzb) *reinterpret_cast<float4*>(nz) = *reinterpret_cast<const float4*>(nzb + p);
        } else {
            o[0] = out[base + p];
            if (nzb) nz[0] = nzb[p];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) g[v] = 0.f;
        if (gu) {
            float u[VEC];
            if (VEC == 4) *reinterpret_cast<float4*>(u) = *reinterpret_cast<const float4*>(gu + base + p);
            else u[0] = gu[base + p];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                rn = fmaf(o[v], u[v], rn);
                g[v] = u[v] * sn;
            }
        }
        if (gb) {
            float g0[VEC], g1[VEC], g2[VEC];
            if (VEC == 4) {
                *reinterpret_cast<float4*>(g0) = *reinterpret_cast<const float4*>(gb + p);
                *reinterpret_cast<float4*>(g1) = *reinterpret_cast<const float4*>(gb + HW + p);
                *reinterpret_cast<float4*>(g2) = *reinterpret_cast<const float4*>(gb + 2 * (int64_t)HW + p);
            } else {
                g0[0] = gb[p]; g1[0] = gb[HW + p]; g2[0] = gb[2 * (int64_t)HW + p];
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float t = sr * (c0 * g0[v] + c1 * g1[v] + c2 * g2[v]);
                g[v] = gu ? g[v] + t : t;
                q0 = fmaf(o[v], g0[v], q0);
                q1 = fmaf(o[v], g1[v], q1);
                q2 = fmaf(o[v], g2[v], q2);
            }
        }
        if (g_add) {
            float a[VEC];
            if (VEC == 4) *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(g_add + base + p);
            else a[0] = g_add[base + p];
#pragma unroll
            for (int v = 0; v < VEC; ++v) g[v] = (gu || gb) ? g[v] + a[v] : a[v];
        }
        float gp[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            gp[v] = g[v] * (o[v] > 0.f ? 1.f : slope) * gain;
            gm = max(gm, __float_as_uint(fabsf(gp[v])));
            const float nzv = nzb ? nz[v] : 0.f;
            s0 += gp[v];
            s1 = fmaf(gp[v], nzv, s1);
            if (want_y) {
                const float pre = o[v] * (o[v] > 0.f ? inv_pos : inv_neg);
                s2 = fmaf(gp[v], pre - nwv * nzv - bv, s2);
            }
        }
        if (VEC == 4) *reinterpret_cast<float4*>(g_pre + base + p) = *reinterpret_cast<const float4*>(gp);
        else g_pre[base + p] = gp[0];
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    rn = wave_sum(rn); q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));
    if (lane == 0) {
        atomicAdd(&sums[pl * 3 + 0], s0);
        atomicAdd(&sums[pl * 3 + 1], s1);
        if (want_y) atomicAdd(&sums[pl * 3 + 2], s2);
        if (gu) atomicAdd(&r_next[pl], rn);
        if (gb) {
            float* rb = r_rgb + (int64_t)b * 3 * C;
            atomicAdd(&rb[c], q0);
            atomicAdd(&rb[C + c], q1);
            atomicAdd(&rb[2 * C + c], q2);
        }
        if (g_absmax && gm != 0u) atomicMax(g_absmax + pl, gm);
    }
}

// The same pass for activations that a ToRGB reads (float4 path): a wave owns GJ_CPW consecutive channels of one (image, chunk)
// and keeps that chunk of the three RGB-gradient planes in registers (96 VGPRs) -- one wave per channel re-read them for every
// channel (20 B of L2 traffic per element on the last layer: 3.2 TB/s effective against 4.7 on the layers without the RGB term).
// Per element the expressions and their order are grad_join_kernel's: same g_pre bits.
constexpr int GJ_CPW = 4;
__global__ __launch_bounds__(256) void grad_join_rgb_kernel(const float* __restrict__ out, const float* __restrict__ gu,
                                                           const float* __restrict__ s_next, const float* __restrict__ g_rgb,
                                                           const float* __restrict__ w_rgb, const float* __restrict__ s_rgb,
                                                           float rgb_scale, const float* __restrict__ noise, int64_t noise_bstride,
                                                           const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                           float* __restrict__ g_pre, float* __restrict__ sums,
                                                           unsigned* __restrict__ g_absmax, float* __restrict__ r_next,
                                                           float* __restrict__ r_rgb, int B, int C, int HW, int chunks, float slope,
                                                           float gain, int want_y) {
    constexpr int IT = kChunk / 256;          // float4 iterations of a chunk
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cgroups = (C + GJ_CPW - 1) / GJ_CPW;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * cgroups * chunks) return;
    const int ch = (int)(wid % chunks);
    const int64_t t = wid / chunks;
    const int cg = (int)(t % cgroups), b = (int)(t / cgroups);
    const int lo = ch * kChunk, hi = min(HW, lo + kChunk);
    const float* gb = g_rgb + (int64_t)b * 3 * HW;
    const float* nzb = noise ? noise + (int64_t)b * noise_bstride : nullptr;
    const float nwv = (noise && noise_w) ? noise_w[0] : 0.f;
    const float inv_pos = 1.f / gain, inv_neg = 1.f / (gain * slope);
    float4 r0[IT], r1[IT], r2[IT], nz4[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int p = lo + (it * 64 + lane) * 4;
        const bool in = p < hi;
        r0[it] = in ? *reinterpret_cast<const float4*>(gb + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        r1[it] = in ? *reinterpret_cast<const float4*>(gb + HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        r2[it] = in ? *reinterpret_cast<const float4*>(gb + 2 * (int64_t)HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        nz4[it] = (in && nzb) ? *reinterpret_cast<const float4*>(nzb + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int cc = 0; cc < GJ_CPW; ++cc) {
        const int c = cg * GJ_CPW + cc;
        if (c >= C) break;
        const int64_t pl = (int64_t)b * C + c;
        const float bv = bias ? bias[c] : 0.f;
        const float sn = gu ? s_next[pl] : 0.f;
        const float c0 = w_rgb[c] * rgb_scale, c1 = w_rgb[C + c] * rgb_scale, c2 = w_rgb[2 * C + c] * rgb_scale;
        const float sr = s_rgb[pl];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, rn = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
        unsigned gm = 0u;
        const int64_t base = pl * HW;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int p = lo + (it * 64 + lane) * 4;
            if (p >= hi) continue;
            float o[4], g[4], gp[4];
            *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(out + base + p);
            const float g0[4] = {r0[it].x, r0[it].y, r0[it].z, r0[it].w}, g1[4] = {r1[it].x, r1[it].y, r1[it].z, r1[it].w},
                        g2[4] = {r2[it].x, r2[it].y, r2[it].z, r2[it].w}, nz[4] = {nz4[it].x, nz4[it].y, nz4[it].z, nz4[it].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) g[v] = 0.f;
            if (gu) {
                float u[4];
                *reinterpret_cast<float4*>(u) = *reinterpret_cast<const float4*>(gu + base + p);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    rn = fmaf(o[v], u[v], rn);
                    g[v] = u[v] * sn;
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float tt = sr * (c0 * g0[v] + c1 * g1[v] + c2 * g2[v]);
                g[v] = gu ? g[v] + tt : tt;
                q0 = fmaf(o[v], g0[v], q0);
                q1 = fmaf(o[v], g1[v], q1);
                q2 = fmaf(o[v], g2[v], q2);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                gp[v] = g[v] * (o[v] > 0.f ? 1.f : slope) * gain;
                gm = max(gm, __float_as_uint(fabsf(gp[v])));
                const float nzv = nzb ? nz[v] : 0.f;
                s0 += gp[v];
                s1 = fmaf(gp[v], nzv, s1);
                if (want_y) {
                    const float pre = o[v] * (o[v] > 0.f ? inv_pos : inv_neg);
                    s2 = fmaf(gp[v], pre - nwv * nzv - bv, s2);
                }
            }
            *reinterpret_cast<float4*>(g_pre + base + p) = *reinterpret_cast<const float4*>(gp);
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
        rn = wave_sum(rn); q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm = max(gm, (unsigned)__shfl_xor((int)gm, o, 64));
        if (lane == 0) {
            atomicAdd(&sums[pl * 3 + 0], s0);
            atomicAdd(&sums[pl * 3 + 1], s1);
            if (want_y) atomicAdd(&sums[pl * 3 + 2], s2);
            if (gu) atomicAdd(&r_next[pl], rn);
            float* rb = r_rgb + (int64_t)b * 3 * C;
            atomicAdd(&rb[c], q0);
            atomicAdd(&rb[C + c], q1);
            atomicAdd(&rb[2 * C + c], q2);
            if (g_absmax && gm != 0u) atomicMax(g_absmax + pl, gm);
        }
    }
}

// Adjoint of blur_bias_act's FIR: g [planes, 2H, 2W] -> gT parity planes [planes, 4, H+1, W+1]
//   gT[i,j] = sum_{oy,ox} g[oy,ox] * K[3-(i+1-oy)][3-(j+1-ox)]   (0 <= i+1-oy, j+1-ox <= 3)
// and, when the forward planes t are given, asum[plane] += sum gT * t  (= d * dL/dd).
// Same shape as the forward blur kernel: a thread owns a vertical strip of ADJ_QV super-pixels (a, b) -- the four parity
// outputs gT[2a+py, 2b+px] read g rows 2a-2..2a+2 x cols 2b-2..2b+2 -- and slides that 5x5 window down the strip: two
// new g rows (float2, float2, float: coalesced 8-byte lanes) per super-pixel, four coalesced plane stores.  (The first
// version gathered 16 branchy taps per output element: 8 ms of a 54 ms trainer step.)
constexpr int ADJ_QV = 4;

__global__ __launch_bounds__(256) void blur_adjoint_kernel(const float* __restrict__ g, const float* __restrict__ fir,
                                                          const float* __restrict__ t, float* __restrict__ gt,
                                                          float* __restrict__ asum, int64_t planes, int H, int W) {
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = fir[i];
    const int GH = H + 1, GW = W + 1, OH = 2 * H, OW = 2 * W;
    const int per_plane = 4 * GH * GW;
    const int HS = (GH + ADJ_QV - 1) / ADJ_QV;                  // strips per column
    const int64_t strips = planes * HS * GW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < strips; idx += (int64_t)gridDim.x * blockDim.x) {
        const int bb = (int)(idx % GW);
        int64_t r = idx / GW;
        const int as = (int)(r % HS) * ADJ_QV;
        const int64_t pl = r / HS;
        const float* gp = g + pl * (int64_t)OH * OW;
        float* gtp = gt + pl * per_plane;
        const float* tp = t ? t + pl * per_plane : nullptr;
        // window column v <-> g column 2*bb - 2 + v ; row u <-> g row 2*a - 2 + u
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
        float acc_a = 0.f;
#pragma unroll
        for (int qv = 0; qv < ADJ_QV; ++qv) {
            const int a = as + qv;
            if (a >= GH) break;
            load_row(2 * a + 1, win[3]);
            load_row(2 * a + 2, win[4]);
            // gT[2a+py, 2b+px] = sum_{dy,dx} g[2a+py+1-dy, 2b+px+1-dx] * K[3-dy][3-dx]:  window row py+3-dy, col px+3-dx
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    float v = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx) v = fmaf(win[py + 3 - dy][px + 3 - dx], k[(3 - dy) * 4 + (3 - dx)], v);
                    // rows / cols 2H+1, 2W+1 of the planes are padding (stored zeros in the forward planes)
                    if (2 * a + py > 2 * H || 2 * bb + px > 2 * W) v = 0.f;
                    const int e = (py * 2 + px) * GH * GW + a * GW + bb;
                    gtp[e] = v;
                    if (tp) acc_a = fmaf(v, tp[e], acc_a);
                }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 5; ++v) win[u][v] = win[u + 2][v];
        }
        if (t) {      // the lanes of a wave almost always sit in one plane: one atomic per wave then
            // wave_sum needs ALL 64 lanes (its v_readlane ignores EXEC): the tail wave of the grid-stride loop, whose upper
            // lanes have left the loop, takes the per-lane atomics like a wave that straddles two planes
            const int pl_lo = (int)pl;
            const bool full = __builtin_popcountll(__ballot(1)) == 64;
            if (full && __all(pl_lo == __builtin_amdgcn_readfirstlane(pl_lo))) {
                const float sum = wave_sum(acc_a);
                if ((threadIdx.x & 63) == 0) atomicAdd(&asum[pl], sum);
            } else {
                atomicAdd(&asum[pl], acc_a);
            }
        }
    }
}

// dx[b,c,q] = gu[b,c,q] * s[b,c]  (may alias gu) ;  r[b,c] += sum_q x[b,c,q] * gu[b,c,q]
__global__ __launch_bounds__(256) void scale_reduce_kernel(const float* __restrict__ gu, const float* __restrict__ x,
                                                          int64_t x_bstride, const float* __restrict__ s,
                                                          float* __restrict__ dx, float* __restrict__ r, int B, int C,
                                                          int HW, int chunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * C * chunks) return;
    const int ch = (int)(wid % chunks);
    const int64_t pl = wid / chunks;
    const int c = (int)(pl % C), b = (int)(pl / C);
    const float sv = s[pl];
    const float* xp = x + (int64_t)b * x_bstride + (int64_t)c * HW;
    const int lo = ch * kChunk, hi = min(HW, lo + kChunk);
    float acc = 0.f;
    for (int p = lo + lane; p < hi; p += 64) {
        const float g = gu[pl * HW + p];
        acc = fmaf(xp[p], g, acc);
        dx[pl * HW + p] = g * sv;
    }
    acc = wave_sum(acc);
    if (lane == 0) atomicAdd(&r[pl], acc);
}

// ToRGB backward.  One wave per (b, channel i, pixel chunk):
//   t[p] = scale * sum_j w[j,i] g[b,j,p] ;  dx[b,i,p] = s[b,i] * t[p] ;  r[b,j,i] += sum_p x[b,i,p] * g[b,j,p]
__global__ __launch_bounds__(256) void torgb_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ w_rgb, const float* __restrict__ s,
                                                       float* __restrict__ dx, float* __restrict__ r, int B, int Cin,
                                                       int HW, int chunks, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= (int64_t)B * Cin * chunks) return;
    const int ch = (int)(wid % chunks);
    const int64_t pl = wid / chunks;
    const int i = (int)(pl % Cin), b = (int)(pl / Cin);
    const float c0 = w_rgb[i] * scale, c1 = w_rgb[Cin + i] * scale, c2 = w_rgb[2 * Cin + i] * scale;
    const float sv = s[pl];
    const float* gb = g + (int64_t)b * 3 * HW;
    const int lo = ch * kChunk, hi = min(HW, lo + kChunk);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    for (int p = lo + lane; p < hi; p += 64) {
        const float g0 = gb[p], g1 = gb[HW + p], g2 = gb[2 * HW + p];
        const float xv = x[pl * HW + p];
        dx[pl * HW + p] = sv * (c0 * g0 + c1 * g1 + c2 * g2);
        r0 = fmaf(xv, g0, r0);
        r1 = fmaf(xv, g1, r1);
        r2 = fmaf(xv, g2, r2);
    }
    r0 = wave_sum(r0); r1 = wave_sum(r1); r2 = wave_sum(r2);
    if (lane == 0) {
        float* rb = r + (int64_t)b * 3 * Cin;
        atomicAdd(&rb[i], r0);
        atomicAdd(&rb[Cin + i], r1);
        atomicAdd(&rb[2 * Cin + i], r2);
    }
}

// w [Cout,Cin,KK] -> wt [Cout, KK, Cin] * scale, taps optionally reversed (flip = 1: the 180-degree rotated
// kernel that turns the forward correlation into its adjoint)
__global__ __launch_bounds__(256) void prepack_t_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                       int Cin, int KK, int flip, float scale) {
    const int64_t n = (int64_t)Cout * Cin * KK;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % Cin);
        const int64_t r = idx / Cin;
        const int k = (int)(r % KK);
        const int o = (int)(r / KK);
        const int ks = flip ? KK - 1 - k : k;
        wt[idx] = w[((int64_t)o * Cin + i) * KK + ks] * scale;
    }
}

}  // namespace sgdfr

using namespace sgdfr;

static int wave_grid(int64_t waves) { return (int)((waves + 3) / 4); }

extern "C" int sgdfr_act_grad_reduce_f32(const float* g_out, const float* out, const float* noise, int64_t noise_bstride,
                                         const float* noise_w, const float* bias, float* g_pre, float* sums, int B, int C,
                                         int HW, float slope, float gain, int want_y, unsigned int* g_absmax, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && HW > 0, "act_grad_reduce: bad shape %d %d %d", B, C, HW);
    if (B == 0) return 0;
    SGDFR_REQUIRE(g_out && out && g_pre && sums, "act_grad_reduce: null pointer");
    SGDFR_REQUIRE(!noise || noise_w, "act_grad_reduce: noise without noise_w");
    hipStream_t st = as_stream(stream);
    // (g_absmax right behind sums: one memset clears both)
    const bool adjacent = g_absmax && reinterpret_cast<float*>(g_absmax) == sums + 3 * (size_t)B * C;
    if (hipMemsetAsync(sums, 0, sizeof(float) * (size_t)B * C * (adjacent ? 4 : 3), st) != hipSuccess) return check_launch("memset");
    if (g_absmax && !adjacent && hipMemsetAsync(g_absmax, 0, sizeof(unsigned) * (size_t)B * C, st) != hipSuccess) return check_launch("memset");
    const int chunks = (HW + kChunk - 1) / kChunk;
    const int64_t waves = (int64_t)B * C * chunks;
    hipLaunchKernelGGL(act_grad_reduce_kernel, dim3(wave_grid(waves)), dim3(256), 0, st, g_out, out, noise, noise_bstride,
                       noise_w, bias, g_pre, sums, B, C, HW, chunks, slope, gain, want_y, g_absmax);
    return check_launch("act_grad_reduce");
}

extern "C" int sgdfr_grad_join_f32(const float* out, const float* gu, const float* s_next, const float* g_rgb, const float* w_rgb,
                                  const float* s_rgb, const float* g_add, const float* noise, int64_t noise_bstride,
                                  const float* noise_w, const float* bias, float* g_pre, float* sums, unsigned int* g_absmax,
                                  float* r_next, float* r_rgb, int B, int C, int HW, float slope, float gain, int want_y,
                                  int zero_outputs, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && HW > 0, "grad_join: bad shape %d %d %d", B, C, HW);
    if (B == 0) return 0;
    SGDFR_REQUIRE(out && g_pre && sums, "grad_join: null pointer");
    SGDFR_REQUIRE(gu || g_rgb || g_add, "grad_join: no incoming gradient term");
    SGDFR_REQUIRE(!gu || (s_next && r_next), "grad_join: gu needs s_next and r_next");
    SGDFR_REQUIRE(!g_rgb || (w_rgb && s_rgb && r_rgb), "grad_join: g_rgb needs w_rgb, s_rgb and r_rgb");
    SGDFR_REQUIRE(!noise || noise_w, "grad_join: noise without noise_w");
    hipStream_t st = as_stream(stream);
    const size_t planes = (size_t)B * C;
    if (zero_outputs) {     // (a caller that carves all reduction buffers of a backward out of ONE zeroed workspace passes 0)
        if (hipMemsetAsync(sums, 0, sizeof(float) * planes * 3, st) != hipSuccess) return check_launch("memset");
        if (g_absmax && hipMemsetAsync(g_absmax, 0, sizeof(unsigned) * planes, st) != hipSuccess) return check_launch("memset");
        if (gu && hipMemsetAsync(r_next, 0, sizeof(float) * planes, st) != hipSuccess) return check_launch("memset");
        if (g_rgb && hipMemsetAsync(r_rgb, 0, sizeof(float) * planes * 3, st) != hipSuccess) return check_launch("memset");
    }
    const int chunks = (HW + kChunk - 1) / kChunk;
    const int64_t waves = (int64_t)B * C * chunks;
    const float rgb_scale = 1.0f / sqrtf((float)C);
    auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = HW % 4 == 0 && noise_bstride % 4 == 0 && aligned(out) && aligned(gu) && aligned(g_rgb) && aligned(g_add) &&
                     aligned(noise) && aligned(g_pre);
    static const int rgb_reuse = getenv("SGDFR_GRAD_JOIN_RGB") ? atoi(getenv("SGDFR_GRAD_JOIN_RGB")) : 1;     // (0: A/B against one wave per channel)
    if (vec && g_rgb && !g_add && rgb_reuse && HW >= 1024) {
        const int64_t w4 = (int64_t)B * ((C + GJ_CPW - 1) / GJ_CPW) * chunks;
        hipLaunchKernelGGL(grad_join_rgb_kernel, dim3(wave_grid(w4)), dim3(256), 0, st, out, gu, s_next, g_rgb, w_rgb, s_rgb, rgb_scale,
                           noise, noise_bstride, noise_w, bias, g_pre, sums, g_absmax, r_next, r_rgb, B, C, HW, chunks, slope, gain, want_y);
        return check_launch("grad_join(rgb)");
    }
    if (vec)
        hipLaunchKernelGGL(grad_join_kernel<4>, dim3(wave_grid(waves)), dim3(256), 0, st, out, gu, s_next, g_rgb, w_rgb, s_rgb, rgb_scale,
                           g_add, noise, noise_bstride, noise_w, bias, g_pre, sums, g_absmax, r_next, r_rgb, B, C, HW, chunks, slope,
                           gain, want_y);
    else
        hipLaunchKernelGGL(grad_join_kernel<1>, dim3(wave_grid(waves)), dim3(256), 0, st, out, gu, s_next, g_rgb, w_rgb, s_rgb, rgb_scale,
                           g_add, noise, noise_bstride, noise_w, bias, g_pre, sums, g_absmax, r_next, r_rgb, B, C, HW, chunks, slope,
                           gain, want_y);
    return check_launch("grad_join");
}

extern "C" int sgdfr_blur_adjoint_f32(const float* g, const float* fir, const float* t, float* gt, float* asum, int B,
                                      int C, int H, int W, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, "blur_adjoint: bad shape %d %d %d %d", B, C, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(g && fir && gt, "blur_adjoint: null pointer");
    SGDFR_REQUIRE(!t || asum, "blur_adjoint: t given without asum");
    hipStream_t st = as_stream(stream);
    const int64_t planes = (int64_t)B * C;
    if (t && hipMemsetAsync(asum, 0, sizeof(float) * (size_t)planes, st) != hipSuccess) return check_launch("memset");
    const int64_t strips = planes * ((H + 1 + ADJ_QV - 1) / ADJ_QV) * (W + 1);
    int64_t grid = (strips + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(blur_adjoint_kernel, dim3((int)grid), dim3(256), 0, st, g, fir, t, gt, asum, planes, H, W);
    return check_launch("blur_adjoint");
}

extern "C" int sgdfr_scale_reduce_f32(const float* gu, const float* x, int64_t x_bstride, const float* s, float* dx,
                                      float* r, int B, int C, int HW, void* stream) {
    SGDFR_REQUIRE(B >= 0 && C > 0 && HW > 0, "scale_reduce: bad shape %d %d %d", B, C, HW);
    if (B == 0) return 0;
    SGDFR_REQUIRE(gu && x && s && dx && r, "scale_reduce: null pointer");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(r, 0, sizeof(float) * (size_t)B * C, st) != hipSuccess) return check_launch("memset");
    const int chunks = (HW + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(scale_reduce_kernel, dim3(wave_grid((int64_t)B * C * chunks)), dim3(256), 0, st, gu, x, x_bstride,
                       s, dx, r, B, C, HW, chunks);
    return check_launch("scale_reduce");
}

extern "C" int sgdfr_torgb_bwd_f32(const float* x, const float* g, const float* w_rgb, const float* s, float* dx, float* r,
                                   int B, int Cin, int H, int W, void* stream) {
    SGDFR_REQUIRE(B >= 0 && Cin > 0 && H > 0 && W > 0, "torgb_bwd: bad shape %d %d %d %d", B, Cin, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && g && w_rgb && s && dx && r, "torgb_bwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(r, 0, sizeof(float) * 3 * (size_t)B * Cin, st) != hipSuccess) return check_launch("memset");
    const int HW = H * W;
    const int chunks = (HW + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(torgb_bwd_kernel, dim3(wave_grid((int64_t)B * Cin * chunks)), dim3(256), 0, st, x, g, w_rgb, s, dx,
                       r, B, Cin, HW, chunks, 1.0f / sqrtf((float)Cin));
    return check_launch("torgb_bwd");
}

extern "C" int sgdfr_modconv_prepack_t_f32(const float* weight, float* wt, int Cout, int Cin, int k, int flip,
                                           void* stream) {
    SGDFR_REQUIRE(Cout > 0 && Cin > 0 && k > 0, "prepack_t: bad shape %d %d %d", Cout, Cin, k);
    SGDFR_REQUIRE(weight && wt, "prepack_t: null pointer");
    const int64_t n = (int64_t)Cout * Cin * k * k;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(prepack_t_kernel, dim3((int)g), dim3(256), 0, as_stream(stream), weight, wt, Cout, Cin, k * k, flip,
                       1.0f / sqrtf((float)Cin * k * k));
    return check_launch("modconv_prepack_t");
}
