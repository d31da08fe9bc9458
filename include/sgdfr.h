/*
 * sgdfr.h -- C ABI of libsgdfr_hip.so: the MI355X (gfx950) StyleGAN2 generator hot path.
 *
 * Every entry point takes raw device pointers + sizes + a hipStream_t (passed as void*),
 * launches asynchronously on that stream, never synchronises, never allocates, and returns
 * 0 on success or a non-zero code with a message available from sgdfr_last_error().
 * Tensors are fp32, contiguous, NCHW unless a stride argument says otherwise.
 * Outputs are caller-allocated (the reference's natives allocate internally; the host mirror
 * in stylegan_directions_face_reenactment_amd/ does the torch.empty for them).
 *
 * Reference interface each symbol replaces (paths relative to the reference repo,
 * libs/gan/StyleGAN2/ unless noted):
 *
 *   sgdfr_fused_bias_act_f32   op/fused_bias_act.cpp:14-24  fused.fused_bias_act(input,bias,refer,act,grad,alpha,scale)
 *                              (kernel op/fused_bias_act_kernel.cu:18-49)
 *   sgdfr_upfirdn2d_f32        op/upfirdn2d.cpp:15-26       upfirdn2d.upfirdn2d(input,kernel,up_x,up_y,down_x,down_y,pad_x0..pad_y1)
 *                              (kernel op/upfirdn2d_kernel.cu:52-137; unlike :172-223 unsupported combos are an ERROR, never a silent no-op)
 *   sgdfr_linear_f32           model.py:148-157 EqualLinear.forward (F.linear + fused_leaky_relu),
 *                              libs/models/direction_matrix.py:41-48 (nn.Linear)
 *   sgdfr_pixelnorm_f32        model.py:11-16   PixelNorm.forward
 *   sgdfr_latent_prepare_f32   model.py:494-508 truncation + W->W+ broadcast, libs/utilities/generic.py:116-135 shift add
 *   sgdfr_modconv_prepack_f32  model.py:215,218-220,236 (scale*weight) -- layout change + sum-of-squares for demodulation
 *   sgdfr_style_demod_f32      model.py:235-240 modulation(style) and rsqrt(sum w^2 + 1e-8)
 *   sgdfr_styles_batched_f32   the same for all 20 modulated convs of Generator.forward (model.py:520-531) at once
 *   sgdfr_modconv2d_fwd_f32    model.py:232-273 ModulatedConv2d.forward (+ :282-287 NoiseInjection, op/fused_act.py:81-86
 *                              FusedLeakyReLU folded into the epilogue); the reference has NO native kernel for this --
 *                              its boundary is the Python method
 *   sgdfr_blur_bias_act_f32    model.py:257 Blur after the transposed conv (upfirdn2d pad (1,1)) + noise + bias + lrelu
 *   sgdfr_torgb_fwd_f32        model.py:350-359 ToRGB.forward (1x1 modconv, bias, Upsample(skip) :38-46, add)
 *   sgdfr_make_shift_f32       run_inference.py:201-254 Inference.make_shift, libs/utilities/utils_train.py:127-175 make_shift_vector
 *   sgdfr_make_shift_random_f32 libs/utilities/utils_train.py:227-286 (single-direction half of make_shift_vector_50)
 */
#ifndef SGDFR_H
#define SGDFR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGDFR_ABI_VERSION 22

/* modes of sgdfr_modconv2d_fwd_f32 */
#define SGDFR_MODE_PLAIN3 0 /* 3x3, pad 1, same resolution                      (model.py:267-271) */
#define SGDFR_MODE_UP3 1    /* 3x3 transposed stride 2 -> phase planes of size (H+1)x(W+1) (model.py:246-256) */
#define SGDFR_MODE_DOWN3 2  /* adjoint of UP3: 3x3 stride-2 conv reading phase planes [B,Cin,4,H+1,W+1] -> [B,Cout,H,W]
                               (the dX of the transposed conv; autograd of model.py:254) */

/* activation codes of sgdfr_linear_f32 */
#define SGDFR_ACT_NONE 0
#define SGDFR_ACT_LRELU 1 /* lrelu(v, slope) * gain */

int sgdfr_abi_version(void);
const char* sgdfr_last_error(void);

/* y[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale ; act==3 (lrelu) only, grad in {0,1,2}:
 *   grad 0: y = (v>0 ? v : alpha*v) * scale, v = x + b
 *   grad 1: y = (ref>0 ? x : alpha*x) * scale           (bias ignored by callers: pass NULL)
 *   grad 2: y = 0
 * bias / ref may be NULL ("empty tensor" in the reference). */
int sgdfr_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* y, int64_t n,
                             int step_b, int size_b, int act, int grad, float alpha, float scale,
                             void* stream);

/* x [major, in_h, in_w, minor] -> y [major, out_h, out_w, minor],
 * out = ((in*up + pad0 + pad1 - k) / down) + 1 per axis; kernel k [kh, kw] is applied flipped
 * (true convolution).  Negative pads crop. */
int sgdfr_upfirdn2d_f32(const float* x, const float* k, float* y, int major, int in_h, int in_w, int minor,
                        int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                        int pad_y0, int pad_y1, void* stream);

/* The two natives for every dtype the reference dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF: fused_bias_act_kernel.cu:79,
 * upfirdn2d_kernel.cu:225): x / bias / ref / k / y all of `dtype`; half is computed in float and rounded once, double in double.
 * SGDFR_DTYPE_F32 forwards to the _f32 entry points above. */
#define SGDFR_DTYPE_F32 0
#define SGDFR_DTYPE_F16 1
#define SGDFR_DTYPE_F64 2
int sgdfr_fused_bias_act(const void* x, const void* bias, const void* ref, void* y, int64_t n, int step_b, int size_b, int act,
                         int grad, float alpha, float scale, int dtype, void* stream);
int sgdfr_upfirdn2d(const void* x, const void* k, void* y, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x,
                    int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype, void* stream);

/* y[m, n] = act( (sum_k x[m*ldx + k] * w[n*K + k]) * wscale + bias[n] * bscale ), m<M, n<N.
 * bias may be NULL.  act: SGDFR_ACT_*. */
int sgdfr_linear_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                     int M, int N, int K, float wscale, float bscale, int act, float slope, float gain,
                     void* stream);

/* y[b,:] = x[b,:] * rsqrt(mean(x[b,:]^2) + eps) */
int sgdfr_pixelnorm_f32(const float* x, float* y, int B, int D, float eps, void* stream);
/* its adjoint: dx[b,:] = r*g - x * r^3 * mean(g*x), r = rsqrt(mean(x^2) + eps)   (autograd of model.py:11-16) */
int sgdfr_pixelnorm_bwd_f32(const float* x, const float* g, float* dx, int B, int D, float eps, void* stream);

/* out[b,l,:] = t + psi * (v - t),   v = w[b,(l),:] + (l < shift_layers ? shift[b,(l),:] : 0)
 *   w:     [B, L, D] if w_is_plus else [B, D] (broadcast over l)
 *   shift: NULL, or [B, shift_layers, D] if shift_is_plus else [B, D] (added to the first shift_layers rows)
 *   trunc: NULL (no truncation, psi ignored) or [D]
 * Order matches generic.py:116-135 then model.py:494-500: shift first, truncation after. */
int sgdfr_latent_prepare_f32(const float* w, int w_is_plus, const float* shift, int shift_is_plus,
                             int shift_layers, const float* trunc, float psi, float* out, int B, int L, int D,
                             void* stream);

/* weight [Cout, Cin, k, k] -> wp [Cin, k*k, Cout] = weight * 1/sqrt(Cin*k*k)  and
 *                              q  [Cout, Cin]     = sum_taps wp^2 ,  qt [Cin, Cout] = q^T   (q, qt may be NULL) */
int sgdfr_modconv_prepack_f32(const float* weight, float* wp, float* q, float* qt, int Cout, int Cin, int k,
                              void* stream);

/* weight [Cout, Cin, k, k] -> wt [Cout, k*k, Cin] * 1/sqrt(Cin*k*k), taps reversed when flip != 0: the weight pack of
 * the ADJOINT convs (dL/dx): sgdfr_modconv2d_fwd_f32 with (Cin, Cout) exchanged, s <- d, d <- s;
 * PLAIN3 backward uses flip = 1 in mode PLAIN3, UP3 backward uses flip = 0 in mode DOWN3. */
int sgdfr_modconv_prepack_t_f32(const float* weight, float* wt, int Cout, int Cin, int k, int flip, void* stream);

/* s[b,i] = (sum_j style[b*ld_style + j] * mod_w[i*D + j]) / sqrt(D) + mod_b[i]
 * d[b,o] = rsqrt( sum_i s[b,i]^2 * q[o*Cin + i] + 1e-8 )        (skipped when d == NULL) */
int sgdfr_style_demod_f32(const float* style, int64_t ld_style, const float* mod_w, const float* mod_b,
                          const float* q, float* s, float* d, int B, int D, int Cin, int Cout, void* stream);

/* All style modulations and demodulation coefficients of one generator forward in two launches
 * (sgdfr_style_demod_f32 for every layer at once).  `layers` is a HOST array; layer i reads
 * latent[:, latent_index, :] of latent [B, L, D] and writes s [B,cin] and, when d != NULL, d [B,cout]. */
#define SGDFR_MAX_STYLE_LAYERS 40
typedef struct sgdfr_style_layer {
    const float* mod_w; /* [cin, D] */
    const float* mod_b; /* [cin]    */
    const float* q;     /* [cout, cin] or NULL */
    float* s;           /* [B, cin]  */
    float* d;           /* [B, cout] or NULL */
    int cin, cout, latent_index;
    /* Range plan of the fp16-split conv (csrc/split.hip), optional -- needs d.  A demodulated conv is invariant to a
     * per-image scale of its style row: y = d * conv(x*s, W) = (d * 2^-e) * conv(x * (s * 2^e), W).  With s_n / d_n given,
     * the launch also writes s_n[b,:] = s[b,:] * 2^e_b and d_n[b,:] = d[b,:] * 2^-e_b (exact powers of two) with e_b chosen
     * so that the conv's fp16 operand x*s_n*2^-4 tops out `headroom` binades below the fp16 maximum:
     *     e_b = 18 - headroom - L - floor(log2 max_i |s[b,i]|),   |x| < 2^L
     * L = x_log2, or floor(log2 *x_absmax)+1 when x_absmax (one fp32 bit pattern = max |x| over the input, written by
     * sgdfr_absmax_f32) is given.  The convs then see the same product range whatever the scale of the styles. */
    float* s_n;               /* [B, cin] or NULL */
    float* d_n;               /* [B, cout] or NULL */
    const unsigned* x_absmax; /* device, 1 word, or NULL */
    int x_log2, headroom;
} sgdfr_style_layer;
int sgdfr_styles_batched_f32(const float* latent, int B, int L, int D, const sgdfr_style_layer* layers, int n_layers,
                             void* stream);

/* The range plan above for ONE layer whose s / d already exist (autograd forward, stand-alone layer calls):
 * s_n = s * 2^e_b, d_n = d * 2^-e_b.  x_absmax: device words of fp32 bit patterns, one per image (x_absmax_bstride 1), n per
 * image whose maximum counts (x_absmax_bstride n > 1: the per-plane words of sgdfr_act_grad_reduce_f32), one for the whole batch
 * (0), or NULL (then |x| < 2^x_log2 is taken on trust). */
int sgdfr_split_range_f32(const float* s, const float* d, float* s_n, float* d_n, const unsigned* x_absmax,
                          int x_absmax_bstride, int x_log2, int headroom, int B, int Cin, int Cout, void* stream);

/* out[b] (or out[0] when per_image == 0) = bit pattern of max |x| over image b (over everything): non-negative floats order
 * like their bit patterns, so this is an atomicMax on words the call zeroes first.  A NaN or Inf anywhere gives a word
 * >= 0x7f800000.  x_bstride 0 = one image shared by the batch: only out[0] is written. */
int sgdfr_absmax_f32(const float* x, int64_t x_bstride, int64_t n_per_image, int B, unsigned* out, int per_image,
                     void* stream);

/* Shared-weight modulated 3x3 convolution on fp32 MFMA.
 *   x      [B, Cin, H, W]  (x_bstride = Cin*H*W, or 0 to broadcast one [Cin,H,W] constant over the batch)
 *   wp     [Cin, 9, Cout]  from sgdfr_modconv_prepack_f32
 *   s      [B, Cin], d [B, Cout] (d may be NULL = no demodulation)
 * mode PLAIN3: y [B, Cout, H, W] = act( d * conv3x3(x*s) + noise_w[0]*noise + bias[o] ), act = lrelu(slope)*gain
 *              when `act` != 0; noise [H*W] with noise_bstride 0 (shared) or H*W (per sample); noise/noise_w/bias
 *              may be NULL.
 * mode UP3:    y [B, Cout, 4, H+1, W+1] = d * conv_transpose2d(x*s, stride 2) split by output parity:
 *              plane ph = 2*(oy&1) + (ox&1) holds T[oy, ox] at [oy>>1, ox>>1]; entries of the odd planes that fall
 *              outside the (2H+1)x(2W+1) result are written as exact zeros.  noise/bias/act are NOT applied
 *              (sgdfr_blur_bias_act_f32 does that after the FIR).
 * mode DOWN3:  x is [B, Cin, 4, H+1, W+1] parity planes (x_bstride = Cin*4*(H+1)*(W+1)), y [B, Cout, H, W]:
 *              y[b,n,a,c] = d[b,n] * sum_{i,ky,kx} s[b,i] * T_i[2a+ky, 2c+kx] * wp[i][ky*3+kx][n]  (+ bias, act like PLAIN3;
 *              noise must be NULL).  With wp = transposed pack it is dL/dx of mode UP3. */
int sgdfr_modconv2d_fwd_f32(const float* x, int64_t x_bstride, const float* wp, const float* s, const float* d,
                            const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                            float* y, int B, int Cin, int Cout, int H, int W, int mode, int act, float slope,
                            float gain, void* stream);

/* K-sliced form of sgdfr_modconv2d_fwd_f32 (modes PLAIN3 / UP3) for launches too small to fill the chip (single-frame
 * reenactment, 4x4 / 8x8 layers): `splits` replicas of the tile grid each reduce a slice of the input channels into
 * partials [splits][elements of y]; a second launch adds them in a fixed order (deterministic) and applies the epilogue.
 * sgdfr_modconv2d_splitk_hint returns the recommended number of slices (1 = use the plain entry point). */
int sgdfr_modconv2d_splitk_hint(int B, int Cin, int Cout, int H, int W, int mode);
int sgdfr_modconv2d_splitk_f32(const float* x, int64_t x_bstride, const float* wp, const float* s, const float* d,
                               const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias, float* y,
                               float* partials, int splits, int B, int Cin, int Cout, int H, int W, int mode, int act,
                               float slope, float gain, void* stream);

/* Winograd F(2x2,3x3) form of mode PLAIN3 (same result up to fp32 rounding, 2.25x fewer MFMA ops).
 *   sgdfr_modconv_prepack_wino_f32: weight [Cout,Cin,3,3] -> u [Cin][Cout][16] = (G g G^T) / sqrt(9*Cin), the four 16-byte
 *       quads of each 16-vector stored at slot (quad ^ (cout & 3)) (LDS bank swizzle, opaque to callers);
 *       transpose_flip != 0 packs the ADJOINT conv instead: u [Cout][Cin][16] of the 180-degree rotated, transposed kernel
 *   sgdfr_modconv2d_wino_supported: 1 when the shape can use it (Cin % 8 == 0, Cout % 64 == 0, H and W even)
 *   sgdfr_modconv2d_wino_f32: arguments as sgdfr_modconv2d_fwd_f32(mode PLAIN3) with u in place of wp, plus `zeros`:
 *       a device buffer of >= 16 zero bytes (the global->LDS DMA reads padding positions from it) */
int sgdfr_modconv_prepack_wino_f32(const float* weight, float* u, int Cout, int Cin, int transpose_flip, void* stream);
int sgdfr_modconv2d_wino_supported(int B, int Cin, int Cout, int H, int W);
int sgdfr_modconv2d_wino_f32(const float* x, int64_t x_bstride, const float* u, const float* s, const float* d,
                             const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                             const float* zeros, float* y, int B, int Cin, int Cout, int H, int W, int act, float slope,
                             float gain, void* stream);

/* t [B*C, 4, H+1, W+1] (phase planes from MODE_UP3) -> y [B, C, 2H, 2W]:
 *   y = act( upfirdn2d(T, fir[4,4], pad=(1,1)) + noise_w[0]*noise[oy,ox] + bias[c] )   (model.py:257,287, fused_act.py:81)
 * y_absmax (optional, [B*C] words ZEROED by the caller): fp32 bit pattern of max |y| per (image, channel) plane from the same pass --
 * the range plan of the fp16-split conv that reads y (sgdfr_split_range_f32 with x_absmax_bstride = C), instead of a separate
 * sgdfr_absmax_f32 pass over y (autograd forward). */
int sgdfr_blur_bias_act_f32(const float* t, const float* fir, const float* noise, int64_t noise_bstride,
                            const float* noise_w, const float* bias, float* y, int B, int C, int H, int W, int act,
                            float slope, float gain, unsigned int* y_absmax, void* stream);

/* sgdfr_blur_bias_act_f32 with the result multiplied by the NEXT layer's modulation s_next [B,C] and written in that layer's
 * split input form xs [B][C/8][2][2H*2W][8] (see sgdfr_to_split_f32) instead of fp32 NCHW.  plane_stride: 0 = t is the dense
 * planar [B*C][4][(H+1)*(W+1)]; otherwise t is the interleaved [B*C][plane_stride][px][py] form of sgdfr_modconv2d_split_f32
 * (16-byte aligned).  wino = 2 | 4 (W = 8 ... 64; wino = 4 on interleaved planes also W = 128): xs receives the
 * Winograd input form [B][C/8][wino+2][2][2H*2W/wino][8] of sgdfr_to_wsplit_f32(f = wino) instead (2x / 1.5x the bytes), for
 * sgdfr_modconv2d_wsplit_f32. */
int sgdfr_blur_bias_act_split_f32(const float* t, const float* fir, const float* noise, int64_t noise_bstride,
                                  const float* noise_w, const float* bias, const float* s_next, unsigned short* xs, int B, int C,
                                  int H, int W, int64_t plane_stride, int arith, int wino, int act, float slope, float gain,
                                  unsigned int* sat, void* stream);

/* y[b,j,p] = sum_i w_rgb[j*Cin+i]/sqrt(Cin) * s[b,i] * x[b,i,p] + bias[j]
 *          + (skip ? upfirdn2d(skip[b,j] (H/2 x W/2), fir[4,4], up=2, pad=(2,1))[p] : 0),  j<3 */
int sgdfr_torgb_fwd_f32(const float* x, const float* w_rgb, const float* s, const float* bias, const float* skip,
                        const float* fir, float* y, int B, int Cin, int H, int W, void* stream);

/* Split-operand arithmetic of the 3x3 modulated convs (same contract as sgdfr_modconv2d_fwd_f32 modes PLAIN3 / UP3,
 * ModulatedConv2d.forward model.py:232-273): fp32 operands are split into two 16-bit terms hi + lo and contracted as
 * hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation.  `arith` selects the terms:
 *   SGDFR_SPLIT_FP16  fp16 hi+lo = 22 operand bits (fp32-grade; 7.7e-6 max-abs vs an fp64 evaluation of the 256x256 generator).
 *                     THE DEFAULT of the Python host layer (functional.PRECISION = 'fp16x3') for inference and the autograd
 *                     forward.  fp16 has 5 exponent bits: the host supplies styles / demodulation scaled by an exact power of
 *                     two per image (the "range plan", sgdfr_styles_batched_f32 / sgdfr_split_range_f32) so the largest operand
 *                     sits a few binades under 65504.  Operands that still leave the range, or are NaN/Inf, are CLAMPED AND
 *                     COUNTED in the caller's saturation word (below) -- never silently.
 *   SGDFR_SPLIT_BF16  bf16 hi+lo = 16 operand bits, full fp32 exponent range (~1.2e-4 on the same images; contract 1e-3).
 *                     Selected AUTOMATICALLY by the host layer as the fallback of a generator whose saturation word became
 *                     non-zero (the affected batch is re-rendered before it is returned by the verified entry points:
 *                     generate_image, ReenactmentSession, Generator.forward(verify_range=True)), or explicitly.
 * A caller pins an arithmetic with functional.set_precision('fp32' | 'fp16x3' | 'bf16x3') / SGDFR_PRECISION, or per call through
 * the `arith` argument here ('fp32' = the sgdfr_modconv2d_fwd_f32 / _wino_f32 kernels, no split entry point involved).
 *
 * Saturation words: every entry point that converts to the fp16 terms takes `unsigned int* sat`: a device word the kernel
 * atomically adds its count of clamped / non-finite operand pairs to (one word per generator, or per batch in flight: the
 * owner zeroes and reads it; nothing is shared between generators or streams).  NULL = the legacy device-wide counter read
 * by sgdfr_split_saturation_count().
 *   sgdfr_modconv_prepack_split_elems: uint16 elements of the packed weight buffer (= 2*9*Cin*Cout, Cout rounded up to the 64-wide cout tiles of the pack:
 *     a forward pack may have Cout = 32 mod 64, its last tile is half padding)
 *   sgdfr_modconv_prepack_split_f32:   weight [Cout,Cin,3,3] -> 16-bit hi/lo of weight/sqrt(9 Cin) in kernel order;
 *                                      transpose_flip = 1 packs the adjoint conv (Cin outputs, Cout inputs, taps rotated),
 *                                      2 the adjoint of the transposed conv (channels swapped, taps stored phase by phase):
 *                                      dL/dx of the plain conv is then the same PLAIN3 kernel
 *   sgdfr_modconv2d_split_supported:   1 when the shape can use it (Cin % 16 == 0, Cout % 64 == 0, tileable H x W)
 *   sgdfr_modconv2d_split_f32:         arguments as sgdfr_modconv2d_wino_f32 with wsp in place of u, plus mode:
 *                                      PLAIN3, or UP3 = stride-2 transposed conv into parity planes
 *                                      [B,Cout,4,H+1,W+1] exactly as sgdfr_modconv2d_fwd_f32(mode UP3) writes them;
 *                                      ksplit > 1 (see ..._ksplit_hint; layers too small to fill the chip) slices the
 *                                      channel blocks over extra thread blocks into `partials` [ksplit][numel(y)] and
 *                                      reduces them in fixed order (deterministic).
 *                                      rgb_part != NULL (PLAIN3, ksplit == 1) also fuses the 1x1 ToRGB conv that follows
 *                                      the layer (ToRGB.forward model.py:350-359): rgb_w [3][Cout], rgb_s [B][Cout] (its
 *                                      modulation), rgb_part [B][T*3][H*W] with T = ..._split_cout_tiles(); channel t*3+j
 *                                      holds the sum over cout tile t, so sgdfr_torgb_fwd_f32 over those T*3 channels with
 *                                      indicator weights finishes it (+ bias + upsampled skip) without re-reading x;
 *                                      with rgb_part, y may be NULL: the activation is then not stored at all (last layer of
 *                                      the generator: only its ToRGB consumes it) */
#define SGDFR_SPLIT_BF16 0   /* bf16 hi+lo: 16 mantissa bits, fp32 range   (~1e-4 on the 256x256 generator) */
#define SGDFR_SPLIT_FP16 1   /* fp16 hi+lo: 22 mantissa bits = fp32-grade; operands range-shifted by exact powers of two,
                                |x*s| saturates at 1.04e6 */
#define SGDFR_SPLIT_FP16F8 2 /* fp16 main term + fp8 (e4m3) cross terms: the "lo" chunk of eight channels holds (8 x fp8 | 8 x fp8)
                                instead of 8 x fp16, both cross terms of 32 K-elements are one v_mfma_scale_f32_32x32x64_f8f6f4:
                                2 MFMA units per product instead of 3, ~1e-5 of max|y| per layer (5e-5 in F(4,3) form).  Taken by
                                the F(4,3) wide-tile conv (sgdfr_modconv2d_wsplit_f32, f = 4), the transposed conv's deep plan and
                                the 4-wave plain plan (sgdfr_modconv2d_split_f8_ok), their packs and the producers of their inputs */
int64_t sgdfr_modconv_prepack_split_elems(int Cout, int Cin);
int sgdfr_modconv_prepack_split_f32(const float* weight, unsigned short* wsp, int Cout, int Cin, int arith, int transpose_flip,
                                    unsigned int* sat, void* stream);
int sgdfr_modconv2d_split_supported(int B, int Cin, int Cout, int H, int W, int mode);
int sgdfr_modconv2d_split_f8_ok(int B, int Cin, int Cout, int H, int W, int mode);      /* arith = SGDFR_SPLIT_FP16F8 allowed (UP3 deep plan, x_is_split = 1, forward pack of the same arith) */
int sgdfr_modconv2d_split_f32(const float* x, int64_t x_bstride, const unsigned short* wsp, const float* s, const float* d,
                              const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                              const float* zeros, float* y, float* partials, int ksplit, const float* rgb_w,
                              const float* rgb_s, float* rgb_part, int x_is_split, unsigned short* xs_out, const float* s_next,
                              int B, int Cin, int Cout, int H, int W, int mode, int64_t plane_stride, int arith, int act,
                              float slope, float gain, unsigned int* sat, void* stream);
/* plane_stride (UP3, ksplit <= 1; 0 = dense planar planes y [B][Cout][4][(H+1)*(W+1)]): positions per (image, cout) of the
 * INTERLEAVED form y [B][Cout][plane_stride][px][py] -- the four parity phases of a super-pixel are one 16-byte quad (element
 * 2*px + py), >= (H+1)*(W+1) positions, the same bytes in total.  A multiple of 32 positions keeps every store run of the kernel
 * on whole 128-byte lines (dense planes are odd-sized: their stores run at half the write bandwidth), a lane stores one quad
 * instead of four dwords, and sgdfr_blur_bias_act_split_f32 (same stride) fetches a super-pixel with one 16-byte load. */
/* x [B,Cin,H,W], s [B,Cin] -> xs [B][Cin/8][2][H*W][8] 16-bit: x*s already split (and range-shifted) the way the kernel
 * stages it; sgdfr_modconv2d_split_f32(x = xs, s = NULL, x_is_split = 1) then fills LDS by DMA only.  Producers can emit
 * that form directly: sgdfr_modconv2d_split_f32(xs_out, s_next = the NEXT layer's modulation [B,Cout]) (y may then be NULL)
 * and sgdfr_blur_bias_act_f32's split variant, so activations between layers never exist as fp32 in HBM. */
/* fp16-split operand pairs clamped to +-65504 by split launches that were given sat = NULL, on the current device since the
 * last reset (launches with their own saturation word do not count here).  Synchronises the device; < 0 on a HIP error. */
long long sgdfr_split_saturation_count(int reset);
int sgdfr_modconv2d_split_xin_supported(int B, int Cin, int Cout, int H, int W, int mode);   /* x_is_split allowed for this shape */
int sgdfr_to_split_f32(const float* x, const float* s, unsigned short* xs, int B, int Cin, int H, int W, int arith,
                       unsigned int* sat, void* stream);
/* Backward of the transposed conv on the split kernels (autograd of ModulatedConv2d.forward model.py:246-256; consumers
 * trainer.py:188, optimization.py:67): dL/d(x*s) = the stride-2 3x3 conv of the gradient's parity planes with the transposed
 * kernel = sgdfr_modconv2d_split_f32(mode = SGDFR_MODE_DOWN3, x_is_split = 1, Cin = plane channels C, Cout = channels of
 * dL/dx, H x W = size of x, weights packed with transpose_flip = 2).  Its input is written by
 *   sgdfr_planes_to_split_f32: gt [B,C,4,H+1,W+1] fp32 (gradient of the parity planes), d [B,C] or NULL (the per-plane
 *                              scale: the layer's demodulation) -> xs [B][(ph*C + c)/8][2][(H+1)*(W+1)][8] 16-bit, the
 *                              phase-major split form of gt*d (16-byte aligned, 4 bytes per element). */
int sgdfr_planes_to_split_f32(const float* gt, const float* d, unsigned short* xs, int B, int C, int H, int W, int arith,
                              unsigned int* sat, void* stream);
/* sgdfr_blur_adjoint_f32 followed by sgdfr_planes_to_split_f32 in one pass (frozen generator: nothing else reads the fp32
 * plane gradient): g [B,C,2H,2W] -> xs as above (times d [B,C] or 1), asum[b,c] = sum gT*t when the forward planes
 * t [B,C,4,H+1,W+1] are given (the demodulation gradient; asum is zeroed first). */
int sgdfr_blur_adjoint_split_f32(const float* g, const float* fir, const float* t, const float* d, unsigned short* xs,
                                 float* asum, int B, int C, int H, int W, int arith, unsigned int* sat, void* stream);
int sgdfr_modconv2d_split_cout_tiles(int B, int Cin, int Cout, int H, int W, int mode);
/* the cout tiling of a launch with x_is_split = 1 and ksplit <= 1 (its tiling plan may differ: T of rgb_part [B][T*3][H*W]) */
int sgdfr_modconv2d_split_cout_tiles_xin(int B, int Cin, int Cout, int H, int W, int mode);
int sgdfr_modconv2d_split_ksplit_hint(int B, int Cin, int Cout, int H, int W, int mode);

/* ---- The plain 3x3 modulated conv in 1-D Winograd form on the split-operand matrix cores (csrc/wsplit.hip; replaces the same
 * reference lines as sgdfr_modconv2d_split_f32 mode PLAIN3: ModulatedConv2d.forward model.py:232-273 + NoiseInjection
 * model.py:284-293 + FusedLeakyReLU op/fused_act.py:79-86).  Along image rows f neighbouring outputs come from f+2 transformed
 * inputs, kernel rows stay direct: F(2,3) (f = 2: 4 multiplies instead of 6, 2/3 of the MFMA work) or F(4,3) (f = 4: 6 instead of
 * 12, 1/2; interpolation points 0, +-1, +-2, inf).  The transforms are applied to fp32 values before the hi/lo split, so the
 * products keep the arithmetic of `arith` (same SGDFR_SPLIT_* constants, same saturation rule); |V| <= 2 max|x*s| (f = 2) /
 * 10 max|x*s| (f = 4): plan 1 / 4 more binades of headroom.
 *   sgdfr_to_wsplit_f32:               x [B,Cin,H,W], s [B,Cin] -> vs "WS" [B][Cin/8][t f+2][hi,lo][H*W/f][8] 16-bit: V = B^T (x*s)
 *                                      per tile of f outputs, d_j = (x*s)[f*tile-1+j] (zero outside the row); f = 2: V0 = d0-d2,
 *                                      V1 = d1+d2, V2 = d2-d1, V3 = d1-d3.  Producers may emit it directly
 *                                      (sgdfr_blur_bias_act_split_f32 with wino = f).
 *   sgdfr_modconv_prepack_wsplit_f32:  weight [Cout,Cin,3,3] -> U = G (weight/sqrt(9 Cin)) per kernel row, 16-bit hi/lo in
 *                                      kernel order, sgdfr_modconv_prepack_wsplit_elems() uint16 elements (= 2*3*(f+2)*Cin*Cout + an 8-element trailer).
 *   sgdfr_modconv2d_wsplit_supported:  Cin % 16 == 0, Cout % 128 == 0, W/f a multiple of the patch width (f = 2: min(16, W/2)
 *                                      >= 8 tiles; f = 4: min(8, W/4) >= 4), H a multiple of the patch height (256 / (f * width)).
 *   sgdfr_modconv2d_wsplit_f32:        outputs as sgdfr_modconv2d_split_f32 (PLAIN3, ksplit = 1, x_is_split = 1): y (may be
 *                                      NULL when xs_out or rgb_part is given), the next conv's split input xs_out (+ s_next),
 *                                      ToRGB partial sums rgb_part [B][(Cout/128)*3][H*W] (+ rgb_w, rgb_s).
 * arith = SGDFR_SPLIT_FP16F8 (f = 4 only, in all three calls and in sgdfr_blur_bias_act_split_f32 with wino = 4; the direct kernels'
 * side of it -- sgdfr_to_split_f32, sgdfr_modconv_prepack_split_f32, sgdfr_modconv2d_split_f32 where sgdfr_modconv2d_split_f8_ok(),
 * sgdfr_blur_bias_act_split_f32 with wino = 0 -- uses the same chunk format): the fp16 main
 * term plus BOTH cross terms in fp8 -- the 16-byte lo chunk of eight channels holds two 8-byte halves (channels 0-3, 4-7) of
 * (4 x e4m3 lo * 2^7 | 4 x e4m3 hi * 2^-4) for activations and (4 x e4m3 hi * 2^-EW | 4 x e4m3 lo * 2^(11-EW)) for weights; the
 * pack's 16-byte trailer keeps max |w * scale|, from which the kernels derive EW.  Same buffer sizes and shapes as the fp16 forms.
 * The conv then runs on the wide-tile kernel only (csrc/wswide.hip: Cin % 32 == 0, Cin >= 64, W % 32 == 0, H % 16 == 0, d and
 * bias given) and fails otherwise; sgdfr_modconv2d_wsplit_wide() = 1 for the shapes whose F(4,3) launch takes that kernel by its
 * tile count anyway (what a host asks before choosing the arithmetic for a layer).  ~5e-5 of max|y| per layer (fp16x3: 4e-6),
 * 1.3-1.5x the speed of the three-product form (scripts/f8_layer_probe.py). */
int sgdfr_modconv2d_wsplit_supported(int B, int Cin, int Cout, int H, int W, int f);
int sgdfr_modconv2d_wsplit_wide(int B, int Cin, int Cout, int H, int W);
/* OR-ed into `arith` of sgdfr_modconv2d_wsplit_f32 (f = 4, either tile size, xs_out given): the hand-over's lo chunks leave as the fp8
 * cross-term operands of SGDFR_SPLIT_FP16F8 -- for a next conv launched with that arithmetic (sgdfr_modconv2d_split_f32, mode UP3,
 * where sgdfr_modconv2d_split_f8_ok() = 1: the transposed conv's deep plan, nine taps pair up per phase with (1,1) beside zeros;
 * mode PLAIN3 where it says so: the 4-wave plan of short-K layers, fed by sgdfr_blur_bias_act_split_f32(arith = FP16F8, wino = 0)). */
#define SGDFR_SPLIT_HANDOVER_F8 0x100
int64_t sgdfr_modconv_prepack_wsplit_elems(int Cout, int Cin, int f);
int sgdfr_modconv_prepack_wsplit_f32(const float* weight, unsigned short* wsp, int Cout, int Cin, int f, int arith,
                                     unsigned int* sat, void* stream);
int sgdfr_to_wsplit_f32(const float* x, const float* s, unsigned short* vs, int B, int Cin, int H, int W, int f, int arith,
                        unsigned int* sat, void* stream);
int sgdfr_modconv2d_wsplit_f32(const unsigned short* v, const unsigned short* wsp, const float* d, const float* noise,
                               int64_t noise_bstride, const float* noise_w, const float* bias, const float* zeros, float* y,
                               const float* rgb_w, const float* rgb_s, float* rgb_part, unsigned short* xs_out,
                               const float* s_next, int B, int Cin, int Cout, int H, int W, int f, int arith, int act,
                               float slope, float gain, unsigned int* sat, void* stream);

/* The rest of ToRGB.forward (model.py:350-359) when its 1x1 conv was accumulated by sgdfr_modconv2d_split_f32(rgb_part):
 *   y[b,j,p] = sum_{t<T} part[b, t*3+j, p] + bias[j] + (skip ? upfirdn2d(skip[b,j], fir[4,4], up=2, pad=(2,1))[p] : 0) */
int sgdfr_torgb_finish_f32(const float* part, int T, const float* bias, const float* skip, const float* fir, float* y, int B,
                           int H, int W, void* stream);

/* sgdfr_torgb_finish_f32 whose result leaves as uint8 HWC instead of fp32 planes -- the output side of the path fused into the
 * last ToRGB (scaling of libs/utilities/image_utils.py:87-110, identical to sgdfr_image_to_u8_f32 applied to
 * the fp32 result):  y[b][oy][x_offset + ox][c] = u8(rgb[b, swap_rb ? 2-c : c, oy, ox]),  row_pitch bytes per row, so the
 * frame can be written as one panel of a wider grid (x_offset, row_pitch multiples of 4). */
int sgdfr_torgb_finish_u8_f32(const float* part, int T, const float* bias, const float* skip, const float* fir,
                              unsigned char* y, int64_t row_pitch, int x_offset, int swap_rb, int B, int H, int W,
                              void* stream);

/* x [B,3,H,W] fp32 -> y [B,H,W,3] uint8:  trunc( (clamp(x,-1,1) + 1) / (2 + 1e-5) * 255 )
 * (libs/utilities/image_utils.py:87-110 tensor_to_image / torch_range_1_to_255, then the writers' uint8 cast) */
int sgdfr_image_to_u8_f32(const float* x, unsigned char* y, int B, int H, int W, void* stream);

/* K panels [B,3,H,W] fp32 side by side -> y [B,H,K*W,3] uint8 video frames, same scaling as above.
 * `panels` / `bstrides` are HOST arrays of K device pointers / batch strides in floats (0 = the same image in
 * every frame, e.g. the source; a NULL panel is skipped -- its columns keep what sgdfr_torgb_finish_u8_f32 wrote).
 * Batched generate_grid_image + tensor_to_image + np.uint8
 * (libs/utilities/utils_inference.py:11-33, run_inference.py:188-194); swap_rb != 0 also applies that path's
 * cv2.cvtColor(.., COLOR_BGR2RGB) channel swap. */
#define SGDFR_MAX_GRID_PANELS 4
int sgdfr_grid_to_u8_f32(const float* const* panels, const int64_t* bstrides, int K, unsigned char* y, int B, int H,
                         int W, int swap_rb, void* stream);

/* ---- shift-vector construction (the DirectionMatrix input), SURVEY.md §8f-2 ---------------------------------------- */

/* One learned direction: which 3DMM parameter feeds it and how it is placed on the shift axis.
 *   SGDFR_DIR_ANGLE: position = angle[col] * a / b      (a = shift_scale, b = angle_scales[col]; col 0 yaw, 1 pitch, 2 roll)
 *   SGDFR_DIR_JAW:   position = a * pose[col] + b       (col = 3 in the reference, params['pose'][:, 3])
 *   SGDFR_DIR_EXP:   position = a * alpha_exp[col] + b  (a, b from the line through (min,-shift_scale), (max,+shift_scale),
 *                                                        libs/utilities/generic.py:84-104)
 *   SGDFR_DIR_ZERO:  the direction is not driven (its entry stays 0) */
#define SGDFR_DIR_ZERO 0
#define SGDFR_DIR_ANGLE 1
#define SGDFR_DIR_JAW 2
#define SGDFR_DIR_EXP 3
#define SGDFR_MAX_DIRECTIONS 64
struct sgdfr_direction {
    int kind;
    int col;
    double a;
    double b;
};

/* shift[n,k] = position_k(target n) - position_k(source n) for N frames in one launch: the batched, sync-free counterpart of
 * run_inference.py:201-254 Inference.make_shift (arith 0: float32 angle scaling, float64 divide / a*x+b / difference, one
 * final rounding -- the numpy scalar arithmetic of that function) and of libs/utilities/utils_train.py:127-175
 * make_shift_vector (arith 1: every operation rounded to float32 as torch does on float32 tensors; division is a true
 * division as in torch's CPU kernel -- the CUDA kernel multiplies by a rounded reciprocal, 1 ulp apart).
 * Source arrays have batch strides in floats (0 = one source identity for all frames, the run_inference case); target
 * arrays are contiguous [N,3], [N,pose_dim], [N,exp_dim].  `table` is a HOST array of D entries (passed by value to the
 * kernel). */
int sgdfr_make_shift_f32(const float* ang_s, int64_t ang_s_bstride, const float* pose_s, int64_t pose_s_bstride,
                         const float* exp_s, int64_t exp_s_bstride, const float* ang_t, const float* pose_t,
                         const float* exp_t, int pose_dim, int exp_dim, const struct sgdfr_direction* table, int D,
                         float* shift, int N, int arith, void* stream);

/* Second half of make_shift_vector_50 (libs/utilities/utils_train.py:227-286): sample n moves along ONE direction which[n]
 * (device int32 [N]) by (lo - hi) * u[n] + hi with lo/hi = -/+shift_scale - position(source n), float32 arithmetic; every
 * other entry of its row is 0.  which / u are drawn by the caller (np.random.choice / torch.rand in the reference). */
int sgdfr_make_shift_random_f32(const float* ang_s, const float* pose_s, const float* exp_s, int pose_dim, int exp_dim,
                                const int* which, const float* u, float shift_scale, const struct sgdfr_direction* table,
                                int D, float* shift, int N, void* stream);

/* ---- backward helpers (autograd of model.py:232-359 as restated in SURVEY.md Appendix C) ------------------------ */

/* Activation gradient + per-(b,c) reductions (op/fused_act.py:19-37 and the adjoints of model.py:287, :240):
 *   g_pre = g_out * (out > 0 ? 1 : slope) * gain
 *   sums[b,c,0] = sum g_pre ; sums[b,c,1] = sum g_pre * noise ;
 *   sums[b,c,2] = sum g_pre * y (want_y != 0), y = lrelu^-1(out) - noise_w*noise - bias[c]  (= d * conv output)
 * g_absmax (optional, [B,C] words): fp32 bit pattern of max |g_pre| per (image, channel) plane, from the same pass -- the range
 * plan of the fp16-split dL/dx conv that consumes g_pre (sgdfr_split_range_f32 with x_absmax_bstride = C). */
int sgdfr_act_grad_reduce_f32(const float* g_out, const float* out, const float* noise, int64_t noise_bstride,
                              const float* noise_w, const float* bias, float* g_pre, float* sums, int B, int C, int HW,
                              float slope, float gain, int want_y, unsigned int* g_absmax, void* stream);

/* Adjoint of the FIR of sgdfr_blur_bias_act_f32 (op/upfirdn2d.py:104-121 for pad (1,1)): g [B,C,2H,2W] ->
 * gt parity planes [B,C,4,H+1,W+1]; with t (forward planes) also asum[b,c] = sum gt * t. */
int sgdfr_blur_adjoint_f32(const float* g, const float* fir, const float* t, float* gt, float* asum, int B, int C, int H,
                           int W, void* stream);

/* dx = gu * s[b,c] (dx may alias gu) ; r[b,c] = sum_q x[b,c,q] * gu[b,c,q]   (x_bstride 0 = broadcast constant) */
int sgdfr_scale_reduce_f32(const float* gu, const float* x, int64_t x_bstride, const float* s, float* dx, float* r, int B,
                           int C, int HW, void* stream);

/* ToRGB backward: dx[b,i,p] = s[b,i]/sqrt(Cin) * sum_j w_rgb[j,i] g[b,j,p] ; r[b,j,i] = sum_p x[b,i,p] g[b,j,p] */
int sgdfr_torgb_bwd_f32(const float* x, const float* g, const float* w_rgb, const float* s, float* dx, float* r, int B,
                        int Cin, int H, int W, void* stream);

/* One pass over a saved StyledConv activation `out` [B,C,HW] in the backward of the frozen generator (libs/trainer.py:177-189:
 * only A is optimised).  The gradient of `out` arrives as up to three terms,
 *     g = gu * s_next[b,c]                                             (gu [B,C,HW] = dL/d(x*s) of the conv that reads `out`)
 *       + s_rgb[b,c]/sqrt(C) * sum_j w_rgb[j,c] * g_rgb[b,j,p]         (g_rgb [B,3,HW] = gradient of the ToRGB that reads `out`)
 *       + g_add                                                        (any of the three may be NULL, not all)
 * and the pass returns, besides everything sgdfr_act_grad_reduce_f32 computes from g for the layer that PRODUCED `out`
 * (g_pre, sums [B,C,3], g_absmax [B,C]):  r_next[b,c] += sum_p out*gu  (sgdfr_scale_reduce_f32's r)  and
 * r_rgb[b,j,c] += sum_p out*g_rgb[b,j]  (sgdfr_torgb_bwd_f32's r).  zero_outputs != 0: the call zeroes sums / g_absmax / r_next /
 * r_rgb first; 0: the caller carved them out of a zeroed workspace (one memset per backward instead of four per layer). */
int sgdfr_grad_join_f32(const float* out, const float* gu, const float* s_next, const float* g_rgb, const float* w_rgb,
                        const float* s_rgb, const float* g_add, const float* noise, int64_t noise_bstride, const float* noise_w,
                        const float* bias, float* g_pre, float* sums, unsigned int* g_absmax, float* r_next, float* r_rgb, int B,
                        int C, int HW, float slope, float gain, int want_y, int zero_outputs, void* stream);

/* Backward of sgdfr_styles_batched_f32 with respect to the latent (frozen weights), all layers in two launches:
 *   ds_l = gs_l + s_l * ((-(a_l / d_l) * d_l^3) @ qt_l^T)      demodulated 3x3 conv (a = d * dL/dd: sums[:,:,2] of the activation-
 *                                                             gradient pass with a_stride 3, or the blur adjoint's asum, stride 1)
 *   ds_l = gs_l                                                a == NULL
 *   ds_l[b,i] = (sum_j rgb_r[b,j,i] * rgb_w[j,i]) / sqrt(cin)  ToRGB (rgb_r [B,3,cin] = sum_p x*g_j, rgb_w [3,cin])
 *   glat[b, latent_index_l, :] = sum_l ds_l[b,:] @ mod_w_l / sqrt(D);  latent rows no layer reads are written as zeros. */
typedef struct sgdfr_style_grad_layer {
    const float* gs;    /* [B, cin] or NULL (rgb_r given) */
    const float* rgb_r; /* [B, 3, cin] or NULL */
    const float* rgb_w; /* [3, cin] */
    const float* a;     /* [B, cout] with element stride a_stride, or NULL */
    const float* d;     /* [B, cout] */
    const float* s;     /* [B, cin] */
    const float* qt;    /* [cin, cout] */
    const float* mod_w; /* [cin, D] */
    float* ds;          /* [B, cin] out */
    float* gmod_w;      /* [cin, D] out or NULL: dL/d(modulation weight) = ds^T @ latent[:, latent_index] / sqrt(D)  (model.py:148-157) */
    float* gmod_b;      /* [cin] out or NULL:    dL/d(modulation bias)   = sum_b ds */
    int64_t a_stride;
    int cin, cout, latent_index;
} sgdfr_style_grad_layer;
/* latent [B,L,D]: needed (non-NULL) only when some layer asks for gmod_w / gmod_b (PTI trains them, libs/optimization.py:31-40);
 * glat may be NULL when only those are wanted. */
int sgdfr_styles_batched_bwd_f32(const sgdfr_style_grad_layer* layers, int n_layers, const float* latent, float* glat, int B, int L,
                                 int D, void* stream);

/* dq[o,i] = sum_b (-0.5 * (a/d) * d^3)[b,o] * s[b,i]^2 = dL/dQ of the demodulation (Q = sum_k Wc^2), the term
 * sgdfr_modconv_wgrad_finish*_f32 adds to the weight gradient; a = d * dL/dd with element stride a_stride (see above). */
int sgdfr_demod_dq_f32(const float* a, int64_t a_stride, const float* d, const float* s, float* dq, int B, int Cin, int Cout,
                       void* stream);

/* The small parameter gradients of one generator backward in ONE launch (the reference gets them from autograd's sum ops:
 * op/fused_act.py:32-37 for the bias, model.py:287 for the noise strength, model.py:350-359 for ToRGB):
 *   SGDFR_PGRAD_BIAS   out[c]     = sum_b in[b,c,0]                       in = sums [B,C,3] of the activation-gradient pass
 *   SGDFR_PGRAD_NOISE  out[0]     = sum_{b,c} in[b,c,1]
 *   SGDFR_PGRAD_RGB_W  out[j,i]   = scale * sum_b in[b,j,i] * aux[b,i]    in = r_rgb [B,3,C], aux = s [B,C], scale = 1/sqrt(C)
 *   SGDFR_PGRAD_RGB_B  out[j]    += sum_{b,p} in[b,j,p]                   in = g_rgb [B,3,HW]; out must be ZEROED by the caller */
#define SGDFR_PGRAD_BIAS 0
#define SGDFR_PGRAD_NOISE 1
#define SGDFR_PGRAD_RGB_W 2
#define SGDFR_PGRAD_RGB_B 3
#define SGDFR_MAX_PARAM_GRADS 64
typedef struct sgdfr_param_grad {
    const float* in;
    const float* aux;
    float* out;
    int kind, C, HW;
    float scale;
} sgdfr_param_grad;
int sgdfr_param_grads_f32(const sgdfr_param_grad* entries, int n, int B, void* stream);

/* One Adam step (torch.optim.Adam as libs/optimization.py:41,66-68 uses it: no weight decay, no amsgrad) over a list of parameter
 * tensors in ONE launch: p, g (gradient), m (exp_avg), v (exp_avg_sq), n elements each; step[0] = the step count t >= 1 as a float on
 * the device (the caller increments it before the call, so a captured step needs no host value):
 *   m += (g - m)(1 - beta1) ; v = v beta2 + (1 - beta2) g^2 ; p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps) */
#define SGDFR_MAX_ADAM_TENSORS 96
typedef struct sgdfr_adam_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} sgdfr_adam_tensor;
int sgdfr_adam_f32(const sgdfr_adam_tensor* tensors, int n, const float* step, float lr, float beta1, float beta2, float eps, void* stream);

/* ds[b,i] = gs[b,i] + s[b,i] * sum_o (-gd[b,o] * d[b,o]^3) * qt[i,o]   (chain rule through d = rsqrt(sum s^2 q + eps)) */
int sgdfr_demod_grad_f32(const float* gd, const float* d, const float* qt, const float* s, const float* gs, float* ds,
                         int B, int Cin, int Cout, void* stream);

/* Weight gradient of the modulated conv in the packed layout (split-K fp32 MFMA, atomically accumulated):
 *   dwp[i][ky*3+kx][o] = sum_{b,p} (d*g)[b,o,shifted p] * (x*s)[b,i,p]
 * mode PLAIN3: g = activation-gradient [B,Cout,H,W];  mode UP3: g = parity planes [B,Cout,4,H+1,W+1] from
 * sgdfr_blur_adjoint_f32.  x [B,Cin,H,W] (x_bstride 0 = broadcast), dwp [Cin,9,Cout] (zeroed here). */
int sgdfr_modconv_wgrad_f32(const float* g, const float* d, const float* x, int64_t x_bstride, const float* s, float* dwp,
                            int B, int Cin, int Cout, int H, int W, int mode, void* stream);

/* dweight[o][i][k] = (dwp[i][k][o] + 2 * wp[i][k][o] * dq[o][i]) / sqrt(9*Cin):  un-pack, add the demodulation path
 * (dq = dL/dQ [Cout,Cin], NULL without demodulation) and apply the equalised-lr scale of model.py:215,236. */
int sgdfr_modconv_wgrad_finish_f32(const float* dwp, const float* wp, const float* dq, float* dweight, int Cout, int Cin,
                                   void* stream);
/* The same gradient without atomics (deterministic, and ~0.3 ms faster per layer: every layer of the 256x256 generator
 * ends up with ~9.4 M atomic adds): sgdfr_modconv_wgrad_ksplit() = number of pixel slices K of a shape (0: the shape needs
 * the direct kernel, use sgdfr_modconv_wgrad_f32); ..._parts_f32 stores slice sums to part [K][9][Cout][Cin]; ..._finish_parts_f32
 * adds the slices in fixed order and applies what sgdfr_modconv_wgrad_finish_f32 applies. */
int sgdfr_modconv_wgrad_ksplit(int B, int Cin, int Cout, int H, int W, int mode);
int sgdfr_modconv_wgrad_parts_f32(const float* g, const float* d, const float* x, int64_t x_bstride, const float* s, float* part,
                                  int B, int Cin, int Cout, int H, int W, int mode, void* stream);
int sgdfr_modconv_wgrad_finish_parts_f32(const float* part, int ksplit, const float* wp, const float* dq, float* dweight,
                                         int Cout, int Cin, void* stream);
/* sgdfr_modconv_wgrad_finish_parts_f32 with the demodulation term taken from the ORIGINAL weight tensor weight [Cout][Cin][3][3]
 * (= wp / scale transposed) instead of the packed copy: every stream of the launch is then contiguous (the packed form forces 4-byte
 * gathers at a stride of 9*Cout floats); slices are summed four-way interleaved, in fixed order (deterministic, not the bits of the
 * sequential sum).  Instead of dq the call may take what sgdfr_demod_dq_f32 takes -- a (= d * dL/dd, element stride a_stride), d [B,Cout],
 * s [B,Cin] -- and forms dq itself (same expression and order): one launch less per layer. */
int sgdfr_modconv_wgrad_finish_parts_oik_f32(const float* part, int ksplit, const float* weight, const float* dq, const float* a,
                                             int64_t a_stride, const float* d, const float* s, int B, float* dweight, int Cout, int Cin,
                                             void* stream);

/* Measurement aid (csrc/probe.hip; no reference counterpart): the rate v_mfma_f32_32x32x16_{f16,bf16} sustains on THIS device,
 * in 16-bit TFLOP/s -- arith SGDFR_SPLIT_FP16/BF16; lds_fragments 1: operands re-read from LDS at the split conv's ratio
 * (8 ds_read_b128 per 12 MFMAs), 0: register operands; random_operands 1: random mantissas, 0: zeros.  The chip clocks to its
 * power budget and MFMA power follows operand toggling, so the random-operand figure (not the nominal 2.5 PFLOP/s) is what a
 * kernel on real data can reach; bench.py prints it beside the roofline.  blocks <= 0: 256 (one 8-wave block per CU);
 * scratch: blocks*512 floats of device memory.  Synchronises `stream`. */
int sgdfr_mfma_ceiling_probe(int arith, int lds_fragments, int random_operands, int iters, int blocks, float* scratch,
                             double* tflops16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGDFR_H */
