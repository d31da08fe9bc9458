/* Experiments that measured SLOWER than the product path and were moved out of libsgdfr_hip.so in round 5.
 * Built on demand by experiments/build.py into experiments/libsgdfr_experiments.so (links against the product library);
 * same conventions as include/sgdfr.h (device pointers, int return, sgdfr_last_error()).  Not part of the drop-in ABI. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- The upsampling StyledConv in ONE launch (experiments/csrc/upfir.hip): sgdfr_modconv2d_split_f32(mode UP3, x_is_split = 1) followed by
 * sgdfr_blur_bias_act_split_f32 without the fp32 parity planes in between -- the stride-2 transposed 3x3 modulated conv
 * (ModulatedConv2d.forward model.py:246-256 conv_transpose2d), the 4x4 FIR (Blur model.py:72-88 = upfirdn2d op/upfirdn2d.py:168-209,
 * pad (1,1)), NoiseInjection (model.py:284-293) and FusedLeakyReLU (op/fused_act.py:79-86), result times the NEXT layer's
 * modulation in that layer's split input form.  Blocks work on TR x TC patches of the (H+1) x (W+1) super-pixel grid and
 * recompute a one-super-pixel halo ring; the sums are formed in the two-pass form's order (bit-identical output).
 *   xs_in  [B][Cin/8][2][H*W][8] 16-bit   the input, already x * s in split form (sgdfr_to_split_f32 / a producer's xs_out)
 *   wsp                                   sgdfr_modconv_prepack_split_f32(transpose_flip = 0) pack, same arith
 *   d [B,Cout], fir [4,4], noise [B or 1][2H*2W] (noise_bstride 0 = shared; 8-byte aligned), noise_w [1], bias [Cout],
 *   s_next [B,Cout], zeros (>= 16 zero bytes), xs_out [B][Cout/8][2][2H*2W][8] 16-bit
 *   sgdfr_modconv2d_upfir_supported: Cin % 16 == 0, Cout % 64 == 0 and the patch fits LDS
 *   sgdfr_modconv2d_upfir_tiles: tiles per image of the patch tiling (0: unsupported), patch shape in *TR / *TC */
int sgdfr_modconv2d_upfir_supported(int B, int Cin, int Cout, int H, int W);
int sgdfr_modconv2d_upfir_tiles(int B, int Cin, int Cout, int H, int W, int* TR, int* TC);
int sgdfr_modconv2d_upfir_split_f32(const unsigned short* xs_in, const unsigned short* wsp, const float* d, const float* fir,
                                    const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                                    const float* s_next, const float* zeros, unsigned short* xs_out, int B, int Cin, int Cout,
                                    int H, int W, int arith, int act, float slope, float gain, unsigned int* sat, void* stream);

/* sgdfr_modconv2d_split_f32(mode = SGDFR_MODE_UP3, x_is_split = 1, plane_stride != 0) as a role-swapping kernel (experiments/csrc/uppp.hip): two wave
 * groups per block, one running a tile's MFMAs while the other DMAs its next operands and stores its own finished tile's planes, so
 * the plane stores leave the matrix cores' critical path.  Same arguments' meaning, same bits in y [B][Cout][plane_stride][px][py].
 *   supported: Cin % 16 == 0, Cout % 64 == 0, plane_stride >= (H+1)*(W+1) and > 129 + W (a 128-position tile touches <= 2 images) */
int sgdfr_modconv2d_up_pp_supported(int B, int Cin, int Cout, int H, int W, int64_t plane_stride);
int sgdfr_modconv2d_up_pp_f32(const unsigned short* xs_in, const unsigned short* wsp, const float* d, const float* zeros, float* y,
                              int B, int Cin, int Cout, int H, int W, int64_t plane_stride, int arith, void* stream);

#ifdef __cplusplus
}
#endif
