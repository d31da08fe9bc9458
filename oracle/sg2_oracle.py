"""ORACLE -- test infrastructure only.  NOT part of the product path.

CPU restatement (PyTorch-CPU tensor ops, fp32 or fp64) of the reference's StyleGAN2
generator hot path, written functionally over a plain ``{state_dict key: tensor}``
mapping.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the shipped package
(``stylegan_directions_face_reenactment_amd``) never does and has no CPU fallback.

Parity pin: the reference publishes no tests or golden vectors for this path
(SURVEY.md §4), so this oracle is pinned by importing the reference itself on CPU in
the build container (``oracle/make_golden.py``, SURVEY.md Appendix E shims) and
(i) asserting oracle == reference on full tensors there, (ii) committing small
fixtures under ``tests/golden/`` that the CPU test-suite re-checks everywhere.

Every function cites the reference lines it restates (paths relative to
/root/reference).  The arithmetic follows the reference's *own* formulation
(per-sample modulated weights + grouped convolution), not the shared-weight algebra
the HIP kernels use, so agreement between the two is a real check.
"""
import math

import numpy
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# ----------------------------------------------------------------------------- ops

def make_fir(taps, gain=1.0, dtype=torch.float32):
    """libs/gan/StyleGAN2/model.py:19-27 -- outer product of 1-D taps, normalised to sum 1."""
    k = torch.tensor(taps, dtype=dtype)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum() * gain


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """libs/gan/StyleGAN2/op/upfirdn2d.py:149-209 (native CPU semantics).

    x [B,C,H,W]; zero-stuff by `up`, pad (negative pad crops), correlate with the
    *flipped* kernel (true convolution), keep every `down`-th sample.
    pad is (p0, p1) applied to both axes or (x0, x1, y0, y1); up / down are one factor for both axes or
    (x, y) pairs -- `upfirdn2d_native` itself takes up_x, up_y, down_x, down_y (upfirdn2d.py:168-170).
    Written as an explicit sum over FIR taps (no conv library call).
    """
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    px0, px1, py0, py1 = pad
    up_x, up_y = up if isinstance(up, (tuple, list)) else (up, up)
    down_x, down_y = down if isinstance(down, (tuple, list)) else (down, down)
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    u = x.new_zeros(B, C, H * up_y, W * up_x)
    u[:, :, ::up_y, ::up_x] = x
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0): u.shape[2] - max(-py1, 0), max(-px0, 0): u.shape[3] - max(-px1, 0)]
    fh, fw = u.shape[2] - kh + 1, u.shape[3] - kw + 1
    acc = x.new_zeros(B, C, fh, fw)
    for a in range(kh):
        for b in range(kw):
            acc = acc + u[:, :, a:a + fh, b:b + fw] * kernel[kh - 1 - a, kw - 1 - b].to(x.dtype)
    out = acc[:, :, ::down_y, ::down_x]
    oh = (H * up_y + py0 + py1 - kh + down_y) // down_y
    ow = (W * up_x + px0 + px1 - kw + down_x) // down_x
    assert out.shape[2:] == (oh, ow), (out.shape, oh, ow)
    return out


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """op/fused_bias_act_kernel.cu:26-47 with act=3, grad=0:  lrelu(x + b[c]) * scale."""
    if bias is not None and bias.numel() > 0:
        x = x + bias.view(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope) * scale


def pixel_norm(x):
    """model.py:11-16."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """model.py:129-157: weight stored /lr_mul, runtime scale lr_mul/sqrt(in)."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias * lr_mul)


# ---------------------------------------------------------------------- modconv

def up_blur_pad(blur_len=4, kernel_size=3, factor=2):
    """model.py:198-204 pad of the Blur after the transposed conv: (1, 1) for 4-tap/3x3."""
    p = (blur_len - factor) - (kernel_size - 1)
    return ((p + 1) // 2 + factor - 1, p // 2 + 1)


def modulated_conv2d(x, style, weight, mod_weight, mod_bias, demodulate=True,
                     upsample=False, blur_taps=(1, 3, 3, 1)):
    """model.py:232-273.  weight [1,Cout,Cin,k,k]; style [B,style_dim]."""
    B, Cin, H, W = x.shape
    _, Cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_weight, mod_bias).view(B, 1, Cin, 1, 1)       # :235
    w = (1.0 / math.sqrt(Cin * k * k)) * weight * s                            # :236
    if demodulate:
        d = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8)                        # :239
        w = w * d.view(B, Cout, 1, 1, 1)
    if upsample:                                                               # :246-257
        wt = w.transpose(1, 2).reshape(B * Cin, Cout, k, k)
        out = F.conv_transpose2d(x.reshape(1, B * Cin, H, W), wt, padding=0, stride=2, groups=B)
        out = out.view(B, Cout, out.shape[2], out.shape[3])
        fir = make_fir(list(blur_taps), gain=4.0, dtype=x.dtype)               # :76-79
        return upfirdn2d(out, fir, pad=up_blur_pad(len(blur_taps), k))
    out = F.conv2d(x.reshape(1, B * Cin, H, W), w.view(B * Cout, Cin, k, k),
                   padding=k // 2, groups=B)                                   # :267-271
    return out.view(B, Cout, out.shape[2], out.shape[3])


def styled_conv(P, prefix, x, style, noise, upsample):
    """model.py:331-337: modconv -> + noise_w*noise -> fused bias + lrelu*sqrt2."""
    out = modulated_conv2d(x, style, P[prefix + '.conv.weight'],
                           P[prefix + '.conv.modulation.weight'],
                           P[prefix + '.conv.modulation.bias'],
                           demodulate=True, upsample=upsample)
    if noise is None:                                                          # :283-285
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3], dtype=out.dtype)
    out = out + P[prefix + '.noise.weight'] * noise                            # :287
    return fused_leaky_relu(out, P[prefix + '.activate.bias'])


def to_rgb(P, prefix, x, style, skip=None):
    """model.py:350-359: 1x1 modconv without demod + bias (+ FIR-upsampled skip)."""
    out = modulated_conv2d(x, style, P[prefix + '.conv.weight'],
                           P[prefix + '.conv.modulation.weight'],
                           P[prefix + '.conv.modulation.bias'], demodulate=False)
    out = out + P[prefix + '.bias']
    if skip is not None:
        fir = make_fir([1, 3, 3, 1], gain=4.0, dtype=x.dtype)                  # :35
        out = out + upfirdn2d(skip, fir, up=2, down=1, pad=(2, 1))             # :38-46
    return out


# -------------------------------------------------------------------- generator

def n_mlp_of(P):
    return sum(1 for k in P if k.startswith('style.') and k.endswith('.weight'))


def mapping(P, z, lr_mlp=0.01):
    """model.py:378-387: PixelNorm + n_mlp x EqualLinear(lr_mul, fused_lrelu)."""
    h = pixel_norm(z)
    for i in range(1, n_mlp_of(P) + 1):
        h = equal_linear(h, P['style.%d.weight' % i], P['style.%d.bias' % i], lr_mul=lr_mlp,
                         activation=True)
    return h


def mean_latent_from(P, z_batch):
    """model.py:460-466 with the z batch injected instead of drawn from torch.randn."""
    return mapping(P, z_batch).mean(0, keepdim=True)


def num_layers_of(P):
    return sum(1 for k in P if k.startswith('noises.noise_'))


def generator_forward(P, styles, return_latents=False, truncation=1, truncation_latent=None,
                      input_is_latent=False, noise=None, randomize_noise=False,
                      return_layers=False):
    """model.py:471-539 (single-style branch; 2-style mixing :510-517 is unused by the repo)."""
    num_layers = num_layers_of(P)
    n_latent = num_layers + 1
    if not input_is_latent:
        styles = [mapping(P, s) for s in styles]                               # :484-485
    if noise is None:                                                          # :488-492
        noise = [None] * num_layers if randomize_noise else \
            [P['noises.noise_%d' % i] for i in range(num_layers)]
    if truncation < 1:                                                         # :494-500
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    assert len(styles) == 1
    latent = styles[0]
    if latent.ndim < 3:                                                        # :504-508
        latent = latent.unsqueeze(1).repeat(1, n_latent, 1)
    B = latent.shape[0]
    layers = {}
    out = P['input.input'].repeat(B, 1, 1, 1)                                  # :296-300
    out = styled_conv(P, 'conv1', out, latent[:, 0], noise[0], upsample=False)  # :520
    layers['conv1'] = out
    skip = to_rgb(P, 'to_rgb1', out, latent[:, 1])                             # :521
    layers['to_rgb1'] = skip
    i = 1
    for k in range((num_layers - 1) // 2):                                     # :526-532
        out = styled_conv(P, 'convs.%d' % (2 * k), out, latent[:, i], noise[2 * k + 1], upsample=True)
        layers['convs.%d' % (2 * k)] = out
        out = styled_conv(P, 'convs.%d' % (2 * k + 1), out, latent[:, i + 1], noise[2 * k + 2], upsample=False)
        layers['convs.%d' % (2 * k + 1)] = out
        skip = to_rgb(P, 'to_rgbs.%d' % k, out, latent[:, i + 2], skip)
        layers['to_rgbs.%d' % k] = skip
        i += 2
    res = (skip, latent if return_latents else None)                           # :536-539
    return res + (layers,) if return_layers else res


# ------------------------------------------------------ DirectionMatrix and glue

def direction_matrix(A, x, input_dim=15, shift_dim=512, num_layers=8, w_plus=True):
    """libs/models/direction_matrix.py:41-48."""
    x = x.reshape(-1, input_dim)
    out = F.linear(x, A['linear.weight'], A.get('linear.bias'))
    if w_plus:
        out = out.view(len(x), num_layers, shift_dim)
    return out


def get_shifted_latent_code(P, z, shift, input_is_latent=False, w_plus=False, num_layers=None):
    """libs/utilities/generic.py:116-135."""
    n_latent = num_layers_of(P) + 1
    if not input_is_latent:
        latent = mapping(P, z).unsqueeze(1).repeat(1, n_latent, 1)             # :119-120
    else:
        latent = z.clone()                                                     # :122
    if not w_plus:
        if num_layers is None:
            latent = latent + shift.unsqueeze(1)                               # :124-127
        else:
            latent = latent.clone()
            latent[:, :num_layers] = latent[:, :num_layers] + shift.unsqueeze(1)  # :129-130
    else:
        latent = torch.cat([latent[:, :shift.shape[1]] + shift, latent[:, shift.shape[1]:]], 1)  # :133
    return latent


def generate_image(P, latent_code, truncation, trunc, w_plus=True, num_layers_shift=8,
                   shift_code=None, input_is_latent=False, return_latents=False):
    """libs/utilities/generic.py:137-151 (the >256 pooling branch never fires at size 256)."""
    if shift_code is None:
        img, lat = generator_forward(P, [latent_code], return_latents=return_latents,
                                     truncation=truncation, truncation_latent=trunc,
                                     input_is_latent=input_is_latent)
    else:
        code = get_shifted_latent_code(P, latent_code, shift_code, input_is_latent=input_is_latent,
                                       w_plus=w_plus, num_layers=num_layers_shift)
        img, lat = generator_forward(P, [code], return_latents=return_latents, truncation=truncation,
                                     truncation_latent=trunc, input_is_latent=True)
    if img.shape[2] > 256:
        img = F.adaptive_avg_pool2d(img, (256, 256))
    return (img, lat) if return_latents else img


# ------------------------------------------------------------------ state layout

def generator_state_shapes(size=256, style_dim=512, n_mlp=8, channel_multiplier=2):
    """Key -> shape of Generator.state_dict() (model.py:362-447)."""
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
          128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
          512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    log_size = int(math.log2(size))
    S = {}
    for i in range(1, n_mlp + 1):
        S['style.%d.weight' % i] = (style_dim, style_dim)
        S['style.%d.bias' % i] = (style_dim,)
    S['input.input'] = (1, ch[4], 4, 4)

    def styled(prefix, cin, cout, up):
        S[prefix + '.conv.weight'] = (1, cout, cin, 3, 3)
        if up:
            S[prefix + '.conv.blur.kernel'] = (4, 4)
        S[prefix + '.conv.modulation.weight'] = (cin, style_dim)
        S[prefix + '.conv.modulation.bias'] = (cin,)
        S[prefix + '.noise.weight'] = (1,)
        S[prefix + '.activate.bias'] = (cout,)

    def rgb(prefix, cin, up):
        S[prefix + '.bias'] = (1, 3, 1, 1)
        if up:
            S[prefix + '.upsample.kernel'] = (4, 4)
        S[prefix + '.conv.weight'] = (1, 3, cin, 1, 1)
        S[prefix + '.conv.modulation.weight'] = (cin, style_dim)
        S[prefix + '.conv.modulation.bias'] = (cin,)

    styled('conv1', ch[4], ch[4], False)
    rgb('to_rgb1', ch[4], False)
    cin = ch[4]
    for j, i in enumerate(range(3, log_size + 1)):      # module registration order: convs, to_rgbs, noises
        cout = ch[2 ** i]
        styled('convs.%d' % (2 * j), cin, cout, True)
        styled('convs.%d' % (2 * j + 1), cout, cout, False)
        cin = cout
    for j, i in enumerate(range(3, log_size + 1)):
        rgb('to_rgbs.%d' % j, ch[2 ** i], True)
    for l in range((log_size - 2) * 2 + 1):
        r = 2 ** ((l + 5) // 2)
        S['noises.noise_%d' % l] = (1, 1, r, r)
    return S


def template_state(size=256, style_dim=512, n_mlp=8, channel_multiplier=2):
    """Zero tensors with the reference's key set; FIR buffers hold their constructor values."""
    T = {}
    for k, shp in generator_state_shapes(size, style_dim, n_mlp, channel_multiplier).items():
        T[k] = make_fir([1, 3, 3, 1], gain=4.0) if k.endswith('.kernel') else torch.zeros(shp)
    return T


def cast_state(P, dtype):
    return {k: v.to(dtype) for k, v in P.items()}


def tensor_to_uint8_hwc(images):
    """[B,3,H,W] in [-1,1] -> [B,H,W,3] uint8: tensor_to_image (libs/utilities/image_utils.py:97-110: clamp, +1,
    /(2+1e-5), *255) followed by the np.uint8 truncation of generate_video (libs/utilities/utils_inference.py:16)."""
    x = images.detach().to(torch.float32).clone()
    x.clamp_(min=-1, max=1).add_(1).div_(2 + 1e-5)
    x = x.mul(255.0)
    return x.permute(0, 2, 3, 1).contiguous().numpy().astype('uint8')


def grid_video_frames(source, target, reenacted, swap_rb=True):
    """Per frame i: generate_grid_image(source, target[i], reenacted[i]) puts the three images side by side
    (utils_inference.py:20-29, called with batch 1 at run_inference.py:188), tensor_to_image + cvtColor(BGR2RGB)
    (run_inference.py:193) turn it into the frame handed to the video writer.  -> [N,H,3W,3] uint8."""
    n = reenacted.shape[0]
    frames = []
    for i in range(n):
        grid = torch.cat([source[0], target[i], reenacted[i]], dim=2)      # [3,H,3W]
        f = tensor_to_uint8_hwc(grid.unsqueeze(0))[0]
        frames.append(f[:, :, ::-1].copy() if swap_rb else f)
    return numpy.stack(frames, 0)
