"""torch.autograd.Function wrappers of the generator path: forward AND backward run on the HIP library.

Gradient algebra: SURVEY.md Appendix C (verified there in fp64 against autograd of the reference's
libs/gan/StyleGAN2/model.py:232-273).  With u = x*s, v = conv(u, Wc), y = d*v:

    dL/dx   = s * gu,          gu = conv^T(d * g)        -- the forward MFMA kernel with the transposed weight pack
                                                            and the roles of s and d exchanged
    dL/ds   = sum_q x*gu  (+ the demodulation term, applied in StyleFn)
    dL/dd   = sum_p g*v = (sum_p g*y) / d
    d       = rsqrt(sum_i s^2 Q + eps)  =>  dL/ds_i += s_i * sum_o (-d^3 dL/dd)[o] * Q[o,i]

Consumers: libs/trainer.py:177-189 (needs dL/dW+ -> A), libs/optimization.py:47-68 (PTI: generator parameters).
Tiny [B,C]-sized glue (divisions, sums over the batch, transposes) uses torch tensor ops, exactly where the
reference itself does (e.g. the bias gradient `grad_input.sum(dim)` of op/fused_act.py:32-37).
"""
import math
import warnings

import torch
from torch.autograd import Function

from . import _native as N
from . import functional as F_

_warned = set()


def _warn_once(key, msg):
    if key not in _warned:
        _warned.add(key)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _t(x):
    return x.t().contiguous()


def _colsum(m):
    """[R, C] -> [C] column sums through the HIP linear kernel (ones @ m)."""
    ones = torch.ones(1, m.shape[0], device=m.device, dtype=torch.float32)
    return F_.linear(ones, _t(m)).view(-1)


# ------------------------------------------------------------------ latent preparation

class LatentPrepareFn(Function):
    """out[b,l] = t + psi*(v - t), v = w[b,(l)] + shift[b,(l)] on the first rows (generic.py:116-135, model.py:494-508)."""

    @staticmethod
    def forward(ctx, w, shift, trunc, n_latent, shift_layers, psi):
        ctx.meta = (w.ndim == 3, None if shift is None else shift.ndim == 3, shift_layers, psi, trunc is not None,
                    None if trunc is None else trunc.shape)
        if shift is not None and shift.ndim == 3:
            ctx.meta = ctx.meta[:2] + (shift.shape[1],) + ctx.meta[3:]
        return F_.latent_prepare(w, n_latent, shift=shift, shift_layers=shift_layers, trunc=trunc, psi=psi)

    @staticmethod
    def backward(ctx, g):
        w_plus, shift_plus, layers, psi, has_trunc, tshape = ctx.meta
        gv = g * psi if has_trunc else g
        gw = gshift = gtrunc = None
        if ctx.needs_input_grad[0]:
            gw = gv if w_plus else gv.sum(1)
        if shift_plus is not None and ctx.needs_input_grad[1]:
            gshift = gv[:, :layers] if shift_plus else gv[:, :layers].sum(1)
        if has_trunc and ctx.needs_input_grad[2]:
            gtrunc = (g.sum((0, 1)) * (1.0 - psi)).view(tshape)
        return gw, gshift, gtrunc, None, None, None


# ------------------------------------------------------------------ dense layers

class EqualLinearFn(Function):
    """y = act(x @ W.T * scale + b * lr_mul)  (model.py:148-157)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, lr_mul, lrelu):
        y = F_.linear(x, weight, bias, wscale=scale, bscale=lr_mul, lrelu=lrelu)
        ctx.save_for_backward(x, weight, y if lrelu else None)
        ctx.cfg = (scale, lr_mul, lrelu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        scale, lr_mul, lrelu, has_bias = ctx.cfg
        g2 = N.f32c(g).reshape(-1, weight.shape[0])
        if lrelu:
            from .op.fused_act import fused_bias_act
            g2 = fused_bias_act(g2, None, y.reshape(-1, weight.shape[0]), 3, 1, 0.2, 2 ** 0.5)
        x2 = x.reshape(-1, weight.shape[1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = F_.linear(g2, _t(weight), wscale=scale).view_as(x)
        if ctx.needs_input_grad[1]:
            gw = F_.linear(_t(g2), _t(x2), wscale=scale)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _colsum(g2) * lr_mul
        return gx, gw, gb, None, None, None


class PixelNormFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return F_.pixel_norm(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors            # y = x*r, r = rsqrt(mean x^2 + eps)  =>  dx = r*g - x * r^3 * mean(g*x)
        x2, g2 = N.f32c(x).reshape(x.shape[0], -1), N.f32c(g).reshape(x.shape[0], -1)
        dx = torch.empty_like(x2)
        N.call('sgdfr_pixelnorm_bwd_f32', N.ptr(x2), N.ptr(g2), N.ptr(dx), x2.shape[0], x2.shape[1], 1e-8, N.stream())
        return dx.view(x.shape)


class StyleFn(Function):
    """s = modulation(style) (model.py:235) and d = rsqrt(sum_i s^2 Q + 1e-8) (model.py:238-239)."""

    @staticmethod
    def forward(ctx, style, mod_w, mod_b, q, qt, cout):
        s, d = F_.style_demod(style, mod_w, mod_b, q, cout)
        ctx.save_for_backward(style, mod_w, s, d, qt)
        ctx.has_d = d is not None
        if d is None:
            return s
        return s, d

    @staticmethod
    def backward(ctx, gs, gd=None):
        style, mod_w, s, d, qt = ctx.saved_tensors
        D = mod_w.shape[1]
        scale = 1.0 / math.sqrt(D)
        if gs is None:
            gs = torch.zeros_like(s)
        ds = F_.demod_grad(gd, d, qt, s, gs) if (ctx.has_d and gd is not None) else N.f32c(gs)
        gstyle = gw = gb = None
        if ctx.needs_input_grad[0]:
            gstyle = F_.linear(ds, _t(mod_w), wscale=scale)
        if ctx.needs_input_grad[1]:
            gw = F_.linear(_t(ds), _t(style), wscale=scale)
        if ctx.needs_input_grad[2]:
            gb = _colsum(ds)
        return gstyle, gw, gb, None, None, None


class StylesBatchedFn(Function):
    """Every layer's s_l = modulation_l(w_l) and d_l of one forward in two launches (functional.styles_batched), for the
    frozen-generator case (the direction trainer optimises A alone, trainer.py:106-111,188): differentiable w.r.t. the
    W+ latent only.  Outputs: s of every entry of `order`, followed by its d when the layer demodulates."""

    @staticmethod
    def forward(ctx, latent, order):
        out = F_.styles_batched(latent, [m.style_spec(li) for m, li in order])
        flat = []
        for s, d in out:
            flat.append(s)
            if d is not None:
                flat.append(d)
        ctx.order = order
        ctx.has_d = [d is not None for _, d in out]
        ctx.lat_shape = latent.shape
        ctx.save_for_backward(*flat)
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grads):
        flat = ctx.saved_tensors
        glat = torch.zeros(ctx.lat_shape, device=flat[0].device, dtype=torch.float32)
        i = 0
        for (m, li), has_d in zip(ctx.order, ctx.has_d):
            s, gs = flat[i], grads[i]
            i += 1
            d = gd = None
            if has_d:
                d, gd = flat[i], grads[i]
                i += 1
            if gs is None and gd is None:
                continue
            if gs is None:
                gs = torch.zeros_like(s)
            ds = F_.demod_grad(gd, d, m.packed()[2], s, gs) if gd is not None else N.f32c(gs)
            mod_w = m.modulation.weight
            glat[:, li] += F_.linear(ds, _t(mod_w), wscale=1.0 / math.sqrt(mod_w.shape[1]))
        return glat, None


# ------------------------------------------------------------------ modulated convs


def _conv_input_grad(mod, g_pre, g_max, d, planes, shape, bw_arith, want_planes_grad):
    """dL/d(x*s) of one modulated 3x3 conv from the gradient g_pre of its (pre-activation) output -- the MFMA part of a layer's
    backward, shared by the per-layer Function and the frozen generator's whole-synthesis Function.
    Returns (gu [B,Cin,H,W], A = sum gT*T [B,Cout] for the transposed conv (else None), gT = the fp32 plane gradient when
    want_planes_grad (the weight gradient reads it) and the conv is a transposed one, else None)."""
    B, cin, cout, H, W = shape
    up = mod.upsample
    dev = g_pre.device
    ones_d = d if d is not None else F_.ones_like_rows(B, cout, dev)
    # Arithmetic of the dL/dx convs on the split kernels.  'bf16x3': 8+8-bit terms, fp32 range -- needs no scale.
    # 'fp16x3' (functional.BACKWARD_ARITH): 11+11-bit terms like the forward, made usable for gradients (which have no
    # natural scale) by the same exact power-of-two plan as the forward: e from the true max |g| of each image, d * 2^e
    # on the way in, 2^-e on the way out.
    d_in, d_out = ones_d, None
    if bw_arith == 'fp16x3':
        # (max |g_pre| per image comes out of the activation-gradient pass.  The plane gradient of the transposed conv is the
        # adjoint of the 4x4 blur FIR, whose taps are scaled by factor^2 and sum to 4 (model.py:78-79): |gT| <= 4 max |g_pre|,
        # two binades of headroom; with the true maximum and that headroom finite gradients cannot saturate)
        d_in, d_out = F_.split_range(ones_d, F_.ones_like_rows(B, cin, dev), g_max, headroom=2 if up else 0)
    A = gT = None
    if up:
        split_down = F_.config().precision != 'fp32' and F_.split_ok(B, cout, cin, H, W, N.MODE_DOWN3)
        if split_down and not want_planes_grad:
            # frozen weights: only the conv below reads the plane gradient -> the blur adjoint writes it directly in the
            # conv's split input form (no fp32 planes, no conversion pass)
            gxs, A = F_.blur_adjoint_split(g_pre, mod.blur.kernel, planes if d is not None else None,
                                           d_in if (d is not None or d_out is not None) else None, bw_arith)
        else:
            gT, A = F_.blur_adjoint(g_pre, mod.blur.kernel, planes if d is not None else None)
            gxs = F_.planes_to_split(gT, d_in if (d is not None or d_out is not None) else None, bw_arith) if split_down else None
        if split_down:
            # dL/d(x*s) of the transposed conv on the split kernels too (bw_arith terms, see above):
            # the planes times d come in the phase-major split form, the conv walks (channel block, phase) pairs
            gu = F_.modconv_split(gxs, mod.packed_split(adjoint='down', arith=bw_arith), None, d_out, cin, mode=N.MODE_DOWN3,
                                  arith=bw_arith, x_split=(B, cout, H, W), batch=B,
                                  desc='bwd split down3 %d->%d @%dx%d' % (cout, cin, H, W))
        else:
            gu = F_.modconv_raw(gT, mod.packed_t(), ones_d, None, cin, N.MODE_DOWN3, H, W,
                                desc='bwd down3 %d->%d @%dx%d' % (cout, cin, H, W))
    else:
        if F_.split_ok(B, cout, cin, H, W):    # dL/dx of a plain conv is a plain conv: same kernels, adjoint packs.
            # Gradients have no natural scale (1e-8 is as likely as 1e+3): the fp16 terms are planned from max |g_pre| of each
            # image (d_in / d_out above), bf16 terms (fp32 range, 2^-17 per product) need no plan.
            gu = F_.modconv_split(g_pre, mod.packed_split(adjoint=True, arith=bw_arith), d_in, d_out, cin,
                                  desc='bwd split3 %d->%d @%dx%d' % (cout, cin, H, W), arith=bw_arith)
        elif F_.wino_ok(B, cout, cin, H, W):
            gu = F_.modconv_wino(g_pre, mod.packed_wino(adjoint=True), ones_d, None, cin,
                                 desc='bwd wino3 %d->%d @%dx%d' % (cout, cin, H, W))
        else:
            gu = F_.modconv_raw(g_pre, mod.packed_t(), ones_d, None, cin, N.MODE_PLAIN3, H, W,
                                desc='bwd plain3 %d->%d @%dx%d' % (cout, cin, H, W))
    return gu, A, gT


class StyledConvFn(Function):
    """act(d * conv(x*s, Wc) + noise_w*noise + bias)  -- StyledConv.forward (model.py:331-337), plain or upsampling.
    `mod` is the owning ModulatedConv2d (packed weights, FIR taps, shapes); it is not a tensor input."""

    @staticmethod
    def forward(ctx, x, s, d, weight, noise_w, bias, noise, mod, activate, batch):
        wp = mod.packed()[0]
        up = mod.upsample
        planes = None
        if up:
            out, planes = F_.modconv3x3(x, wp, s, d, mod.out_channel, upsample=True, fir=mod.blur.kernel, noise=noise,
                                        noise_weight=noise_w, bias=bias, activate=activate, batch=batch,
                                        return_planes=True, split=mod.packed_split)
        else:
            out = F_.modconv3x3(x, wp, s, d, mod.out_channel, noise=noise, noise_weight=noise_w, bias=bias,
                                activate=activate, batch=batch, wino=mod.packed_wino, split=mod.packed_split)
        ctx.save_for_backward(x, s, d, out, noise_w, bias, noise, planes)
        ctx.mod, ctx.activate = mod, activate
        ctx.sat = F_.current_sink()         # the owning generator's saturation word: the backward's launches count there too
        ctx.cfg = F_.config()               # ... and run under the configuration of the forward (its generator's), not the caller's
        return out

    @staticmethod
    def backward(ctx, g):
        with F_.using(ctx.cfg), F_.saturation_sink(ctx.sat):
            return StyledConvFn._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        x, s, d, out, noise_w, bias, noise, planes = ctx.saved_tensors
        mod, up = ctx.mod, ctx.mod.upsample
        B, cout = out.shape[0], out.shape[1]
        cin, H, W = x.shape[1], x.shape[2], x.shape[3]
        slope, gain = (0.2, 2 ** 0.5) if ctx.activate else (1.0, 1.0)
        bw_arith = F_.config().backward_arith if F_.config().precision != 'fp32' else 'bf16x3'
        if bw_arith == 'fp16x3':
            g_pre, sums, g_max = F_.act_grad_reduce(g, out, noise, noise_w, bias, want_y=(not up) and d is not None,
                                                    slope=slope, gain=gain, want_absmax=True)
        else:
            (g_pre, sums), g_max = F_.act_grad_reduce(g, out, noise, noise_w, bias, want_y=(not up) and d is not None,
                                                      slope=slope, gain=gain), None
        gu, A_up, gT = _conv_input_grad(mod, g_pre, g_max, d, planes, (B, cin, cout, H, W), bw_arith, ctx.needs_input_grad[3])
        A = A_up if up else (sums[:, :, 2] if d is not None else None)
        dx, r = F_.scale_reduce(gu, x, s)
        if x.shape[0] == 1 and B != 1:            # broadcast ConstantInput: gradient sums over the batch
            dx = dx.sum(0, keepdim=True)
        gx = dx if ctx.needs_input_grad[0] else None
        gs = r if ctx.needs_input_grad[1] else None
        gd = (A / d) if (d is not None and ctx.needs_input_grad[2]) else None
        gweight = None
        if ctx.needs_input_grad[3]:
            dq = None
            if d is not None:      # dL/dQ[o,i] = sum_b dL/dd * (-d^3/2) * s^2   (d = rsqrt(sum_i s^2 Q + eps))
                coeff = (A / d) * d.pow(3) * -0.5
                dq = F_.linear(_t(coeff), _t(s * s))
            gweight = F_.wgrad(gT if up else g_pre, d, x, s, cout, up, wp=mod.packed()[0], dq=dq, weight=mod.weight)
        gnw = sums[:, :, 1].sum().view(1) if (noise_w is not None and ctx.needs_input_grad[4]) else None
        gb = sums[:, :, 0].sum(0) if (bias is not None and ctx.needs_input_grad[5]) else None
        return gx, gs, gd, gweight, gnw, gb, None, None, None, None


class ToRGBFn(Function):
    """ToRGB.forward (model.py:350-359): 1x1 modconv without demodulation + bias + FIR-upsampled skip."""

    @staticmethod
    def forward(ctx, x, s, weight, bias, skip, fir):
        cin = x.shape[1]
        y = F_.torgb(x, weight.view(3, cin), s, bias=None if bias is None else bias.view(3), skip=skip, fir=fir)
        ctx.save_for_backward(x, s, weight, fir)
        ctx.has_skip, ctx.has_bias = skip is not None, bias is not None
        ctx.bias_shape = None if bias is None else bias.shape
        return y

    @staticmethod
    def backward(ctx, g):
        x, s, weight, fir = ctx.saved_tensors
        B, cin = x.shape[0], x.shape[1]
        w = weight.view(3, cin)
        scale = 1.0 / math.sqrt(cin)
        g = N.f32c(g)
        dx, r = F_.torgb_bwd(x, g, w, s)                       # r[b,j,i] = sum_p x*g_j
        gx = dx if ctx.needs_input_grad[0] else None
        gs = (r * w.unsqueeze(0)).sum(1) * scale if ctx.needs_input_grad[1] else None
        gw = ((r * s.unsqueeze(1)).sum(0) * scale).view_as(weight) if ctx.needs_input_grad[2] else None
        gb = g.sum((0, 2, 3)).view(ctx.bias_shape) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        gskip = None
        if ctx.has_skip and ctx.needs_input_grad[4]:
            from .op.upfirdn2d import upfirdn2d_native_op
            H, W = g.shape[2], g.shape[3]
            # adjoint of upfirdn2d(up=2, pad=(2,1)): flipped taps, down=2, pad (1,1)  (op/upfirdn2d.py:104-117)
            gskip = upfirdn2d_native_op(g.reshape(B * 3, H, W, 1), torch.flip(fir, [0, 1]), 1, 1, 2, 2, 1, 1, 1, 1)
            gskip = gskip.view(B, 3, H // 2, W // 2)
        return gx, gs, gw, gb, gskip, None


# ------------------------------------------------------------------ the whole synthesis network as one Function

_flip_cache = {}


def _flipped(fir):
    """torch.flip(fir, [0, 1]) of a FIR buffer, cached per (storage, version): six flips per backward otherwise."""
    key = (fir.data_ptr(), fir._version, fir.device)
    t = _flip_cache.get(key)
    if t is None:
        if len(_flip_cache) > 32:
            _flip_cache.clear()
        t = _flip_cache[key] = torch.flip(fir, [0, 1]).contiguous()
    return t


def synthesis_params(gen, layers, to_rgbs):
    """The parameter tensors SynthesisFn takes (and returns gradients for), in its order."""
    ps = [gen.input.input]
    for l in layers:
        ps += [l.conv.weight, l.conv.modulation.weight, l.conv.modulation.bias, l.noise.weight, l.activate.bias]
    for r in to_rgbs:
        ps += [r.conv.weight, r.conv.modulation.weight, r.conv.modulation.bias, r.bias]
    return ps


class SynthesisFn(Function):
    """image = synthesis(latent) -- conv1 ... convs / to_rgbs (model.py:519-534) -- as ONE Function, differentiable w.r.t. the W+
    latent (the direction trainer, libs/trainer.py:177-189: the generator is frozen) and w.r.t. every generator parameter
    (PTI, libs/optimization.py:47-68).

    Forward = the launches of StyledConvFn / ToRGBFn layer by layer, all styles in two launches.  Backward = the same MFMA
    launches (`_conv_input_grad`, `functional.wgrad`), but every saved activation is walked ONCE: the per-layer Functions pay, per
    layer, scale_reduce (dx = s*gu, r = sum x*gu), torgb_bwd (dx_rgb, r_rgb), autograd's add of the two dx and act_grad_reduce of
    the layer below -- four HBM-bound passes over the same tensor, 36 bytes per element -- which `functional.grad_join` does in one
    (12-16 bytes per element).  The 20 style modulations' backward (A/d, demod_grad, a transposed copy of the modulation weight, a
    linear and an indexed add PER LAYER, plus two more linears and a column sum when the modulation is trained) is two or three
    launches (`functional.styles_batched_bwd`), dL/dQ of a layer one (`functional.demod_dq`), all bias / noise-strength / ToRGB
    parameter gradients one (`functional.param_grads`), and all reduction buffers of a backward come from one zeroed workspace:
    ~700 -> ~200 device launches for a PTI step.  Same expressions per term as the per-layer path
    (tests/test_gpu_backward.py compares the two)."""

    N_FIXED = 6          # latent, gen, order, layers, to_rgbs, noise -- then synthesis_params(...)

    @staticmethod
    def forward(ctx, latent, gen, order, layers, to_rgbs, noise, *params):
        B = latent.shape[0]
        sd = F_.styles_batched(latent, [m.style_spec(li) for m, li in order])
        sd_of_layer = [0] + [2 + 3 * (i // 2) + (i % 2) for i in range(len(layers) - 1)]
        sd_of_rgb = [1] + [4 + 3 * k for k in range(len(to_rgbs) - 1)]
        x, skip = gen.input.input, None
        saved, noises = [], []
        # max |x| per (image, channel) plane of every plain layer's input comes out of the blur that produced it (one zeroed buffer
        # for all of them) instead of a separate pass over the largest activations of the forward
        want_max = F_.config().precision == 'fp16x3' and bool(F_.config().range_plan)
        ups = [l.conv for l in layers if l.conv.upsample]
        words = torch.zeros(B * sum(m.out_channel for m in ups), device=latent.device, dtype=torch.int32) if (want_max and ups) else None
        woff, x_max = 0, None
        for li, layer in enumerate(layers):
            mod = layer.conv
            s, d = sd[sd_of_layer[li]]
            nz = noise[li]
            if nz is None:          # fresh per-sample noise, model.py:283-285
                r = x.shape[-1] * (2 if mod.upsample else 1)
                nz = torch.empty(B, 1, r, r, device=latent.device, dtype=torch.float32).normal_()
            planes = None
            if mod.upsample:
                out_max = None
                if words is not None:
                    out_max = words[woff:woff + B * mod.out_channel].view(B, mod.out_channel)
                    woff += B * mod.out_channel
                out, planes = F_.modconv3x3(x, mod.packed()[0], s, d, mod.out_channel, upsample=True, fir=mod.blur.kernel, noise=nz,
                                            noise_weight=layer.noise.weight, bias=layer.activate.bias, activate=True,
                                            batch=B if li == 0 else None, return_planes=True, split=mod.packed_split, absmax_out=out_max)
                x_max = out_max
            else:
                k = li // 2
                rgb = to_rgbs[k]
                H = x.shape[2]
                # the ToRGB that reads this layer is accumulated in the conv's own epilogue where the split kernel runs the layer in
                # one pass (as on the no-grad path: per-cout-tile partial sums + one small finish launch instead of a pass that
                # re-reads the activation)
                fuse = F_.config().precision != 'fp32' and F_.rgb_fusable(B, mod.in_channel, mod.out_channel, H, H)
                res = F_.modconv3x3(x, mod.packed()[0], s, d, mod.out_channel, noise=nz, noise_weight=layer.noise.weight,
                                    bias=layer.activate.bias, activate=True, batch=B if li == 0 else None, wino=mod.packed_wino,
                                    split=mod.packed_split,
                                    rgb=(rgb.conv.weight.view(3, mod.out_channel), sd[sd_of_rgb[k]][0]) if fuse else None, x_absmax=x_max)
                x_max = None
                if fuse:
                    out, part = res
                    skip = rgb.finish(part, skip)
                else:
                    out = res
                    fir = None
                    if skip is not None:
                        up = getattr(rgb, 'upsample', None)
                        if up is None or tuple(up.kernel.shape) != (4, 4) or up.pad != (2, 1):
                            raise NotImplementedError('ToRGB skip path is built for the 4-tap 2x Upsample')
                        fir = up.kernel
                    skip = F_.torgb(out, rgb.conv.weight.view(3, mod.out_channel), sd[sd_of_rgb[k]][0], bias=rgb.bias.view(3), skip=skip,
                                    fir=fir)
            saved.append((out, planes))
            noises.append(nz)
            x = out
        ctx.gen, ctx.order, ctx.layers, ctx.to_rgbs = gen, order, layers, to_rgbs
        ctx.sd, ctx.saved, ctx.noises = sd, saved, noises
        ctx.latent = latent if any(ctx.needs_input_grad[SynthesisFn.N_FIXED:]) else None     # (the modulation weight gradients read it)
        ctx.lat_shape = tuple(latent.shape)
        ctx.sat = F_.current_sink()
        ctx.cfg = F_.config()
        return skip

    @staticmethod
    def backward(ctx, g_img):
        with F_.using(ctx.cfg), F_.saturation_sink(ctx.sat):
            return SynthesisFn._backward(ctx, g_img)

    @staticmethod
    def _backward(ctx, g_img):
        from .op.upfirdn2d import upfirdn2d_native_op
        gen, order, layers, to_rgbs, sd = ctx.gen, ctx.order, ctx.layers, ctx.to_rgbs, ctx.sd
        B, L, D = ctx.lat_shape
        n, nr = len(layers), len(to_rgbs)
        F0 = SynthesisFn.N_FIXED
        need = ctx.needs_input_grad
        grads = [None] * len(need)
        lay_p = lambda li, j: F0 + 1 + 5 * li + j                   # weight, modulation.weight, modulation.bias, noise.weight, activate.bias
        rgb_p = lambda k, j: F0 + 1 + 5 * n + 4 * k + j             # weight, modulation.weight, modulation.bias, bias
        sd_of_layer = [0] + [2 + 3 * (i // 2) + (i % 2) for i in range(n - 1)]
        sd_of_rgb = [1] + [4 + 3 * k for k in range(nr - 1)]
        dev = g_img.device
        bw_arith = F_.config().backward_arith if F_.config().precision != 'fp32' else 'bf16x3'
        # gradient of every level's RGB image: the skip path is Upsample (upfirdn2d up=2, pad (2,1)); its adjoint = flipped taps,
        # down=2, pad (1,1)  (op/upfirdn2d.py:104-117) -- 3-channel tensors
        g_rgb = [None] * nr
        g_rgb[-1] = N.f32c(g_img)
        for k in range(nr - 1, 0, -1):
            g = g_rgb[k]
            H, W = g.shape[2], g.shape[3]
            g_rgb[k - 1] = upfirdn2d_native_op(g.reshape(B * 3, H, W, 1), _flipped(to_rgbs[k].upsample.kernel), 1, 1, 2, 2, 1, 1, 1, 1).view(B, 3, H // 2, W // 2)
        # one zeroed workspace for every reduction of this backward (sums, max |g|, r, r_rgb: 8 words per (image, channel) plane)
        work = torch.zeros(8 * B * sum(l.conv.out_channel for l in layers), device=dev, dtype=torch.float32)
        woff = 0
        gs = [None] * n            # dL/ds of layer li's modulation (sum_q x * gu)
        A = [None] * n             # d * dL/dd
        rgb_r = [None] * nr
        small = []                 # (index into grads, shape, functional.param_grads entry)
        gu_next = None
        for li in range(n - 1, -1, -1):
            layer, mod = layers[li], layers[li].conv
            out, planes = ctx.saved[li]
            s, d = sd[sd_of_layer[li]]
            C = mod.out_channel
            cin = mod.in_channel                       # (H, W: of the conv's INPUT)
            H, W = (out.shape[2] // 2, out.shape[3] // 2) if mod.upsample else (out.shape[2], out.shape[3])
            k = li // 2
            has_rgb = not mod.upsample
            w = work[woff:woff + 8 * B * C]
            woff += 8 * B * C
            g_pre, sums, g_max, r_next, r_rgb = F_.grad_join(
                out, gu=gu_next, s_next=sd[sd_of_layer[li + 1]][0] if gu_next is not None else None,
                g_rgb=g_rgb[k] if has_rgb else None, w_rgb=to_rgbs[k].conv.weight.view(3, C) if has_rgb else None,
                s_rgb=sd[sd_of_rgb[k]][0] if has_rgb else None, noise=ctx.noises[li], noise_weight=layer.noise.weight,
                bias=layer.activate.bias, want_y=(not mod.upsample) and d is not None, work=w)
            if gu_next is not None:
                gs[li + 1] = r_next
            if has_rgb:
                rgb_r[k] = r_rgb
            want_w = need[lay_p(li, 0)]
            gu_next, A_up, gT = _conv_input_grad(mod, g_pre, g_max if bw_arith == 'fp16x3' else None, d, planes,
                                                 (B, cin, C, H, W), bw_arith, want_w)
            A[li] = A_up if mod.upsample else (sums[:, :, 2] if d is not None else None)
            if want_w:
                x = ctx.saved[li - 1][0] if li > 0 else gen.input.input
                grads[lay_p(li, 0)] = F_.wgrad(gT if mod.upsample else g_pre, d, x, s, C, mod.upsample, wp=mod.packed()[0],
                                               a=A[li] if d is not None else None, weight=mod.weight)
            if need[lay_p(li, 3)]:
                small.append((lay_p(li, 3), (1,), (N.PGRAD_NOISE, sums, None, C, 0)))
            if need[lay_p(li, 4)]:
                small.append((lay_p(li, 4), tuple(layer.activate.bias.shape), (N.PGRAD_BIAS, sums, None, C, 0)))
        dx0, gs[0] = F_.scale_reduce(gu_next, gen.input.input, sd[sd_of_layer[0]][0])        # conv1 reads the broadcast ConstantInput
        if need[F0]:
            grads[F0] = dx0.sum(0, keepdim=True)
        for k, rgb in enumerate(to_rgbs):
            C = rgb.conv.in_channel
            if need[rgb_p(k, 0)]:
                small.append((rgb_p(k, 0), tuple(rgb.conv.weight.shape), (N.PGRAD_RGB_W, rgb_r[k], sd[sd_of_rgb[k]][0], C, 0)))
            if need[rgb_p(k, 3)]:
                g = g_rgb[k]
                small.append((rgb_p(k, 3), tuple(rgb.bias.shape), (N.PGRAD_RGB_B, g, None, 3, g.shape[2] * g.shape[3])))
        if small:
            for (gi, shape, _), o in zip(small, F_.param_grads([e for _, _, e in small], B)):
                grads[gi] = o.view(shape)
        layer_of = {id(l.conv): i for i, l in enumerate(layers)}
        rgb_of = {id(r.conv): i for i, r in enumerate(to_rgbs)}
        entries, slots = [], []
        for (m, lat_i), (s, d) in zip(order, sd):
            e = {'latent_index': lat_i, 'mod_w': m.modulation.weight}
            if id(m) in rgb_of:
                k = rgb_of[id(m)]
                e['rgb_r'], e['rgb_w'] = rgb_r[k], m.weight.view(3, m.in_channel)
                slot = (rgb_p(k, 1), rgb_p(k, 2))
            else:
                li = layer_of[id(m)]
                e['gs'] = gs[li]
                if d is not None:
                    e['a'], e['d'], e['s'], e['qt'] = A[li], d, s, m.packed()[2]
                slot = (lay_p(li, 1), lay_p(li, 2))
            e['want_w'], e['want_b'] = need[slot[0]], need[slot[1]]
            entries.append(e)
            slots.append(slot)
        grads[0] = F_.styles_batched_bwd(entries, B, L, D, latent=ctx.latent, want_latent=need[0])
        for e, (iw, ib) in zip(entries, slots):
            if e.get('gmod_w') is not None:
                grads[iw] = e['gmod_w']
            if e.get('gmod_b') is not None:
                grads[ib] = e['gmod_b']
        return tuple(grads)
