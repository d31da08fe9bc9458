"""``fused_leaky_relu`` / ``FusedLeakyReLU``: y = sqrt(2) * lrelu_0.2(x + bias[c]) on the gfx950 kernel
``sgdfr_fused_bias_act_f32``.

API contract kept from the reference (libs/gan/StyleGAN2/op/fused_act.py:73-86: names, argument order,
defaults, ``.bias``/``.negative_slope``/``.scale`` attributes; native call convention of
op/fused_bias_act.cpp:14-24 where an empty tensor means "absent").  The autograd design is this
package's own: one differentiable primitive ``_gate`` (y = g * slope_mask(ref) * scale) whose derivative
is itself, so first- and second-order gradients come from the same kernel.
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _native as N


def _bias_geometry(x):
    inner = 1
    for d in x.shape[2:]:
        inner *= d
    return inner, (x.shape[1] if x.ndim > 1 else 1)


def _launch(x, bias, ref, grad_order, slope, scale):
    """fp32, fp16 or fp64 (the dtypes of the reference's dispatch, fused_bias_act_kernel.cu:79); bias / ref follow x's dtype."""
    dt = N.native_dtype(x)
    if bias is not None and bias.dtype != x.dtype:        # (a module's fp32 bias Parameter under a half / double input)
        bias = bias.to(x.dtype)
    if ref is not None and ref.dtype != x.dtype:
        ref = ref.to(x.dtype)
    N.require_device(x, bias, ref, dtype=x.dtype)
    x = N.f32c(x)
    bias = N.f32c(bias) if bias is not None else None
    ref = N.f32c(ref) if ref is not None else None
    inner, channels = _bias_geometry(x)
    if bias is not None and bias.numel() != channels:
        raise RuntimeError('bias has %d elements but input has %d channels' % (bias.numel(), channels))
    y = torch.empty_like(x)
    if dt == 0:
        N.call('sgdfr_fused_bias_act_f32', N.ptr(x), N.ptr(bias), N.ptr(ref), N.ptr(y), x.numel(), inner, channels,
               3, grad_order, float(slope), float(scale), N.stream())
    else:
        N.call('sgdfr_fused_bias_act', N.ptr(x), N.ptr(bias), N.ptr(ref), N.ptr(y), x.numel(), inner, channels,
               3, grad_order, float(slope), float(scale), dt, N.stream())
    return y


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Call-compatible with the reference's native ``fused.fused_bias_act`` (fused_bias_act.cpp:14-24)."""
    if act != 3:
        raise RuntimeError('only act=3 (leaky ReLU) exists on this path')
    bias = bias if bias is not None and bias.numel() else None
    refer = refer if refer is not None and refer.numel() else None
    return _launch(input, bias, refer, int(grad), alpha, scale)


def _sum_to_channels(t):
    dims = [0] + list(range(2, t.ndim))
    return t.sum(dims)


class _Gate(Function):
    """g -> g * (ref > 0 ? 1 : slope) * scale  (linear in g, so it is its own derivative)."""

    @staticmethod
    def forward(ctx, g, ref, slope, scale):
        ctx.save_for_backward(ref)
        ctx.cfg = (slope, scale)
        return _launch(g, None, ref, 1, slope, scale)

    @staticmethod
    def backward(ctx, gg):
        ref, = ctx.saved_tensors
        return _Gate.apply(gg, ref, *ctx.cfg), None, None, None


class _BiasLeakyReLU(Function):
    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        y = _launch(x, bias, None, 0, slope, scale)
        ctx.save_for_backward(y)          # the sign of y equals the sign of x + bias
        ctx.cfg = (slope, scale)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gx = _Gate.apply(gy, y, *ctx.cfg)
        gb = _sum_to_channels(gx) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    if bias is not None and bias.numel() == 0:
        bias = None
    return _BiasLeakyReLU.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
