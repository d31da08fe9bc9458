"""``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` on the gfx950 kernel ``sgdfr_upfirdn2d_f32``.

Semantics and call signature are the reference's (libs/gan/StyleGAN2/op/upfirdn2d.py:149-165; native
entry op/upfirdn2d.cpp:15-26): zero-stuff by ``up``, pad (negative pad crops), convolve with the FIR
``kernel`` (flipped = true convolution), keep every ``down``-th sample; pad is (p0, p1) for both axes
or (x0, x1, y0, y1).

Autograd: the adjoint of an upfirdn is another upfirdn (flipped taps, up and down exchanged, pads chosen
so the result has the input's size -- same arithmetic as upfirdn2d.py:104-117), so the backward simply
re-enters the same differentiable function; any order of derivative works without extra classes.
The reference sends CPU tensors to a PyTorch implementation (upfirdn2d.py:159-160); this package is
GPU-only and raises instead.
"""
from collections import namedtuple

import torch
from torch.autograd import Function

from .. import _native as N

Geometry = namedtuple('Geometry', 'up_x up_y down_x down_y pad_x0 pad_x1 pad_y0 pad_y1')


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def out_size(g, in_h, in_w, kh, kw):
    return ((in_h * g.up_y + g.pad_y0 + g.pad_y1 - kh + g.down_y) // g.down_y,
            (in_w * g.up_x + g.pad_x0 + g.pad_x1 - kw + g.down_x) // g.down_x)


def upfirdn2d_native_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Call-compatible with the reference's native ``upfirdn2d_op.upfirdn2d``: input [major, H, W, minor]."""
    dt = N.native_dtype(input)                    # fp32 | fp16 | fp64 (upfirdn2d_kernel.cu:225); the taps follow the input's dtype
    if kernel.dtype != input.dtype:
        kernel = kernel.to(input.dtype)
    N.require_device(input, kernel, dtype=input.dtype)
    g = Geometry(up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    x, k = N.f32c(input), N.f32c(kernel)
    major, in_h, in_w, minor = x.shape
    oh, ow = out_size(g, in_h, in_w, k.shape[0], k.shape[1])
    if oh <= 0 or ow <= 0:
        raise RuntimeError('upfirdn2d: empty output %dx%d' % (oh, ow))
    y = torch.empty(major, oh, ow, minor, device=x.device, dtype=x.dtype)
    if dt == 0:
        N.call('sgdfr_upfirdn2d_f32', N.ptr(x), N.ptr(k), N.ptr(y), major, in_h, in_w, minor, k.shape[0], k.shape[1],
               *g, N.stream())
    else:
        N.call('sgdfr_upfirdn2d', N.ptr(x), N.ptr(k), N.ptr(y), major, in_h, in_w, minor, k.shape[0], k.shape[1],
               *g, dt, N.stream())
    return y


def _adjoint_geometry(g, in_h, in_w, kh, kw):
    oh, ow = out_size(g, in_h, in_w, kh, kw)
    return Geometry(g.down_x, g.down_y, g.up_x, g.up_y,
                    kw - g.pad_x0 - 1, in_w * g.up_x - ow * g.down_x + g.pad_x0 - g.up_x + 1,
                    kh - g.pad_y0 - 1, in_h * g.up_y - oh * g.down_y + g.pad_y0 - g.up_y + 1)


class _UpFirDn(Function):
    @staticmethod
    def forward(ctx, x, kernel, g):
        b, c, h, w = x.shape
        ctx.save_for_backward(kernel)
        ctx.geom, ctx.in_hw = g, (h, w)
        y = upfirdn2d_native_op(x.reshape(b * c, h, w, 1), kernel, *g)
        return y.view(b, c, y.shape[1], y.shape[2])

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        h, w = ctx.in_hw
        adj = _adjoint_geometry(ctx.geom, h, w, kernel.shape[0], kernel.shape[1])
        gx = _UpFirDn.apply(gy, torch.flip(kernel, [0, 1]), adj)
        assert gx.shape[2:] == (h, w)
        return gx, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    up, down = _pair(up), _pair(down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    N.native_dtype(input)
    N.require_device(input, dtype=input.dtype)
    return _UpFirDn.apply(input, kernel, Geometry(up[0], up[1], down[0], down[1], *pad))
