"""Host wrappers of the shelved kernels (ctypes over experiments/libsgdfr_experiments.so; declarations in
experiments/sgdfr_experiments.h).  Build first: `python experiments/build.py`."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from stylegan_directions_face_reenactment_amd import _native as N, functional as F_      # noqa: E402

LIB = os.path.join(HERE, 'libsgdfr_experiments.so')
_i, _i64, _f, _p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
SIGNATURES = {
    'sgdfr_modconv2d_upfir_supported': [_i] * 5,
    'sgdfr_modconv2d_upfir_tiles': [_i] * 5 + [ctypes.POINTER(ctypes.c_int)] * 2,
    'sgdfr_modconv2d_upfir_split_f32': [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p],
    'sgdfr_modconv2d_up_pp_supported': [_i] * 5 + [_i64],
    'sgdfr_modconv2d_up_pp_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i64, _i, _p],
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError('%s is missing: run `python experiments/build.py`' % LIB)
        N.load()
        _lib = ctypes.CDLL(LIB)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, ctypes.c_int
    return _lib


def _call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, N.load().sgdfr_last_error().decode()))


def upfir_ok(B, cin, cout, H, W):
    return F_.config().precision in F_._SPLIT_ARITH and bool(load().sgdfr_modconv2d_upfir_supported(B, cin, cout, H, W))


def up_pp_ok(B, cin, cout, H, W, plane_stride):
    return bool(load().sgdfr_modconv2d_up_pp_supported(B, cin, cout, H, W, int(plane_stride)))


def modconv_upfir_split(xs_in, shape, wsp, d, cout, fir, s_next, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2,
                        gain=F_.SQRT2, arith=None):
    """Upsampling StyledConv in one launch: xs_in = the split form of x*s (shape = (B, Cin, H, W)) -> the split form of
    act(blur(conv_transpose(x*s) * d) + noise + bias) * s_next, [B, cout/8, 2, 2H*2W, 8] int16 (same bits as
    modconv_split(mode=UP3) + blur_bias_act_split)."""
    arith = F_._SPLIT_ARITH[arith or F_.config().precision]
    N.require_device(d, fir, bias, noise_weight, s_next)
    B, cin, H, W = shape
    nz, nzb = F_._noise_args(noise, B, 2 * H, 2 * W)
    xs = torch.empty(B, cout // 8, 2, 4 * H * W, 8, device=xs_in.device, dtype=torch.int16)
    _call('sgdfr_modconv2d_upfir_split_f32', N.ptr(xs_in), N.ptr(wsp), N.ptr(N.f32c(d)), N.ptr(N.f32c(fir)), N.ptr(nz), nzb,
          N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(N.f32c(s_next)), N.ptr(F_._zero_words(xs_in.device)),
          N.ptr(xs), B, cin, cout, H, W, arith, int(activate), float(slope), float(gain), F_._sat(), N.stream())
    return xs


def modconv_up_pp(xs_in, shape, wsp, d, cout, plane_stride, arith=None):
    """The transposed conv on the role-swapping kernel: interleaved planes [B, cout, 4, plane_stride] (same bits as
    modconv_split(mode=UP3, x_split=shape, plane_stride=...))."""
    arith = F_._SPLIT_ARITH[arith or F_.config().precision]
    B, cin, H, W = shape
    y = torch.empty(B, cout, 4, int(plane_stride), device=xs_in.device, dtype=torch.float32)
    _call('sgdfr_modconv2d_up_pp_f32', N.ptr(xs_in), N.ptr(wsp), N.ptr(N.f32c(d)), N.ptr(F_._zero_words(xs_in.device)), N.ptr(y),
          B, cin, cout, H, W, int(plane_stride), arith, N.stream())
    return y
