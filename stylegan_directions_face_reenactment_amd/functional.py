"""Functional host layer over the C ABI (include/sgdfr.h): allocation, shape checks, stream.

Each function is one (or a fixed short sequence of) native launch(es) on torch's current HIP stream;
nothing here computes with torch ops.  Reference lines each function stands for are cited inline
(paths relative to the reference's libs/gan/StyleGAN2/).

The package's host layer is four modules; this one is the launch wrappers and the facade the rest of the package (and the
tests) import as `F_`:
    config.py    the frozen Config, `config()`, `using`, `precision`, `set_default`        (re-exported here)
    timing.py    per-launch HIP-event timing for bench.py: `timing.collect()`             (used here)
    chain.py     the split-chain dataflow: SplitAct, styled_conv_split, wsplit_chain_f,
                 rgb_fusable, xin_ok, StreamPipeline                                       (served lazily as F_.<name>)
"""
import functools
import math
import sys
import threading
import types

import torch
from torch.autograd import Function

from . import _native as N
from . import config as _config
from .config import Config, config, precision, set_default, set_precision, using      # noqa: F401  (the facade's exports)
from . import timing
from .timing import timed_conv as _timed_conv, timed_hbm as _timed_hbm

SQRT2 = 2 ** 0.5

_CHAIN_NAMES = ('SplitAct', 'xin_ok', 'styled_conv_split', 'wsplit_chain_f', 'wsplit_chain_arith', 'xs_chain_arith', 'xs_plain_arith', 'wsplit_chain_ok', 'rgb_fusable', 'StreamPipeline')


class _Facade(types.ModuleType):
    """functional's module class: F_.DEFAULT and the old switch names (F_.PRECISION, F_.USE_WSPLIT ...) are live views of
    config.py; the chain layer's names come from chain.py on first use (it imports this module, so not at import time); the old
    switch names cannot be assigned (an assignment would shadow the view while every launch ignores it)."""

    def __getattr__(self, name):
        if name in _config.LEGACY:
            return getattr(config(), _config.LEGACY[name])
        if name == 'DEFAULT':
            return _config.DEFAULT
        if name in _CHAIN_NAMES:
            from . import chain
            return getattr(chain, name)
        raise AttributeError('module %r has no attribute %r' % (self.__name__, name))

    def __setattr__(self, name, value):
        if name in _config.LEGACY or name == 'DEFAULT':
            raise AttributeError('functional.%s is a read-only view of the ambient Config: use `with functional.using(functional.'
                                 'config().replace(%s=...))`, functional.set_default(...), or a Generator\'s `config`'
                                 % (name, _config.LEGACY.get(name, 'field')))
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _Facade

# ------------------------------------------------------------------ small dense ops

def pixel_norm(x, eps=1e-8):
    """model.py:11-16."""
    N.require_device(x)
    x2 = N.f32c(x).reshape(x.shape[0], -1) if x.ndim != 2 else N.f32c(x)
    y = torch.empty_like(x2)
    N.call('sgdfr_pixelnorm_f32', N.ptr(x2), N.ptr(y), x2.shape[0], x2.shape[1], float(eps), N.stream())
    return y.view(x.shape)


def linear(x, weight, bias=None, wscale=1.0, bscale=1.0, lrelu=False, slope=0.2, gain=SQRT2):
    """act((x @ weight.T) * wscale + bias * bscale): EqualLinear (model.py:148-157) and nn.Linear
    (libs/models/direction_matrix.py:44).  x [..., K] (last dim contiguous), weight [N, K]."""
    N.require_device(x, weight, bias)
    K = weight.shape[1]
    if x.shape[-1] != K:
        raise RuntimeError('linear: input has %d features, weight expects %d' % (x.shape[-1], K))
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < K):
        x2 = x2.contiguous()
    ldx = x2.stride(0) if x2.shape[0] > 1 else K
    w = N.f32c(weight)
    b = N.f32c(bias) if bias is not None else None
    M, Nn = x2.shape[0], w.shape[0]
    y = torch.empty(M, Nn, device=x.device, dtype=torch.float32)
    N.call('sgdfr_linear_f32', N.ptr(x2), ldx, N.ptr(w), N.ptr(b), N.ptr(y), Nn, M, Nn, K, float(wscale),
           float(bscale), N.ACT_LRELU if lrelu else N.ACT_NONE, float(slope), float(gain), N.stream())
    return y.view(*x.shape[:-1], Nn)


def latent_prepare(w, n_latent, shift=None, shift_layers=0, trunc=None, psi=1.0):
    """W / W+ -> [B, n_latent, D] with optional direction shift and truncation in one pass
    (libs/utilities/generic.py:116-135 followed by model.py:494-508)."""
    N.require_device(w, shift, trunc)
    w = N.f32c(w)
    B, D = w.shape[0], w.shape[-1]
    w_plus = w.ndim == 3
    if w_plus and w.shape[1] != n_latent:
        raise RuntimeError('W+ code has %d rows, generator needs %d' % (w.shape[1], n_latent))
    shift_plus = 0
    if shift is not None:
        shift = N.f32c(shift)
        shift_plus = int(shift.ndim == 3)
        if shift_plus:
            shift_layers = shift.shape[1]
        if shift.shape[0] != B or shift.shape[-1] != D:
            raise RuntimeError('shift shape %s does not match latent %s' % (tuple(shift.shape), tuple(w.shape)))
    if trunc is not None:
        trunc = N.f32c(trunc).reshape(-1)
        if trunc.numel() != D:
            raise RuntimeError('truncation latent must have %d elements' % D)
    out = torch.empty(B, n_latent, D, device=w.device, dtype=torch.float32)
    N.call('sgdfr_latent_prepare_f32', N.ptr(w), int(w_plus), N.ptr(shift), shift_plus, int(shift_layers),
           N.ptr(trunc), float(psi), N.ptr(out), B, n_latent, D, N.stream())
    return out


# ------------------------------------------------------------------ modulated conv

def prepack(weight):
    """weight [1, Cout, Cin, k, k] (model.py:218-220) -> (wp [Cin, k*k, Cout] scaled, q [Cout, Cin], qt = q^T)."""
    N.require_device(weight)
    w = N.f32c(weight)
    _, cout, cin, k, _ = w.shape
    wp = torch.empty(cin, k * k, cout, device=w.device, dtype=torch.float32)
    q = torch.empty(cout, cin, device=w.device, dtype=torch.float32)
    qt = torch.empty(cin, cout, device=w.device, dtype=torch.float32)
    N.call('sgdfr_modconv_prepack_f32', N.ptr(w), N.ptr(wp), N.ptr(q), N.ptr(qt), cout, cin, k, N.stream())
    return wp, q, qt


def prepack_t(weight, flip):
    """weight [1, Cout, Cin, k, k] -> [Cout, k*k, Cin] scaled (taps reversed when flip): weight pack of dL/dx."""
    N.require_device(weight)
    w = N.f32c(weight)
    _, cout, cin, k, _ = w.shape
    wt = torch.empty(cout, k * k, cin, device=w.device, dtype=torch.float32)
    N.call('sgdfr_modconv_prepack_t_f32', N.ptr(w), N.ptr(wt), cout, cin, k, int(bool(flip)), N.stream())
    return wt


def style_demod(style, mod_weight, mod_bias, q=None, cout=0):
    """s = modulation(style) (model.py:235) and, when q is given, d = rsqrt(sum (scale W s)^2 + 1e-8)
    (model.py:238-239) in the shared-weight form sum_i s^2 q[o,i]."""
    N.require_device(style, mod_weight, mod_bias, q)
    B, D = style.shape
    if style.stride(1) != 1:
        style = style.contiguous()
    ld = style.stride(0) if B > 1 else D
    cin = mod_weight.shape[0]
    s = torch.empty(B, cin, device=style.device, dtype=torch.float32)
    d = torch.empty(B, cout, device=style.device, dtype=torch.float32) if q is not None else None
    N.call('sgdfr_style_demod_f32', N.ptr(style), ld, N.ptr(N.f32c(mod_weight)), N.ptr(N.f32c(mod_bias)),
           N.ptr(q), N.ptr(s), N.ptr(d), B, D, cin, cout, N.stream())
    return s, d


def styles_batched(latent, specs, plans=None):
    """All modulations / demodulation coefficients of one forward in two launches.
    latent [B, L, D] contiguous; specs: list of (latent_index, mod_weight, mod_bias, q or None, cout).
    Returns [(s [B,cin], d [B,cout] or None), ...] as views into flat buffers.
    plans (optional, aligned with specs): None or (x_log2 | absmax word tensor, headroom) -- the range plan of the
    fp16-split conv (include/sgdfr.h, sgdfr_style_layer): that entry then comes back as (s * 2^e, d * 2^-e) per image."""
    N.require_device(latent)
    latent = N.f32c(latent)
    B, L, D = latent.shape
    if len(specs) > N.MAX_STYLE_LAYERS:
        raise RuntimeError('too many modulated layers (%d)' % len(specs))
    planned = [plans is not None and plans[i] is not None and specs[i][3] is not None for i in range(len(specs))]
    n_s = sum(mw.shape[0] for _, mw, _, _, _ in specs)
    n_d = sum(cout for _, _, _, q, cout in specs if q is not None)
    n_sn = sum(specs[i][1].shape[0] for i in range(len(specs)) if planned[i])
    n_dn = sum(specs[i][4] for i in range(len(specs)) if planned[i])
    buf = torch.empty(B * (n_s + n_d + n_sn + n_dn) + 1, device=latent.device, dtype=torch.float32)
    offs = [0]

    def take(n, shape):
        v = buf[offs[0]:offs[0] + n].view(shape)
        offs[0] += n
        return v
    arr = (N.StyleLayer * len(specs))()
    out, keep = [], []
    for i, (li, mw, mb, q, cout) in enumerate(specs):
        N.require_device(mw, mb, q)
        cin = mw.shape[0]
        s = take(B * cin, (B, cin))
        d = take(B * cout, (B, cout)) if q is not None else None
        e = arr[i]
        e.mod_w, e.mod_b = N.f32c(mw).data_ptr(), N.f32c(mb).data_ptr()
        e.q = q.data_ptr() if q is not None else None
        e.s, e.d = s.data_ptr(), (d.data_ptr() if d is not None else None)
        e.cin, e.cout, e.latent_index = cin, cout, li
        e.s_n = e.d_n = e.x_absmax = None
        e.x_log2 = e.headroom = 0
        if planned[i]:
            bound, headroom = plans[i]
            s, d = take(B * cin, (B, cin)), take(B * cout, (B, cout))
            e.s_n, e.d_n, e.headroom = s.data_ptr(), d.data_ptr(), int(headroom)
            if isinstance(bound, torch.Tensor):
                keep.append(bound)
                e.x_absmax = bound.data_ptr()
            else:
                e.x_log2 = int(bound)
        out.append((s, d))
    N.call('sgdfr_styles_batched_f32', N.ptr(latent), B, L, D, arr, len(specs), N.stream())
    return out


# ---- range plan of the fp16-split conv.  SGDFR_RANGE_PLAN / functional.RANGE_PLAN:
#   True  ('1', default)  Generator: calibrated once per weight version (max |x| of every conv input on the fp32 kernels, 6 binades
#                         of headroom), checked through the generator's saturation word; stand-alone layers: exact (below)
#   'exact'               every conv input's true per-image max |x| is measured on the device before the conv (absmax +
#                         split_range, headroom 0): fp16 saturation is impossible for finite inputs, no calibration and no host
#                         read -- at the price of fp32 hand-over between layers (no split chain, no fused ToRGB) and one extra
#                         read of every activation
#   False ('0')           off: the fixed 2^-4 pre-scale of round 1
# (Config.range_plan)
DESIGN_X_LOG2 = 10          # uncalibrated bound taken on trust for a conv input: |x| < 2^10
CALIBRATION_HEADROOM = 6    # binades kept free above a calibrated activation maximum before the fp16 terms saturate


def absmax(x, per_image=True, batch=None, out=None):
    """int32 words [B] (per image) or [1] (whole tensor): fp32 bit pattern of max |x| (sgdfr_absmax_f32).  A [1,...] tensor
    standing for `batch` images yields one word."""
    N.require_device(x)
    x = N.f32c(x)
    B = x.shape[0]
    shared = batch is not None and B == 1 and batch != 1
    per_image = bool(per_image) and not shared
    n = x[0].numel()
    if out is None:
        out = torch.empty(B if per_image else 1, device=x.device, dtype=torch.int32)
    if per_image:
        N.call('sgdfr_absmax_f32', N.ptr(x), n, n, B, N.ptr(out), 1, N.stream())
    else:       # one word over everything: a single "image" of B*n elements
        N.call('sgdfr_absmax_f32', N.ptr(x), B * n, B * n, 1, N.ptr(out), 0, N.stream())
    return out


def split_range(s, d, x_absmax=None, x_log2=DESIGN_X_LOG2, headroom=0):
    """(s * 2^e, d * 2^-e) per image with e from the range plan (sgdfr_split_range_f32): exact, and the conv result is the same."""
    N.require_device(s, d)
    s, d = N.f32c(s), N.f32c(d)
    B, cin = s.shape
    cout = d.shape[1]
    buf = torch.empty(B * (cin + cout), device=s.device, dtype=torch.float32)
    s_n, d_n = buf[:B * cin].view(B, cin), buf[B * cin:].view(B, cout)
    bstride = 0
    if x_absmax is not None:
        if x_absmax.dtype != torch.int32 or not x_absmax.is_cuda or not x_absmax.is_contiguous() or \
                (x_absmax.numel() != 1 and x_absmax.numel() % B != 0):
            raise RuntimeError('split_range: x_absmax must be the int32 device words made by absmax() (1, B, or n per image)')
        bstride = 0 if x_absmax.numel() == 1 and B > 1 else x_absmax.numel() // B
    N.call('sgdfr_split_range_f32', N.ptr(s), N.ptr(d), N.ptr(s_n), N.ptr(d_n), N.ptr(x_absmax), bstride, int(x_log2), int(headroom),
           B, cin, cout, N.stream())
    return s_n, d_n


_ones = {}


def ones_like_rows(B, C, device):
    """cached [B, C] tensor of ones (the neutral output scale of a range plan)."""
    key = (B, C, str(device))
    t = _ones.get(key)
    if t is None:
        if len(_ones) > 64:
            _ones.clear()
        t = _ones[key] = torch.ones(B, C, device=device, dtype=torch.float32)
    return t


def _exact_range(x, s, d, batch, x_absmax=None):
    """Range plan from the true max |x| of each image (two small launches): what a stand-alone layer call uses.  x_absmax: the
    per-plane words a producer already measured (one launch)."""
    B = s.shape[0] if batch is None else batch
    return split_range(s, d, x_absmax if x_absmax is not None else absmax(x, per_image=True, batch=B))


def _noise_args(noise, B, H, W):
    """(tensor, batch stride) for a [1,1,H,W] shared or [B,1,H,W] per-sample noise map."""
    if noise is None:
        return None, 0
    N.require_device(noise)
    noise = N.f32c(noise)
    if noise.numel() == H * W:
        return noise, 0
    if noise.numel() == B * H * W:
        return noise, H * W
    raise RuntimeError('noise of shape %s does not broadcast to [%d,1,%d,%d]' % (tuple(noise.shape), B, H, W))


# Inference chain, plain layers fed by a transposed conv + blur: 1-D Winograd form of the split conv (csrc/wsplit.hip), F(4,3)
# by default (half the MFMA work; F(2,3): 2/3).  The blur then writes 6 (F(2,3): 8) instead of 4 bytes per element, so it pays
# where the conv's K loop dominates: layers with at least WSPLIT_MIN_CIN input channels (same-box A/B at B=64,
# scripts/wsplit_ab.py + scripts/blur_wino_time.py; 0 = never).
# Arithmetic of the 3x3 modulated convs (inference path, autograd forward, and dL/dx of the plain convs; the strided dL/dx of
# the transposed convs and the weight gradients always use the fp32 MFMA kernels):
#   'fp16x3' (default)  fp32 operands split into fp16 hi + lo (11+11 significant bits), hi*hi + hi*lo + lo*hi on
#                       v_mfma_f32_32x32x16_f16, fp32 accumulation (csrc/split.hip).  fp16 has 5 exponent bits, so the operands
#                       are range-shifted by exact powers of two: statically (x*2^-4, W*2^6, result*2^-2) and per image by the
#                       RANGE PLAN below (styles scaled by 2^e, demodulation by 2^-e -- the conv is invariant to it), which
#                       puts the largest |x*s| of the layer `headroom` binades under the fp16 maximum.  Terms of an element
#                       keep 22 bits down to 2^-17 of that maximum and an absolute floor of 2^-39 of it below.  Operands that
#                       still leave the range (or are NaN/Inf) are clamped AND counted (split_saturation_count); Generator
#                       polls the count and falls back to 'bf16x3'.  Measured 7.7e-6 max-abs vs an fp64 evaluation on
#                       256x256 images (fp32 MFMA kernels: 9.5e-6); tests/test_gpu_split.py holds it to the fp32 kernels'
#                       per-layer bound RELATIVE to max|y| for inputs scaled by 2^-20 ... 2^15 and styles in [1e-3, 1e3].
#   'fp32'              fp32 MFMA kernels (direct + Winograd) everywhere.
#   'bf16x3'            as fp16x3 with bf16 terms: full fp32 range, 8+8 bits (~1e-4 on images; contract 1e-3).
_zeros = {}


# ---- saturation words.  Every launch that converts operands to the fp16 terms adds its count of clamped / non-finite operand
# pairs to ONE device word chosen by the caller (include/sgdfr.h "saturation words"): a Generator owns one, so saturation is
# attributed to the generator (and batch) that caused it, never to a neighbour in the same process.  `saturation_sink(word)`
# names the word for the launches issued inside the block by this host thread; autograd Functions remember it for their
# backward.  Outside any block the launches count into the device-wide legacy counter (split_saturation_count()).
_sink = threading.local()


def current_sink():
    return getattr(_sink, 'word', None)


class saturation_sink:
    """`with saturation_sink(word):` -- word: int32 device tensor [1] (new_saturation_word) or None (legacy counter)."""

    def __init__(self, word):
        if word is not None and (word.dtype != torch.int32 or not word.is_cuda or word.numel() != 1):
            raise RuntimeError('a saturation word is one int32 on the device (new_saturation_word)')
        self.word, self.prev = word, None

    def __enter__(self):
        self.prev = current_sink()
        _sink.word = self.word
        return self.word

    def __exit__(self, *exc):
        _sink.word = self.prev


def new_saturation_word(device):
    return torch.zeros(1, device=device, dtype=torch.int32)


def _sat():
    return N.ptr(getattr(_sink, 'word', None))


def split_saturation_count(reset=True):
    """The device-wide LEGACY counter: operand pairs clamped (or NaN/Inf) by fp16x3 launches issued outside any
    saturation_sink on this device since the last reset (stand-alone functional calls; a Generator counts into its own
    word, Generator.saturated_pairs()).  Synchronises the device."""
    n = N.load().sgdfr_split_saturation_count(int(bool(reset)))
    if n < 0:
        raise RuntimeError('sgdfr_split_saturation_count failed')
    return int(n)


def mfma_ceiling(arith='fp16x3', lds_fragments=True, random_operands=True, iters=2000):
    """Measured sustained rate of the 16-bit MFMA of the split kernels on the current device, TFLOP/s of 16-bit products
    (sgdfr_mfma_ceiling_probe: the chip clocks to its power budget, so random operands run slower than the nominal peak)."""
    import ctypes
    if not hasattr(N.load(), 'sgdfr_mfma_ceiling_probe'):
        raise RuntimeError('this build of libsgdfr_hip.so has no sgdfr_mfma_ceiling_probe (measurement-only symbol)')
    scratch = torch.empty(256 * 512, device='cuda', dtype=torch.float32)
    out = ctypes.c_double(0.0)
    N.call('sgdfr_mfma_ceiling_probe', N.SPLIT_FP16 if arith == 'fp16x3' else N.SPLIT_BF16, int(bool(lds_fragments)),
           int(bool(random_operands)), int(iters), 256, N.ptr(scratch), ctypes.byref(out), N.stream())
    return float(out.value)


class capture_graph:
    """`with capture_graph(g):` -- torch.cuda.graph(g) with the garbage collector held off for the duration of the capture.
    A collection that runs mid-capture can finalise an unrelated CUDAGraph (an old generator's or session's), and
    hipGraphDestroy is "not permitted when stream is capturing": the process aborts in the destructor.  torch.cuda.graph
    collects once on entry; nothing is collected until the capture has ended."""

    def __init__(self, g):
        self.ctx = torch.cuda.graph(g, capture_error_mode='thread_local')
        self.was = False

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            return self.ctx.__enter__()
        except BaseException:
            if self.was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was:
                gc.enable()


def _zero_words(device):
    z = _zeros.get(device)
    if z is None:
        z = _zeros[device] = torch.zeros(64, device=device, dtype=torch.float32)
    return z


def prepack_wino(weight, adjoint=False):
    """weight [1,Cout,Cin,3,3] -> Winograd-domain pack U [Cin][Cout][16] (16-byte quads XOR-swizzled by cout & 3;
    adjoint: [Cout][Cin][16] of the rotated, transposed kernel = the pack of dL/dx)."""
    N.require_device(weight)
    w = N.f32c(weight)
    _, cout, cin, k, _ = w.shape
    u = torch.empty((cout if adjoint else cin), (cin if adjoint else cout), 16, device=w.device, dtype=torch.float32)
    N.call('sgdfr_modconv_prepack_wino_f32', N.ptr(w), N.ptr(u), cout, cin, int(bool(adjoint)), N.stream())
    return u


def wino_ok(B, cin, cout, H, W):
    if not config().use_winograd:
        return False
    if not N.load().sgdfr_modconv2d_wino_supported(B, cin, cout, H, W):
        return False
    return ((B * (H // 2) * (W // 2) + 63) // 64) * (cout // 64) >= config().winograd_min_blocks


def modconv_wino(x, u, s, d, cout, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2, gain=SQRT2,
                 batch=None, desc=None):
    """Plain 3x3 modulated conv through the Winograd kernel (same contract as modconv_raw(mode PLAIN3))."""
    N.require_device(x, u, s, d, bias, noise_weight)
    x = N.f32c(x)
    B = s.shape[0] if batch is None else batch
    _, cin, H, W = x.shape
    xb = 0 if (x.shape[0] == 1 and B != 1) else cin * H * W
    nz, nzb = _noise_args(noise, B, H, W)
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    st = N.stream()
    _timed_conv(desc or ('wino3 %d->%d @%dx%d' % (cin, cout, H, W)), B * conv_flops(cin, cout, H, W), lambda: N.call(
        'sgdfr_modconv2d_wino_f32', N.ptr(x), xb, N.ptr(u), N.ptr(s), N.ptr(d), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(_zero_words(x.device)), N.ptr(y), B, cin,
        cout, H, W, int(activate), float(slope), float(gain), st))
    return y


_SPLIT_ARITH = {'bf16x3': N.SPLIT_BF16, 'fp16x3': N.SPLIT_FP16}
# the F(4,3) wide-tile conv, its pack and its WS producers also take fp16 + fp8 cross terms (include/sgdfr.h SGDFR_SPLIT_FP16F8)
_WSPLIT_ARITH = dict(_SPLIT_ARITH, fp16f8=N.SPLIT_FP16F8)


def _f8_tag(arith):
    """Suffix of a conv launch's timing description when its cross terms run as ONE fp8 MFMA (SGDFR_SPLIT_FP16F8): bench.py prices
    those rows against 1/(1/2500 + 1/5000) = 1667 TFLOP/s (one fp16 + one fp8 MFMA per product), not the three-fp16-product peak."""
    return ' [f8 cross]' if (arith & 0xff) == N.SPLIT_FP16F8 else ''


def _plan_tag(B, cin, cout, H, W, mode, ks):
    """'/deep' for transposed-conv launches on split_kernel.h's deep plan (<1,1,2,4,1,2,2,1,true>: all nine taps of a channel block per
    stage -- the kernel VERDICT r5 calls dominant), so bench.py can report that instantiation apart from the small K-sliced layers."""
    if mode == N.MODE_UP3 and ks <= 1 and timing.active() is not None and _shape_query('sgdfr_modconv2d_split_f8_ok', B, cin, cout, H, W, mode):
        return '/deep'
    return ''


def prepack_split(weight, arith=None, adjoint=False):
    """weight [1,Cout,Cin,3,3] -> uint16 buffer of 16-bit hi/lo terms of weight/sqrt(9 Cin) in split.hip's LDS order
    (arith: 'bf16x3' or 'fp16x3', default = the current PRECISION; adjoint: True = the pack of dL/dx of the plain conv,
    'down' = of dL/dx of the transposed conv, mode DOWN3)."""
    arith = _WSPLIT_ARITH[arith or config().precision]      # ('fp16f8': forward packs of the transposed conv's deep plan)
    N.require_device(weight)
    w = N.f32c(weight)
    _, cout, cin, k, _ = w.shape
    n = N.load().sgdfr_modconv_prepack_split_elems(cout, cin)
    wsp = torch.empty(n, device=w.device, dtype=torch.int16)
    N.call('sgdfr_modconv_prepack_split_f32', N.ptr(w), N.ptr(wsp), cout, cin, arith, 2 if adjoint == 'down' else int(bool(adjoint)),
           _sat(), N.stream())
    return wsp


@functools.lru_cache(maxsize=None)
def _shape_query(name, *shape):
    """Pure shape -> int queries of the library (tiling plans), memoised: the launch-bound small batches feel every call."""
    return getattr(N.load(), name)(*shape)


def split_ok(B, cin, cout, H, W, mode=N.MODE_PLAIN3):
    return config().precision in _SPLIT_ARITH and bool(_shape_query('sgdfr_modconv2d_split_supported', B, cin, cout, H, W, mode))


def modconv_split(x, wsp, s, d, cout, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2, gain=SQRT2,
                  batch=None, desc=None, mode=N.MODE_PLAIN3, arith=None, rgb=None, out=None, want_y=True, x_split=None, s_next=None,
                  plane_stride=0):
    """3x3 modulated conv in a split arithmetic (same contract as modconv_raw, modes PLAIN3 and UP3); `wsp` must have
    been packed for the same `arith`.  rgb = (w_rgb [3,Cout], s_rgb [B,Cout]) also returns the per-cout-tile partial sums
    [B, T*3, H, W] of the ToRGB 1x1 conv that follows the layer (see rgb_fusable / torgb_finish)."""
    arith = _WSPLIT_ARITH[arith or config().precision]      # ('fp16f8': where sgdfr_modconv2d_split_f8_ok says so; the library checks)
    N.require_device(s, d, bias, noise_weight)
    if not wsp.is_cuda or wsp.dtype != torch.int16:
        raise RuntimeError('modconv_split: wsp must be the int16 device buffer made by prepack_split')
    if x_split is not None:         # (B, Cin, H, W) of the pre-split int16 input made by to_split(): x*s is already applied
        if not x.is_cuda or x.dtype != torch.int16:
            raise RuntimeError('modconv_split: a pre-split input is the int16 buffer made by to_split')
        B, cin, H, W = x_split
        xb = cin * H * W
        if mode == N.MODE_DOWN3:        # (B, plane channels, H, W of dL/dx): the planes are (H+1) x (W+1), 4 per channel
            xb = cin * 4 * (H + 1) * (W + 1)
    else:
        N.require_device(x)
        x = N.f32c(x)
        B = s.shape[0] if batch is None else batch
        _, cin, H, W = x.shape
        xb = 0 if (x.shape[0] == 1 and B != 1) else cin * H * W
    if mode == N.MODE_DOWN3:
        if x_split is None:
            raise RuntimeError('modconv_split: DOWN3 reads the buffer made by planes_to_split (x_split=(B, C, H, W))')
        nz, nzb = None, 0
        y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    elif mode == N.MODE_UP3:
        nz, nzb = None, 0
        pshape = (B, cout, 4, H + 1, W + 1) if not plane_stride else (B, cout, 4, int(plane_stride))      # padded planes: flat
        y = out if out is not None else torch.empty(pshape, device=x.device, dtype=torch.float32)
        if tuple(y.shape) != pshape or not y.is_contiguous():
            raise RuntimeError('modconv_split: out must be a contiguous %s tensor' % (pshape,))
    else:
        nz, nzb = _noise_args(noise, B, H, W)
        if not want_y and rgb is None and s_next is None:
            raise RuntimeError('modconv_split: want_y=False only makes sense with the fused ToRGB (rgb=...) or s_next')
        y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32) if want_y else None
    st, sat = N.stream(), _sat()
    ks = _shape_query('sgdfr_modconv2d_split_ksplit_hint', B, cin, cout, H, W, mode) if config().use_splitk else 1
    if ks > 1 and y is None:
        raise RuntimeError('modconv_split: this launch is K-sliced and cannot fuse ToRGB (check rgb_fusable first)')
    partials = torch.empty((ks,) + tuple(y.shape), device=x.device, dtype=torch.float32) if ks > 1 else None
    rgb_w = rgb_s = part = xs_out = None
    if s_next is not None:
        if ks > 1 or mode != N.MODE_PLAIN3:
            raise RuntimeError('modconv_split: this launch cannot emit the split activation (K-sliced or not PLAIN3)')
        N.require_device(s_next)
        s_next = N.f32c(s_next)
        xs_out = torch.empty(B, cout // 8, 2, H * W, 8, device=s_next.device, dtype=torch.int16)
    if rgb is not None:
        if ks > 1 or mode != N.MODE_PLAIN3:
            raise RuntimeError('modconv_split: this launch cannot fuse ToRGB (check rgb_fusable first)')
        rgb_w, rgb_s = N.f32c(rgb[0]), N.f32c(rgb[1])
        N.require_device(rgb_w, rgb_s)
        tiles = _shape_query('sgdfr_modconv2d_split_cout_tiles_xin' if (x_split is not None and ks <= 1) else
                             'sgdfr_modconv2d_split_cout_tiles', B, cin, cout, H, W, mode)
        part = torch.empty(B, tiles * 3, H, W, device=x.device, dtype=torch.float32)
    _timed_conv((desc or ('split mode%d%s %d->%d @%dx%d%s' % (mode, _plan_tag(B, cin, cout, H, W, mode, ks), cin, cout, H, W,
                                                              ' K/%d' % ks if ks > 1 else ''))) + _f8_tag(arith),
                B * conv_flops(cin, cout, H, W), lambda: N.call(
        'sgdfr_modconv2d_split_f32', N.ptr(x), xb, N.ptr(wsp), N.ptr(s), N.ptr(d), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(_zero_words(x.device)), N.ptr(y),
        N.ptr(partials), ks, N.ptr(rgb_w), N.ptr(rgb_s), N.ptr(part), int(x_split is not None), N.ptr(xs_out),
        N.ptr(s_next) if s_next is not None else None, B, cin, cout, H, W, mode, int(plane_stride), arith, int(activate),
        float(slope), float(gain), sat, st))
    if s_next is not None:      # (activation or None, ToRGB partials or None, the activation in the next layer's split form)
        return y, part, xs_out
    return y if rgb is None else (y, part)


# ---- 1-D Winograd F(2,3) form of the plain split conv (csrc/wsplit.hip): 2/3 of the MFMA work of the direct split kernel
def prepack_wsplit(weight, arith=None, f=2):
    """weight [1,Cout,Cin,3,3] -> int16 buffer of the hi/lo terms of U = G (weight/sqrt(9 Cin)) per kernel row in wsplit.hip's
    LDS order (Cout % 128 == 0); f = outputs per Winograd tile (2: F(2,3), 4: F(4,3))."""
    arith = _WSPLIT_ARITH[arith or config().precision]
    N.require_device(weight)
    w = N.f32c(weight)
    _, cout, cin, k, _ = w.shape
    wsp = torch.empty(N.load().sgdfr_modconv_prepack_wsplit_elems(cout, cin, f), device=w.device, dtype=torch.int16)
    N.call('sgdfr_modconv_prepack_wsplit_f32', N.ptr(w), N.ptr(wsp), cout, cin, f, arith, _sat(), N.stream())
    return wsp


def wsplit_ok(B, cin, cout, H, W, f=2):
    return config().precision in _SPLIT_ARITH and bool(_shape_query('sgdfr_modconv2d_wsplit_supported', B, cin, cout, H, W, f))


WSPLIT_GROWTH_LOG2 = {2: 1, 4: 4}      # |B^T d| <= 2 max|d| (F(2,3)) / 10 max|d| (F(4,3)): binades the range plan adds


def to_wsplit(x, s, arith=None, f=2):
    """x [B,Cin,H,W], s [B,Cin] -> int16 buffer [B, Cin/8, f+2, 2, H*W/f, 8]: the Winograd input transform of x*s per tile of f
    outputs, split (the "WS" form modconv_wsplit stages by DMA)."""
    arith = _WSPLIT_ARITH[arith or config().precision]
    N.require_device(x, s)
    x, s = N.f32c(x), N.f32c(s)
    B, cin, H, W = x.shape
    vs = torch.empty(B, cin // 8, f + 2, 2, H * W // f, 8, device=x.device, dtype=torch.int16)
    N.call('sgdfr_to_wsplit_f32', N.ptr(x), N.ptr(s), N.ptr(vs), B, cin, H, W, f, arith, _sat(), N.stream())
    return vs


def modconv_wsplit(vs, shape, wsp, d, cout, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2, gain=SQRT2,
                   arith=None, rgb=None, want_y=True, s_next=None, desc=None, f=2, xs_arith=None):
    """Plain 3x3 modulated conv of a WS input (to_wsplit / the blur's Winograd hand-over; shape = (B, Cin, H, W)) with the pack of
    prepack_wsplit (same f).  Outputs as modconv_split with a pre-split input: y | (y, part) | (y, part, xs_out).  xs_arith='fp16f8':
    the hand-over xs_out carries the fp8 cross-term operands (for a next conv launched with that arithmetic; f = 4)."""
    arith = _WSPLIT_ARITH[arith or config().precision]
    if xs_arith not in (None, 'fp16f8'):
        raise ValueError("modconv_wsplit: xs_arith is None or 'fp16f8'")
    if xs_arith == 'fp16f8':
        arith |= N.SPLIT_HANDOVER_F8
    N.require_device(d, bias, noise_weight)
    if not vs.is_cuda or vs.dtype != torch.int16 or not wsp.is_cuda or wsp.dtype != torch.int16:
        raise RuntimeError('modconv_wsplit: vs / wsp are the int16 device buffers made by to_wsplit / prepack_wsplit')
    B, cin, H, W = shape
    nz, nzb = _noise_args(noise, B, H, W)
    if not want_y and rgb is None and s_next is None:
        raise RuntimeError('modconv_wsplit: want_y=False only makes sense with the fused ToRGB (rgb=...) or s_next')
    y = torch.empty(B, cout, H, W, device=vs.device, dtype=torch.float32) if want_y else None
    rgb_w = rgb_s = part = xs_out = None
    if s_next is not None:
        N.require_device(s_next)
        s_next = N.f32c(s_next)
        xs_out = torch.empty(B, cout // 8, 2, H * W, 8, device=vs.device, dtype=torch.int16)
    if rgb is not None:
        rgb_w, rgb_s = N.f32c(rgb[0]), N.f32c(rgb[1])
        N.require_device(rgb_w, rgb_s)
        part = torch.empty(B, (cout // 128) * 3, H, W, device=vs.device, dtype=torch.float32)
    _timed_conv((desc or ('wsplit F(%d,3) %d->%d @%dx%d' % (f, cin, cout, H, W))) + _f8_tag(arith), B * conv_flops(cin, cout, H, W), lambda: N.call(
        'sgdfr_modconv2d_wsplit_f32', N.ptr(vs), N.ptr(wsp), N.ptr(d), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(_zero_words(vs.device)), N.ptr(y), N.ptr(rgb_w),
        N.ptr(rgb_s), N.ptr(part), N.ptr(xs_out), N.ptr(s_next) if s_next is not None else None, B, cin, cout, H, W, f, arith,
        int(activate), float(slope), float(gain), _sat(), N.stream()))
    if s_next is not None:
        return y, part, xs_out
    return y if rgb is None else (y, part)


def planes_to_split(gt, d, arith=None):
    """gt [B,C,4,H+1,W+1] (gradient of the transposed conv's parity planes), d [B,C] or None -> int16 buffer
    [B, 4*C/8, 2, (H+1)*(W+1), 8]: gt*d in the phase-major split form that modconv_split(mode=DOWN3) stages."""
    arith = _SPLIT_ARITH[arith or config().precision]
    N.require_device(gt, d)
    gt = N.f32c(gt)
    B, C, _, R, P = gt.shape
    xs = torch.empty(B, 4 * C // 8, 2, R * P, 8, device=gt.device, dtype=torch.int16)
    N.call('sgdfr_planes_to_split_f32', N.ptr(gt), N.ptr(N.f32c(d)) if d is not None else None, N.ptr(xs), B, C, R - 1, P - 1, arith,
           _sat(), N.stream())
    return xs


def to_split(x, s, arith=None):
    """x [B,Cin,H,W], s [B,Cin] -> int16 buffer [B, Cin/8, 2, H*W, 8]: x*s in the split form the kernels stage
    (modconv_split(x=that, s=None, x_split=(B,Cin,H,W)) then fills LDS by DMA)."""
    arith = _WSPLIT_ARITH[arith or config().precision]
    N.require_device(x, s)
    x, s = N.f32c(x), N.f32c(s)
    B, cin, H, W = x.shape
    xs = torch.empty(B, cin // 8, 2, H * W, 8, device=x.device, dtype=torch.int16)
    N.call('sgdfr_to_split_f32', N.ptr(x), N.ptr(s), N.ptr(xs), B, cin, H, W, arith, _sat(), N.stream())
    return xs


class U8Target:
    """Where a generator forward should leave its image as uint8 HWC (libs/utilities/image_utils.py:87-110 scaling) instead of
    fp32 NCHW: `frames` [B, H, K*W, 3] uint8 on the device (K = 1: plain frames), the image goes to panel `panel` (columns
    panel*W ...), swap_rb applies the video writers' channel swap.  frames=None: a fresh [B,H,W,3] tensor."""
    __slots__ = ('frames', 'panel', 'swap_rb')

    def __init__(self, frames=None, panel=0, swap_rb=False):
        self.frames, self.panel, self.swap_rb = frames, int(panel), bool(swap_rb)


def torgb_finish(part, bias=None, skip=None, fir=None, u8=None):
    """part [B, T*3, H, W] (from modconv_split(rgb=...)) -> rgb [B,3,H,W] = sum over the T cout tiles + bias + upsampled
    skip (one small launch; the activation is not read again).  u8 = U8Target: the result is written as uint8 HWC
    straight from the sums (the fp32 image is never stored) and the uint8 tensor is returned."""
    N.require_device(part, bias, skip, fir)
    B, c3, H, W = part.shape
    if skip is not None and tuple(skip.shape) != (B, 3, H // 2, W // 2):
        raise RuntimeError('skip shape %s does not match output [%d,3,%d,%d]/2' % (tuple(skip.shape), B, H, W))
    skip_bytes = 3 * (H // 2) * (W // 2) * 4 if skip is not None else 0
    if u8 is not None:
        frames = u8.frames
        if frames is None:
            frames = torch.empty(B, H, W, 3, device=part.device, dtype=torch.uint8)
        if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous() or frames.ndim != 4 or \
                frames.shape[0] != B or frames.shape[1] != H or frames.shape[3] != 3 or frames.shape[2] < (u8.panel + 1) * W:
            raise RuntimeError('uint8 target %s cannot hold panel %d of [%d,%d,%d,3] frames' % (tuple(frames.shape), u8.panel, B, H, W))
        # algorithmic bytes: the partial sums once, the skip once, one byte per output channel and pixel
        _timed_hbm('torgb_finish_u8 T=%d @%dx%d' % (c3 // 3, H, W), B * (c3 * H * W * 4 + skip_bytes + 3 * H * W), lambda: N.call(
            'sgdfr_torgb_finish_u8_f32', N.ptr(part), c3 // 3, N.ptr(N.f32c(bias)) if bias is not None else None,
            N.ptr(N.f32c(skip)) if skip is not None else None, N.ptr(N.f32c(fir)) if fir is not None else None, N.ptr(frames),
            frames.shape[2] * 3, u8.panel * W, int(u8.swap_rb), B, H, W, N.stream()))
        return frames
    y = torch.empty(B, 3, H, W, device=part.device, dtype=torch.float32)
    _timed_hbm('torgb_finish T=%d @%dx%d' % (c3 // 3, H, W), B * (c3 * H * W * 4 + skip_bytes + 3 * H * W * 4), lambda: N.call(
        'sgdfr_torgb_finish_f32', N.ptr(part), c3 // 3, N.ptr(N.f32c(bias)) if bias is not None else None,
        N.ptr(N.f32c(skip)) if skip is not None else None, N.ptr(N.f32c(fir)) if fir is not None else None, N.ptr(y), B, H, W,
        N.stream()))
    return y


def modconv_raw(x, wp, s, d, cout, mode, H, W, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2,
                gain=SQRT2, batch=None, desc=None):
    """One sgdfr_modconv2d_fwd_f32 launch.  mode PLAIN3: x [B,Cin,H,W] -> [B,cout,H,W]; UP3: -> parity planes
    [B,cout,4,H+1,W+1]; DOWN3: x = planes [B,Cin,4,H+1,W+1] -> [B,cout,H,W]."""
    N.require_device(x, wp, s, d, bias, noise_weight)
    x = N.f32c(x)
    B = s.shape[0] if batch is None else batch
    cin = x.shape[1]
    per_img = x[0].numel()
    xb = 0 if (x.shape[0] == 1 and B != 1) else per_img
    if x.shape[0] not in (1, B):
        raise RuntimeError('input batch %d does not match styles %d' % (x.shape[0], B))
    if mode == N.MODE_UP3:
        y = torch.empty(B, cout, 4, H + 1, W + 1, device=x.device, dtype=torch.float32)
    else:
        y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    nz, nzb = _noise_args(noise, B, H, W) if mode == N.MODE_PLAIN3 else (None, 0)
    st = N.stream()
    splits = N.load().sgdfr_modconv2d_splitk_hint(B, cin, cout, H, W, mode) if config().use_splitk else 1
    if splits > 1:      # too few tiles to fill the chip: slice K across extra blocks, reduce deterministically
        partials = torch.empty((splits,) + tuple(y.shape), device=x.device, dtype=torch.float32)
        _timed_conv((desc or 'conv') + ' splitK%d' % splits, B * conv_flops(cin, cout, H, W), lambda: N.call(
            'sgdfr_modconv2d_splitk_f32', N.ptr(x), xb, N.ptr(wp), N.ptr(s), N.ptr(d), N.ptr(nz), nzb,
            N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(y), N.ptr(partials), splits, B, cin, cout,
            H, W, mode, int(activate), float(slope), float(gain), st))
        return y
    _timed_conv(desc or ('mode%d %d->%d @%dx%d' % (mode, cin, cout, H, W)), B * conv_flops(cin, cout, H, W), lambda: N.call(
        'sgdfr_modconv2d_fwd_f32', N.ptr(x), xb, N.ptr(wp), N.ptr(s), N.ptr(d), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(y), B, cin, cout, H, W, mode,
        int(activate), float(slope), float(gain), st))
    return y


def blur_bias_act(planes, fir, H, W, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2, gain=SQRT2, out=None,
                  absmax_out=None):
    """planes [B,C,4,H+1,W+1] -> [B,C,2H,2W]: 4x4 FIR (pad 1,1) + noise + bias + leaky-ReLU.  absmax_out: a ZEROED int32 [B,C]
    tensor that receives the bit pattern of max |y| per (image, channel) plane (for the range plan of the conv that reads y)."""
    N.require_device(planes, fir, bias, noise_weight)
    B, C = planes.shape[0], planes.shape[1]
    nz, nzb = _noise_args(noise, B, 2 * H, 2 * W)
    y = out if out is not None else torch.empty(B, C, 2 * H, 2 * W, device=planes.device, dtype=torch.float32)
    if tuple(y.shape) != (B, C, 2 * H, 2 * W) or not y.is_contiguous():
        raise RuntimeError('blur_bias_act: out must be a contiguous [B,C,2H,2W] tensor')
    if absmax_out is not None and (absmax_out.dtype != torch.int32 or not absmax_out.is_cuda or absmax_out.numel() != B * C or
                                   not absmax_out.is_contiguous()):
        raise RuntimeError('blur_bias_act: absmax_out must be a contiguous int32 device tensor with B*C words')
    # algorithmic bytes: every parity plane read once (4 (H+1)(W+1) floats per channel), every output element written once
    _timed_hbm('blur %d ch %dx%d -> %dx%d fp32' % (C, H, W, 2 * H, 2 * W), B * C * (16 * (H + 1) * (W + 1) + 16 * H * W), lambda: N.call(
        'sgdfr_blur_bias_act_f32', N.ptr(planes), N.ptr(N.f32c(fir)), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(y), B, C, H, W, int(activate),
        float(slope), float(gain), N.ptr(absmax_out), N.stream()))
    return y


def blur_bias_act_split(planes, fir, H, W, s_next, noise=None, noise_weight=None, bias=None, activate=False, slope=0.2,
                        gain=SQRT2, arith=None, plane_stride=0, wino=False):
    """blur_bias_act whose result goes out as the next layer's split input (x * s_next as 16-bit hi/lo pairs,
    [B, C/8, 2, 2H*2W, 8] int16) instead of fp32 NCHW.  plane_stride: floats between the parity planes when `planes` is the
    padded [B, C, 4, plane_stride] buffer of modconv_split(mode=UP3, plane_stride=...).  wino = 2 | 4: the Winograd input form
    of to_wsplit(f=wino) instead ([B, C/8, wino+2, 2, 4HW/wino, 8], for modconv_wsplit; wino = 4 also takes arith='fp16f8')."""
    wino = 2 if wino is True else int(wino or 0)
    arith = (_WSPLIT_ARITH if wino != 2 else _SPLIT_ARITH)[arith or config().precision]
    N.require_device(planes, fir, bias, noise_weight, s_next)
    B, C = planes.shape[0], planes.shape[1]
    nz, nzb = _noise_args(noise, B, 2 * H, 2 * W)
    if wino:
        xs = torch.empty(B, C // 8, wino + 2, 2, 4 * H * W // wino, 8, device=planes.device, dtype=torch.int16)
    else:
        xs = torch.empty(B, C // 8, 2, 4 * H * W, 8, device=planes.device, dtype=torch.int16)
    # algorithmic bytes: the planes once + the hand-over (4 bytes per element in the direct split form, 4 (f+2)/f in Winograd form)
    out_b = 4 * H * W * (4 * (wino + 2) // wino if wino else 4)
    _timed_hbm('blur %d ch %dx%d -> %dx%d %s' % (C, H, W, 2 * H, 2 * W, 'WS F(%d,3)' % wino if wino else 'XS'),
               B * C * (16 * (H + 1) * (W + 1) + out_b), lambda: N.call(
        'sgdfr_blur_bias_act_split_f32', N.ptr(planes), N.ptr(N.f32c(fir)), N.ptr(nz), nzb,
        N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(N.f32c(s_next)), N.ptr(xs), B, C, H, W,
        int(plane_stride), arith, wino, int(activate), float(slope), float(gain), _sat(), N.stream()))
    return xs


def modconv3x3(x, wp, s, d, cout, upsample=False, fir=None, noise=None, noise_weight=None, bias=None,
               activate=False, slope=0.2, gain=SQRT2, batch=None, return_planes=False, wino=None, split=None, rgb=None,
               want_y=True, ranged=False, x_absmax=None, absmax_out=None):
    """Shared-weight modulated 3x3 conv (model.py:232-273) with the StyledConv tail fused in
    (noise model.py:287, bias + leaky-ReLU op/fused_act.py:81-86).
    x_absmax: int32 [B,Cin] words of max |x| per plane when the producer of x already measured them (blur_bias_act(absmax_out=)):
    the exact range plan then needs no pass over x; absmax_out (upsample=True): zeroed int32 [B,Cout] for the same words of the result.

    x [B,Cin,H,W], or a [1,Cin,H,W] constant broadcast over `batch` images (ConstantInput,
    model.py:296-300, without materialising the repeat).  upsample=True runs the stride-2 transposed
    conv into parity planes and finishes with the 4x4 FIR pass (model.py:246-257)."""
    _, cin, H, W = x.shape
    if not upsample:
        B = s.shape[0] if batch is None else batch
        if split is not None and split_ok(B, cin, cout, H, W):
            if config().range_plan and not ranged and config().precision == 'fp16x3' and d is not None:
                s, d = _exact_range(x, s, d, batch, x_absmax)       # (ranged: the caller's s, d already carry a plan)
            return modconv_split(x, split() if callable(split) else split, s, d, cout, noise, noise_weight, bias,
                                 activate, slope, gain, batch, rgb=rgb, want_y=want_y)
        if rgb is not None:
            raise RuntimeError('modconv3x3: ToRGB fusion needs the split kernel (check rgb_fusable first)')
        if wino is not None and wino_ok(B, cin, cout, H, W):
            return modconv_wino(x, wino() if callable(wino) else wino, s, d, cout, noise, noise_weight, bias, activate,
                                slope, gain, batch)
        return modconv_raw(x, wp, s, d, cout, N.MODE_PLAIN3, H, W, noise, noise_weight, bias, activate, slope, gain,
                           batch, 'plain3 %d->%d @%dx%d' % (cin, cout, H, W))
    if fir is None:
        raise RuntimeError('upsample modconv needs the blur FIR taps')
    Bu = s.shape[0] if batch is None else batch
    use_split = split is not None and split_ok(Bu, cin, cout, H, W, N.MODE_UP3)
    if use_split:
        if config().range_plan and not ranged and config().precision == 'fp16x3' and d is not None:
            s, d = _exact_range(x, s, d, batch, x_absmax)
        planes = modconv_split(x, split() if callable(split) else split, s, d, cout, batch=batch, mode=N.MODE_UP3)
    else:
        planes = modconv_raw(x, wp, s, d, cout, N.MODE_UP3, H, W, batch=batch, desc='up3 %d->%d @%dx%d' % (cin, cout, H, W))
    y = blur_bias_act(planes, fir, H, W, noise, noise_weight, bias, activate, slope, gain, absmax_out=absmax_out)
    return (y, planes) if return_planes else y


def torgb(x, w_rgb, s, bias=None, skip=None, fir=None):
    """ToRGB (model.py:350-359): 1x1 modconv without demodulation + bias + FIR-upsampled skip."""
    N.require_device(x, w_rgb, s, bias, skip, fir)
    x = N.f32c(x)
    B, cin, H, W = x.shape
    if skip is not None and tuple(skip.shape) != (B, 3, H // 2, W // 2):
        raise RuntimeError('skip shape %s does not match output [%d,3,%d,%d]/2' % (tuple(skip.shape), B, H, W))
    y = torch.empty(B, 3, H, W, device=x.device, dtype=torch.float32)
    _timed_hbm('torgb %d ch @%dx%d' % (cin, H, W), B * ((cin + 3) * H * W * 4 + (3 * (H // 2) * (W // 2) * 4 if skip is not None else 0)),
               lambda: N.call(
        'sgdfr_torgb_fwd_f32', N.ptr(x), N.ptr(N.f32c(w_rgb)), N.ptr(s), N.ptr(N.f32c(bias)) if bias is not None else None,
        N.ptr(N.f32c(skip)) if skip is not None else None, N.ptr(N.f32c(fir)) if fir is not None else None,
        N.ptr(y), B, cin, H, W, N.stream()))
    return y


def conv_flops(cin, cout, h, w, upsample=False):
    """Algorithmic FLOPs of one 3x3 modconv on an h x w INPUT (the transposed conv is
    counted at input resolution, 9 MACs per input pixel per (cin, cout))."""
    return 2.0 * 9 * cin * cout * h * w


# ------------------------------------------------------------------ differentiable affine map

class _Affine(Function):
    """y = x @ W.T + b with all three gradients computed by the same HIP linear kernel
    (dX = g @ W, dW = g.T @ x, db = 1.T @ g).  Used by DirectionMatrix, the only trainable block of the
    reference's trainer (libs/trainer.py:144,175)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g2 = g.reshape(-1, weight.shape[0])
        x2 = x.reshape(-1, weight.shape[1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = linear(g2, weight.t().contiguous()).view_as(x)
        if ctx.needs_input_grad[1]:
            gw = linear(g2.t().contiguous(), x2.t().contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ones = torch.ones(1, g2.shape[0], device=g.device, dtype=torch.float32)
            gb = linear(ones, g2.t().contiguous()).view(-1)
        return gx, gw, gb


def affine(x, weight, bias=None):
    return _Affine.apply(x, weight, bias)


# ------------------------------------------------------------------ backward launches

def act_grad_reduce(g_out, out, noise, noise_weight, bias, want_y, slope=0.2, gain=SQRT2, want_absmax=False):
    """g_pre and sums [B,C,3] = (sum g_pre, sum g_pre*noise, sum g_pre*y); want_absmax: also absmax-style words [B,C] (bit pattern
    of max |g_pre| per plane, from the same pass: split_range() takes them for the plan of the fp16-split dL/dx conv)."""
    N.require_device(g_out, out, noise_weight, bias)
    g_out, out = N.f32c(g_out), N.f32c(out)
    B, C, H, W = out.shape
    nz, nzb = _noise_args(noise, B, H, W)
    g_pre = torch.empty_like(out)
    buf = torch.empty(B * C * (4 if want_absmax else 3), device=out.device, dtype=torch.float32)      # one memset clears both
    sums = buf[:B * C * 3].view(B, C, 3)
    gmax = buf[B * C * 3:].view(torch.int32).view(B, C) if want_absmax else None
    N.call('sgdfr_act_grad_reduce_f32', N.ptr(g_out), N.ptr(out), N.ptr(nz), nzb,
           N.ptr(noise_weight) if nz is not None else None, N.ptr(bias), N.ptr(g_pre), N.ptr(sums), B, C, H * W,
           float(slope), float(gain), int(bool(want_y)), N.ptr(gmax), N.stream())
    return (g_pre, sums, gmax) if want_absmax else (g_pre, sums)


def blur_adjoint(g, fir, planes=None):
    """g [B,C,2H,2W] -> parity planes of dL/dT [B,C,4,H+1,W+1] (+ asum [B,C] = sum gT*T when planes given)."""
    N.require_device(g, fir, planes)
    g = N.f32c(g)
    B, C, H2, W2 = g.shape
    H, W = H2 // 2, W2 // 2
    gt = torch.empty(B, C, 4, H + 1, W + 1, device=g.device, dtype=torch.float32)
    asum = torch.empty(B, C, device=g.device, dtype=torch.float32) if planes is not None else None
    N.call('sgdfr_blur_adjoint_f32', N.ptr(g), N.ptr(N.f32c(fir)), N.ptr(planes), N.ptr(gt), N.ptr(asum), B, C, H, W,
           N.stream())
    return gt, asum


def blur_adjoint_split(g, fir, planes=None, d=None, arith=None):
    """blur_adjoint + planes_to_split in one pass: g [B,C,2H,2W] -> (the int16 phase-major split form of gT*d that
    modconv_split(mode=DOWN3) stages, asum [B,C] = sum gT*T when the forward planes are given)."""
    arith = _SPLIT_ARITH[arith or config().precision]
    N.require_device(g, fir, planes, d)
    g = N.f32c(g)
    B, C, H2, W2 = g.shape
    H, W = H2 // 2, W2 // 2
    xs = torch.empty(B, 4 * C // 8, 2, (H + 1) * (W + 1), 8, device=g.device, dtype=torch.int16)
    asum = torch.empty(B, C, device=g.device, dtype=torch.float32) if planes is not None else None
    N.call('sgdfr_blur_adjoint_split_f32', N.ptr(g), N.ptr(N.f32c(fir)), N.ptr(planes), N.ptr(N.f32c(d)) if d is not None else None,
           N.ptr(xs), N.ptr(asum), B, C, H, W, arith, _sat(), N.stream())
    return xs, asum


def scale_reduce(gu, x, s):
    """dx = gu * s[b,c] (in place) and r[b,c] = sum_q x*gu; x may be a [1,C,H,W] broadcast constant."""
    N.require_device(gu, x, s)
    x = N.f32c(x)
    B, C, H, W = gu.shape
    xb = 0 if (x.shape[0] == 1 and B != 1) else C * H * W
    r = torch.empty(B, C, device=gu.device, dtype=torch.float32)
    N.call('sgdfr_scale_reduce_f32', N.ptr(gu), N.ptr(x), xb, N.ptr(s), N.ptr(gu), N.ptr(r), B, C, H * W, N.stream())
    return gu, r


def torgb_bwd(x, g, w_rgb, s):
    """dx [B,Cin,H,W] and r [B,3,Cin] = sum_p x*g_j."""
    N.require_device(x, g, w_rgb, s)
    x, g = N.f32c(x), N.f32c(g)
    B, cin, H, W = x.shape
    dx = torch.empty_like(x)
    r = torch.empty(B, 3, cin, device=x.device, dtype=torch.float32)
    N.call('sgdfr_torgb_bwd_f32', N.ptr(x), N.ptr(g), N.ptr(N.f32c(w_rgb)), N.ptr(s), N.ptr(dx), N.ptr(r), B, cin, H, W,
           N.stream())
    return dx, r


def grad_join(out, gu=None, s_next=None, g_rgb=None, w_rgb=None, s_rgb=None, g_add=None, noise=None, noise_weight=None, bias=None,
              want_y=True, slope=0.2, gain=SQRT2, work=None):
    """One pass over a saved StyledConv activation `out` [B,C,H,W] in the frozen generator's backward (sgdfr_grad_join_f32): the
    gradient of `out` = gu*s_next (the conv that reads it) + the ToRGB term (g_rgb [B,3,H,W], w_rgb [3,C], s_rgb [B,C]) + g_add;
    returns (g_pre, sums [B,C,3], gmax int32 [B,C], r_next [B,C] | None, r_rgb [B,3,C] | None) -- act_grad_reduce's outputs for the
    layer that produced `out`, scale_reduce's r for the reading conv, torgb_bwd's r for the reading ToRGB.
    work: a ZEROED float32 tensor with at least B*C*8 elements to carve the reduction buffers from (None: allocated and zeroed here)."""
    N.require_device(out, gu, s_next, g_rgb, w_rgb, s_rgb, g_add, bias, noise_weight)
    out = N.f32c(out)
    B, C, H, W = out.shape
    nz, nzb = _noise_args(noise, B, H, W)
    g_pre = torch.empty_like(out)
    n = B * C
    zero = work is None
    if work is None:
        work = torch.empty(n * 8, device=out.device, dtype=torch.float32)
    sums, gmax = work[:n * 3].view(B, C, 3), work[n * 3:n * 4].view(torch.int32).view(B, C)
    r_next = work[n * 4:n * 5].view(B, C) if gu is not None else None
    r_rgb = work[n * 5:n * 8].view(B, 3, C) if g_rgb is not None else None
    N.call('sgdfr_grad_join_f32', N.ptr(out), N.ptr(N.f32c(gu)) if gu is not None else None, N.ptr(s_next),
           N.ptr(N.f32c(g_rgb)) if g_rgb is not None else None, N.ptr(N.f32c(w_rgb)) if w_rgb is not None else None, N.ptr(s_rgb),
           N.ptr(N.f32c(g_add)) if g_add is not None else None, N.ptr(nz), nzb, N.ptr(noise_weight) if nz is not None else None,
           N.ptr(bias), N.ptr(g_pre), N.ptr(sums), N.ptr(gmax), N.ptr(r_next), N.ptr(r_rgb), B, C, H * W, float(slope), float(gain),
           int(bool(want_y)), int(zero), N.stream())
    return g_pre, sums, gmax, r_next, r_rgb


def styles_batched_bwd(entries, B, L, D, latent=None, want_latent=True):
    """dL/dlatent [B, L, D] of functional.styles_batched, two launches (sgdfr_styles_batched_bwd_f32).
    entries: one dict per layer with latent_index, mod_w [cin,D], and either gs [B,cin] (+ for demodulated convs a = d*dL/dd
    ([B,cout] view, any element stride), d, s, qt) or rgb_r [B,3,cin] + rgb_w [3,cin].  Entries with want_w / want_b set get
    'gmod_w' [cin,D] / 'gmod_b' [cin] (the modulation's parameter gradients, a third launch) written into the dict; needs `latent`.
    Returns glat (None when want_latent is False)."""
    if len(entries) > N.MAX_STYLE_LAYERS:
        raise RuntimeError('too many modulated layers (%d)' % len(entries))
    arr = (N.StyleGradLayer * len(entries))()
    dev = entries[0]['mod_w'].device
    ds = torch.empty(B * sum(e['mod_w'].shape[0] for e in entries), device=dev, dtype=torch.float32)
    keep, off = [ds], 0
    for i, e in enumerate(entries):
        c = arr[i]
        mw = N.f32c(e['mod_w'])
        cin = mw.shape[0]
        N.require_device(mw, e.get('gs'), e.get('rgb_r'), e.get('rgb_w'), e.get('a'), e.get('d'), e.get('s'), e.get('qt'))
        c.mod_w, c.cin, c.latent_index = mw.data_ptr(), cin, int(e['latent_index'])
        c.ds = ds[off:off + B * cin].data_ptr()
        off += B * cin
        c.gs = c.rgb_r = c.rgb_w = c.a = c.d = c.s = c.qt = c.gmod_w = c.gmod_b = None
        c.a_stride, c.cout = 1, 0
        keep.append(mw)
        if e.get('want_w'):
            e['gmod_w'] = torch.empty(cin, D, device=dev, dtype=torch.float32)
            c.gmod_w = e['gmod_w'].data_ptr()
        if e.get('want_b'):
            e['gmod_b'] = torch.empty(cin, device=dev, dtype=torch.float32)
            c.gmod_b = e['gmod_b'].data_ptr()
        if e.get('rgb_r') is not None:
            r, w = N.f32c(e['rgb_r']), N.f32c(e['rgb_w'])
            keep += [r, w]
            c.rgb_r, c.rgb_w = r.data_ptr(), w.data_ptr()
            continue
        gs = N.f32c(e['gs'])
        keep.append(gs)
        c.gs = gs.data_ptr()
        if e.get('a') is not None:
            a, d, s_, qt = e['a'], N.f32c(e['d']), N.f32c(e['s']), N.f32c(e['qt'])
            if a.dim() != 2 or a.stride(0) != a.shape[1] * a.stride(1):
                raise RuntimeError('styles_batched_bwd: a must be a [B,cout] view with a uniform element stride')
            keep += [a, d, s_, qt]
            c.a, c.a_stride, c.d, c.s, c.qt, c.cout = a.data_ptr(), a.stride(1), d.data_ptr(), s_.data_ptr(), qt.data_ptr(), d.shape[1]
    glat = torch.empty(B, L, D, device=dev, dtype=torch.float32) if want_latent else None
    if latent is not None:
        N.require_device(latent)
        latent = N.f32c(latent)
    N.call('sgdfr_styles_batched_bwd_f32', arr, len(entries), N.ptr(latent), N.ptr(glat), B, L, D, N.stream())
    return glat


def demod_dq(a, d, s):
    """dL/dQ [Cout,Cin] of the demodulation from a = d*dL/dd ([B,Cout] view, any element stride), d [B,Cout], s [B,Cin] (one launch)."""
    N.require_device(a, d, s)
    if a.dim() != 2 or a.stride(0) != a.shape[1] * a.stride(1):
        raise RuntimeError('demod_dq: a must be a [B,cout] view with a uniform element stride')
    B, cout = d.shape
    cin = s.shape[1]
    dq = torch.empty(cout, cin, device=d.device, dtype=torch.float32)
    N.call('sgdfr_demod_dq_f32', N.ptr(a), a.stride(1), N.ptr(N.f32c(d)), N.ptr(N.f32c(s)), N.ptr(dq), B, cin, cout, N.stream())
    return dq


def param_grads(entries, B):
    """The small parameter gradients of one backward in one launch (sgdfr_param_grads_f32).  entries: (kind, in, aux | None, C, HW);
    kind in N.PGRAD_*.  Returns one output tensor per entry ([C] bias, [1] noise strength, [3,C] ToRGB weight, [3] ToRGB bias)."""
    if len(entries) > N.MAX_PARAM_GRADS:
        raise RuntimeError('too many parameter-gradient entries (%d)' % len(entries))
    sizes = [{N.PGRAD_BIAS: C, N.PGRAD_NOISE: 1, N.PGRAD_RGB_W: 3 * C, N.PGRAD_RGB_B: 3}[kind] for kind, _, _, C, _ in entries]
    dev = entries[0][1].device
    flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)      # (zeroed: the ToRGB bias entries accumulate with atomics)
    arr = (N.ParamGrad * len(entries))()
    outs, off, keep = [], 0, []
    for i, (kind, inp, aux, C, HW) in enumerate(entries):
        N.require_device(inp, aux)
        if not inp.is_contiguous() or (aux is not None and not aux.is_contiguous()):
            raise RuntimeError('param_grads: inputs must be contiguous')
        c = arr[i]
        c.inp, c.aux, c.kind, c.C, c.HW = inp.data_ptr(), (aux.data_ptr() if aux is not None else None), int(kind), int(C), int(HW)
        c.scale = 1.0 / math.sqrt(C)
        o = flat[off:off + sizes[i]]
        c.out = o.data_ptr()
        off += sizes[i]
        outs.append(o.view(3, C) if kind == N.PGRAD_RGB_W else o)
        keep += [inp, aux]
    N.call('sgdfr_param_grads_f32', arr, len(entries), B, N.stream())
    return outs


def demod_grad(gd, d, qt, s, gs):
    """ds = gs + s * ((-gd * d^3) @ Q)."""
    N.require_device(gd, d, qt, s, gs)
    B, cin = s.shape
    ds = torch.empty_like(s)
    N.call('sgdfr_demod_grad_f32', N.ptr(N.f32c(gd)), N.ptr(d), N.ptr(qt), N.ptr(s), N.ptr(N.f32c(gs)), N.ptr(ds), B,
           cin, d.shape[1], N.stream())
    return ds


def config_wgrad_oik():
    """SGDFR_WGRAD_OIK=0 keeps the gather-form finish launch (read per call: same-process A/B)."""
    import os
    return os.environ.get('SGDFR_WGRAD_OIK', '1') != '0'


def wgrad(g, d, x, s, cout, upsample, wp=None, dq=None, weight=None, a=None):
    """dL/dW [1,Cout,Cin,3,3] of a modulated 3x3 conv.  g: activation gradient [B,Cout,H,W] (plain) or the
    parity planes of dL/dT [B,Cout,4,H+1,W+1] (upsample); x [B or 1,Cin,H,W]; dq [Cout,Cin] = dL/dQ or None.  weight: the layer's
    own [1,Cout,Cin,3,3] parameter -- with it the finish launch reads the demodulation term from there (contiguous) instead of
    gathering it from the packed copy `wp`; a ([B,Cout] view of d*dL/dd, instead of dq): that launch then forms dL/dQ itself
    (demod_dq's expression) where the shape takes the sliced form."""
    N.require_device(g, d, x, s, wp, dq, weight)
    x, g = N.f32c(x), N.f32c(g)
    B = s.shape[0]
    _, cin, H, W = x.shape
    xb = 0 if (x.shape[0] == 1 and B != 1) else cin * H * W
    mode = N.MODE_UP3 if upsample else N.MODE_PLAIN3
    dw = torch.empty(1, cout, cin, 3, 3, device=x.device, dtype=torch.float32)
    ks = _shape_query('sgdfr_modconv_wgrad_ksplit', B, cin, cout, H, W, mode)
    if ks > 0:      # slice sums in a partials buffer, added in fixed order by the finish launch (no atomics)
        part = torch.empty(ks, 9, cout, cin, device=x.device, dtype=torch.float32)
        N.call('sgdfr_modconv_wgrad_parts_f32', N.ptr(g), N.ptr(d), N.ptr(x), xb, N.ptr(s), N.ptr(part), B, cin, cout, H, W, mode,
               N.stream())
        if weight is not None and config_wgrad_oik():
            a_ptr, a_stride = (N.ptr(a), a.stride(1)) if (a is not None and dq is None) else (None, 1)
            N.call('sgdfr_modconv_wgrad_finish_parts_oik_f32', N.ptr(part), ks, N.ptr(N.f32c(weight.detach())), N.ptr(dq), a_ptr, a_stride,
                   N.ptr(d) if a_ptr is not None else None, N.ptr(s) if a_ptr is not None else None, B, N.ptr(dw), cout, cin, N.stream())
        else:
            if dq is None and a is not None:
                dq = demod_dq(a, d, s)
            N.call('sgdfr_modconv_wgrad_finish_parts_f32', N.ptr(part), ks, N.ptr(wp), N.ptr(dq), N.ptr(dw), cout, cin, N.stream())
        return dw
    if dq is None and a is not None:
        dq = demod_dq(a, d, s)
    dwp = torch.empty(cin, 9, cout, device=x.device, dtype=torch.float32)
    N.call('sgdfr_modconv_wgrad_f32', N.ptr(g), N.ptr(d), N.ptr(x), xb, N.ptr(s), N.ptr(dwp), B, cin, cout, H, W, mode, N.stream())
    N.call('sgdfr_modconv_wgrad_finish_f32', N.ptr(dwp), N.ptr(wp), N.ptr(dq), N.ptr(dw), cout, cin, N.stream())
    return dw
