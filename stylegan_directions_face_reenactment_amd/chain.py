"""The inference dataflow between split convs ("split chain"): which hand-over form each layer of the no-grad
forward takes, the StyledConv that runs it, and the two-stream pipeline of independent batches.  Launch wrappers live in
functional.py; nothing here touches the C ABI directly except shape queries.
"""
import torch

from . import _native as N
from .config import config
from .functional import (_shape_query, blur_bias_act, blur_bias_act_split, modconv_split, modconv_wsplit, split_ok, wsplit_ok)


class SplitAct:
    """An activation that only exists in the NEXT conv's split input form (x * s_next as 16-bit hi/lo pairs,
    [B, C/8, 2, H*W, 8] int16): written by the producing kernel's epilogue, staged by DMA in the consumer.  wino=True: the
    Winograd input form of that conv instead ([B, C/8, 4, 2, H*W/2, 8], see to_wsplit / modconv_wsplit)."""
    __slots__ = ('xs', 'shape', 'wino', 'arith')

    def __init__(self, xs, shape, wino=0, arith=None):
        self.xs, self.shape, self.wino = xs, tuple(shape), (2 if wino is True else int(wino or 0))
        self.arith = arith          # None: the ambient precision; 'fp16f8': F(4,3) form with fp8 cross terms (wsplit_chain_arith)


def xin_ok(B, cin, cout, H, W, mode=N.MODE_PLAIN3):
    """Can the split conv of this shape take its input as a SplitAct?"""
    return config().use_split_chain and split_ok(B, cin, cout, H, W, mode) and \
        bool(_shape_query('sgdfr_modconv2d_split_xin_supported', B, cin, cout, H, W, mode))


def styled_conv_split(x, wsp, s, d, cout, upsample=False, fir=None, noise=None, noise_weight=None, bias=None, batch=None,
                      s_next=None, rgb=None, want_y=True, wino_next=False, arith_next=None, xs_arith=None):
    """One StyledConv on the split kernels with the inference-only dataflow options: x may be a SplitAct (then `s` is
    already applied), s_next asks for the output as a SplitAct for the next conv, rgb for the fused ToRGB partial sums.
    A SplitAct in Winograd form (x.wino) runs on modconv_wsplit with `wsp` = the prepack_wsplit pack; wino_next (transposed
    conv + blur only) asks for the output in that form, arith_next ('fp16f8' | None) for the arithmetic of that hand-over;
    xs_arith ('fp16f8' | None, F(4,3) layers only) for the arithmetic of the plain split hand-over s_next.
    Returns (activation: fp32 tensor | SplitAct | None, ToRGB partials | None)."""
    if isinstance(x, SplitAct) and x.wino:
        if upsample:
            raise RuntimeError('styled_conv_split: the Winograd input form feeds plain convs only')
        B, cin, H, W = x.shape
        res = modconv_wsplit(x.xs, x.shape, wsp, d, cout, noise, noise_weight, bias, True, rgb=rgb,
                             want_y=want_y and s_next is None, s_next=s_next, f=x.wino, arith=x.arith,
                             xs_arith=xs_arith if s_next is not None else None)
        if s_next is not None:
            _, part, xs = res
            return SplitAct(xs, (B, cout, H, W), arith=xs_arith), part
        return res if rgb is not None else (res, None)
    x_arith = None
    if isinstance(x, SplitAct):
        B, cin, H, W = x.shape
        xin, x_split, s_arg = x.xs, x.shape, None
        batch = B
        x_arith = x.arith           # 'fp16f8': the producer wrote fp8 cross-term operands (where sgdfr_modconv2d_split_f8_ok; `wsp` packed alike)
    else:
        xin, x_split, s_arg = x, None, s
        B = s.shape[0] if batch is None else batch
        H, W = x.shape[2], x.shape[3]
    if not upsample:
        res = modconv_split(xin, wsp, s_arg, d, cout, noise, noise_weight, bias, True, batch=batch, rgb=rgb,
                            want_y=want_y and s_next is None, x_split=x_split, s_next=s_next, arith=x_arith)
        if s_next is not None:          # the activation leaves only as the next conv's split input
            _, part, xs = res
            return SplitAct(xs, (B, cout, H, W)), part
        return res if rgb is not None else (res, None)
    if s_next is not None and config().use_plane_padding and \
            _shape_query('sgdfr_modconv2d_split_ksplit_hint', B, x_split[1] if x_split else x.shape[1], cout, H, W, N.MODE_UP3) == 1:
        # parity planes padded to whole 128-byte lines: the odd-sized dense planes make every store run straddle two lines
        ps = ((H + 1) * (W + 1) + 31) // 32 * 32
        planes = modconv_split(xin, wsp, s_arg, d, cout, batch=batch, mode=N.MODE_UP3, x_split=x_split, plane_stride=ps, arith=x_arith)
        xs = blur_bias_act_split(planes, fir, H, W, s_next, noise, noise_weight, bias, True, plane_stride=ps, wino=wino_next,
                                 arith=arith_next if wino_next != 2 else None)
        return SplitAct(xs, (B, cout, 2 * H, 2 * W), wino_next, arith_next if wino_next != 2 else None), None
    planes = modconv_split(xin, wsp, s_arg, d, cout, batch=batch, mode=N.MODE_UP3, x_split=x_split, arith=x_arith)
    if s_next is not None:
        xs = blur_bias_act_split(planes, fir, H, W, s_next, noise, noise_weight, bias, True, wino=wino_next,
                                 arith=arith_next if wino_next != 2 else None)
        return SplitAct(xs, (B, cout, 2 * H, 2 * W), wino_next, arith_next if wino_next != 2 else None), None
    return blur_bias_act(planes, fir, H, W, noise, noise_weight, bias, True), None


def _interleaved_planes_ok(B, cin, H, W):
    """256-wide rows reach the Winograd hand-over only from INTERLEAVED padded parity planes (sgdfr_blur_bias_act_split_f32 REQUIREs
    them for an input W of 128): styled_conv_split takes that branch when plane padding is on, the library's SGDFR_PLANE_IL switch is
    not 0 and the producing transposed conv (?->cin at H/2 x W/2) is not K-sliced.  Otherwise the plain layer stays on the direct
    kernel (a cm=2 generator under use_plane_padding=False raised instead)."""
    import os
    if not config().use_plane_padding or os.environ.get('SGDFR_PLANE_IL', '1') == '0':
        return False
    # (the producer's Cin is not known here -- this generator's channel table steps by x1 or x2 between resolutions; the hint
    #  depends on the block count, and on Cin only through the cap of two channel blocks per slice)
    return all(_shape_query('sgdfr_modconv2d_split_ksplit_hint', B, c, cin, H // 2, W // 2, N.MODE_UP3) == 1 for c in (cin, 2 * cin))


def wsplit_chain_f(B, cin, cout, H, W):
    """Winograd form the inference chain runs this plain layer (fed by a transposed conv + blur) in: 0 (direct), 2 or 4 outputs
    per tile."""
    if not (config().use_wsplit and config().use_split_chain and config().wsplit_min_cin > 0 and cin >= config().wsplit_min_cin and W <= 256):
        return 0
    if W > 128 and not _interleaved_planes_ok(B, cin, H, W):
        return 0
    for f in ((4, 2) if config().wsplit_f == 4 else (2,)):
        # (F(2,3) hands over 8 bytes per element and saves a third of the MFMAs: it only pays from 256 input channels on;
        #  256-wide rows -- ffhq-256's 128 -> 128 @ 256^2 -- come from the blur's two column tiles: F(4,3) hand-over only, round 5)
        if wsplit_ok(B, cin, cout, H, W, f) and (f == 4 or cin >= max(config().wsplit_min_cin, 256)) and (W <= 128 or f == 4):
            return f
    return 0


def _f8_cross_on():
    """The opt-in fp8 cross terms apply: fp16x3 with the CALIBRATED range plan -- the fixed e4m3 exponents (WS_F8_XHI / WS_F8_XLO)
    assume the plan has put max|x*s| near 2^10; without it the cross terms would fall into e4m3 subnormals silently."""
    return config().precision == 'fp16x3' and config().cross_terms == 'fp8' and config().range_plan is True


def wsplit_chain_arith(B, cin, cout, H, W, f):
    """Arithmetic of a plain layer the chain runs in F(f,3) form: 'fp16f8' (fp16 main term + fp8 cross terms, Config.cross_terms) when
    its launch takes the wide-tile kernel anyway -- the only reader of that form -- else None (the ambient precision)."""
    if f == 4 and _f8_cross_on() and cin % 32 == 0 and \
            _shape_query('sgdfr_modconv2d_wsplit_wide', B, cin, cout, H, W):
        return 'fp16f8'
    return None


def xs_chain_arith(B, cin, cout, H, W, nxt_cout):
    """Arithmetic of the plain split hand-over from an F(4,3) layer (B, cin -> cout @ H x W) to the transposed conv cout -> nxt_cout
    that follows it: 'fp16f8' when Config.cross_terms says so and the consumer runs its deep plan (the only reader of the form; both
    F(4,3) kernels write it); else None."""
    if _f8_cross_on() and \
            _shape_query('sgdfr_modconv2d_split_f8_ok', B, cout, nxt_cout, H, W, N.MODE_UP3) and \
            _shape_query('sgdfr_modconv2d_split_ksplit_hint', B, cout, nxt_cout, H, W, N.MODE_UP3) == 1:
        return 'fp16f8'
    return None


def xs_plain_arith(B, cin, cout, H, W):
    """Arithmetic of the plain split hand-over from a transposed conv + blur to the DIRECT plain conv (B, cin -> cout @ H x W) after it:
    'fp16f8' when Config.cross_terms says so and that conv runs the plan that reads the form (sgdfr_modconv2d_split_f8_ok: the 4-wave
    plan of the 64 -> 64 @ 256^2 layer), without K slices; else None."""
    if _f8_cross_on() and \
            _shape_query('sgdfr_modconv2d_split_f8_ok', B, cin, cout, H, W, N.MODE_PLAIN3) and \
            _shape_query('sgdfr_modconv2d_split_ksplit_hint', B, cin, cout, H, W, N.MODE_PLAIN3) == 1:
        return 'fp16f8'
    return None


def wsplit_chain_ok(B, cin, cout, H, W):
    return wsplit_chain_f(B, cin, cout, H, W) != 0


def rgb_fusable(B, cin, cout, H, W):
    """True when the plain 3x3 conv of this shape runs on the split kernel in one pass, so the ToRGB that follows it can be
    accumulated in its epilogue instead of re-reading the activation."""
    return config().use_rgb_fusion and split_ok(B, cin, cout, H, W) and \
        (not config().use_splitk or _shape_query('sgdfr_modconv2d_split_ksplit_hint', B, cin, cout, H, W, N.MODE_PLAIN3) == 1)


class StreamPipeline:
    """Consecutive INDEPENDENT batches on alternating HIP streams.  The head of a generator forward (4x4 ... 16x16 layers:
    K-sliced launches that under-fill the chip, ~30 dependent launches with ~6 us of gap each) then runs beside the big layers
    of the previous batch instead of in front of its own: 9.69 k -> 10.08 k frames/s at B=64 with two streams (three: 9.91 k),
    bit-identical images (scripts/two_stream_probe.py).

        pipe = StreamPipeline(2)
        for w in batches:
            with pipe.next():                 # the slot's stream first waits for what the caller's stream has queued so far
                img, _ = G([w], input_is_latent=True, verify_range=False)      # (the default verifies: it would WAIT per call
                out.append(img); tokens.append(G.take_range_token())           #  and serialise the two streams)
        pipe.join(*out)                       # the caller's stream now waits for every slot; tensors are handed over to it
        bad = [i for i, t in enumerate(tokens) if t is not None and not G.range_ok(t)]      # re-render those (ReenactmentSession does)
    """

    def __init__(self, n=2, device=None):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, int(n)))]
        self._i = 0
        self.last = None                      # stream of the latest slot

    def next(self):
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream(s.device))
        self.last = s
        return torch.cuda.stream(s)

    def join(self, *tensors, stream=None):
        """The current stream waits for `stream` (default: every slot); `tensors` (made on slot streams) may then be used on it."""
        cur = torch.cuda.current_stream(self.streams[0].device)
        for s in ([stream] if stream is not None else self.streams):
            cur.wait_stream(s)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
