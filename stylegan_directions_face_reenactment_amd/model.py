"""StyleGAN2 generator on MI355X: drop-in for the reference's ``libs/gan/StyleGAN2/model.py`` generator half.

Kept from the reference (it is the compatibility contract): class names, constructor
signatures, ``forward`` keyword arguments and return tuple, attribute names (``style``, ``input.input``,
``conv1``, ``to_rgb1``, ``convs``, ``to_rgbs``, ``noises.noise_i``, ``n_latent``, ``num_layers``,
``log_size``, ``size``, ``style_dim``, ``channels``) and the exact ``state_dict`` key set / shapes
(model.py:362-447), so ``load_state_dict(ckpt['g_ema'])``, ``copy.deepcopy`` and
``convs[i].parameters()`` behave as before.

Different by design: no per-sample weight tensor is ever built.  Each modulated conv is one
shared-weight fp32-MFMA launch with style scaling folded into input staging and demodulation, noise,
bias and leaky-ReLU folded into its epilogue (csrc/modconv.hip); weights are re-packed once per weight
version and cached outside the state_dict.  All arithmetic runs in the HIP library -- modules raise on
CPU tensors instead of falling back.

The discriminator-side classes of the reference file (model.py:542-709: ConvLayer, ResBlock,
Discriminator, Encoder, ...) are never instantiated by the reenactment scripts and are out of scope.
"""
import math
import os
import random
import struct
import warnings

import torch
from torch import nn

from . import autograd as AG
from . import functional as F_
from .graph_runner import GraphReplayMixin
from .planner import PlannerMixin
from .range_plan import RangePlanMixin, RangeToken  # noqa: F401  (RangeToken re-exported: callers hold tokens)
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401  (re-exported like model.py:8)


def make_kernel(k):
    """Normalised 2-D FIR from 1-D taps (model.py:19-27)."""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


_SIDE_STREAMS = {}


def _side_stream(device):
    """One side HIP stream per device (module-level: streams must not live on nn.Modules that get deep-copied)."""
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class PixelNorm(nn.Module):
    def forward(self, input):
        return AG.PixelNormFn.apply(input) if _needs_grad(input) else F_.pixel_norm(input)


class Upsample(nn.Module):
    """2x FIR upsampling of the RGB skip (model.py:30-48).  Inside ToRGB the same taps are applied by
    the fused kernel; this standalone module goes through ``upfirdn2d``."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    """FIR blur (model.py:72-88)."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer('kernel', kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    """Equalised-lr linear layer (model.py:129-162); also imported by the e4e encoder
    (libs/gan/encoder4editing/models/encoders/psp_encoders.py:9)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        if _needs_grad(input, self.weight, self.bias):
            return AG.EqualLinearFn.apply(input, self.weight, self.bias, self.scale, self.lr_mul,
                                          bool(self.activation))
        return F_.linear(input, self.weight, self.bias, wscale=self.scale, bscale=self.lr_mul,
                         lrelu=bool(self.activation))

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.weight.shape[1], self.weight.shape[0])


class ModulatedConv2d(nn.Module):
    """Modulated / demodulated convolution (model.py:177-273) in shared-weight form.

    3x3 (plain or 2x upsampling) runs on the MFMA kernel; 1x1 without demodulation is the ToRGB
    kernel.  ``downsample`` exists only on the discriminator side of the reference and is refused."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError('downsample modulated conv is not on the generator path')
        if kernel_size not in (1, 3):
            raise NotImplementedError('kernel_size %d: the generator only uses 3x3 and 1x1' % kernel_size)
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
            if kernel_size != 3 or tuple(self.blur.kernel.shape) != (4, 4) or self.blur.pad != (1, 1):
                raise NotImplementedError('upsampling modconv is built for 3x3 weights and a 4-tap blur')
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._pack = None          # (key, wp, q, qt): re-packed weights, not part of the state_dict
        self._pack_t = None        # (key, wt): adjoint pack for backward
        self._pack_w = None        # [key, U, U_adjoint]: Winograd-domain packs

    def __repr__(self):
        return '{}({}, {}, {}, upsample={}, downsample={})'.format(
            self.__class__.__name__, self.in_channel, self.out_channel, self.kernel_size, self.upsample,
            self.downsample)

    def _key(self):
        w = self.weight
        return (w.data_ptr(), w._version, w.device)

    def invalidate_packs(self):
        """Drop every cached re-pack of `weight`.  The caches are keyed on (storage, tensor version, device), which sees
        optimizer steps, `copy_`, `load_state_dict` and `.cuda()`, but NOT in-place writes through `.data`
        (`w.data.mul_()`, `w.data.copy_()`, the EMA `accumulate()` idiom): those do not bump the version counter, so
        call this (or `Generator.invalidate_packs()`) after them."""
        self._pack = self._pack_t = self._pack_w = None
        self._pack_s = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packs()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_packs()

    def packed(self):
        """([Cin, k*k, Cout] scaled weights, Q[o,i] = sum_taps (scale W)^2, Q^T), rebuilt when the parameter's
        storage or version changes (optimizer step, load_state_dict, .cuda())."""
        key = self._key()
        if self._pack is None or self._pack[0] != key:
            with torch.no_grad():
                self._pack = (key,) + F_.prepack(self.weight.detach())
            self._pack_t = None
        return self._pack[1], self._pack[2], self._pack[3]

    def packed_t(self):
        """[Cout, 9, Cin] weight pack of the adjoint conv (dL/dx): taps rotated for the plain conv, as-is for the
        stride-2 transposed one.  Built lazily, only when a backward pass needs it."""
        key = self._key()
        if getattr(self, '_pack_t', None) is None or self._pack_t[0] != key:
            with torch.no_grad():
                self._pack_t = (key, F_.prepack_t(self.weight.detach(), flip=not self.upsample))
        return self._pack_t[1]

    def packed_wino(self, adjoint=False):
        """Winograd-domain weights of the plain 3x3 conv (adjoint=True: of its dL/dx conv), cached per weight version."""
        key = self._key()
        cache = getattr(self, '_pack_w', None)
        if cache is None or cache[0] != key:
            cache = self._pack_w = [key, None, None]
        if cache[1 + int(adjoint)] is None:
            with torch.no_grad():
                cache[1 + int(adjoint)] = F_.prepack_wino(self.weight.detach(), adjoint=adjoint)
        return cache[1 + int(adjoint)]

    def packed_split(self, adjoint=False, arith=None):
        """16-bit hi/lo weight pack of the split precision modes (default arith = functional.PRECISION), cached per weight
        version (adjoint=True: the pack of the plain conv's dL/dx conv; 'down': of the transposed conv's, mode DOWN3)."""
        arith = arith or F_.config().precision
        key = self._key()
        cache = getattr(self, '_pack_s', None)
        if cache is None or cache[0] != key:
            cache = self._pack_s = [key, {}]
        slot = (arith, adjoint if adjoint == 'down' else bool(adjoint))
        if slot not in cache[1]:
            with torch.no_grad():
                cache[1][slot] = F_.prepack_split(self.weight.detach(), arith=arith, adjoint=adjoint)
        return cache[1][slot]

    def packed_wsplit(self, arith=None, f=2):
        """Weight pack of the 1-D Winograd form F(f,3) of the plain split conv (functional.prepack_wsplit), cached like packed_split."""
        arith = arith or F_.config().precision
        key = self._key()
        cache = getattr(self, '_pack_s', None)
        if cache is None or cache[0] != key:
            cache = self._pack_s = [key, {}]
        slot = (arith, 'wino', f)
        if slot not in cache[1]:
            with torch.no_grad():
                cache[1][slot] = F_.prepack_wsplit(self.weight.detach(), arith=arith, f=f)
        return cache[1][slot]

    def style_spec(self, latent_index):
        """(latent row, modulation weight, bias, Q or None, Cout) for functional.styles_batched."""
        q = self.packed()[1] if (self.kernel_size == 3 and self.demodulate) else None
        return (latent_index, self.modulation.weight, self.modulation.bias, q, self.out_channel)

    def styles(self, style):
        """(s, d) for one layer; differentiable when anything upstream needs gradients."""
        mod = self.modulation
        q = qt = None
        if self.kernel_size == 3 and self.demodulate:
            _, q, qt = self.packed()
        if _needs_grad(style, mod.weight, mod.bias):
            out = AG.StyleFn.apply(style, mod.weight, mod.bias, q, qt, self.out_channel)
            return out if q is not None else (out, None)
        return F_.style_demod(style, mod.weight, mod.bias, q, self.out_channel)

    def fused(self, input, style, noise=None, noise_weight=None, bias=None, activate=False, batch=None, sd=None, rgb=None,
              want_y=True, ranged=False):
        """conv (+ noise + bias + leaky-ReLU) in one pass; what StyledConv.forward calls.
        `sd` = precomputed (s, d), e.g. from the generator's batched style launch; ranged=True says that pair already
        carries the fp16-split range plan (functional.styles_batched(plans=...)), otherwise the split path plans from the
        true max |x| of each image."""
        if self.kernel_size != 3:
            raise NotImplementedError('only the 3x3 modulated conv has a fused StyledConv form (1x1 lives in ToRGB)')
        s, d = self.styles(style) if sd is None else sd
        if _needs_grad(input, s, d, self.weight, noise_weight, bias):
            return AG.StyledConvFn.apply(input, s, d, self.weight, noise_weight, bias, noise, self, activate, batch)
        return F_.modconv3x3(input, self.packed()[0], s, d, self.out_channel, upsample=self.upsample,
                             fir=self.blur.kernel if self.upsample else None, noise=noise,
                             noise_weight=noise_weight, bias=bias, activate=activate, batch=batch,
                             wino=None if self.upsample else self.packed_wino,
                             split=self.packed_split, rgb=rgb, want_y=want_y, ranged=ranged)

    def forward(self, input, style):
        if self.kernel_size == 1:
            if self.demodulate or self.out_channel != 3:
                raise NotImplementedError('1x1 modulated conv is only built as ToRGB (3 outputs, no demodulation)')
            s, _ = self.styles(style)
            if _needs_grad(input, s, self.weight):
                return AG.ToRGBFn.apply(input, s, self.weight, None, None, None)
            return F_.torgb(input, self.weight.view(3, self.in_channel), s)
        return self.fused(input, style)


class NoiseInjection(nn.Module):
    """image + weight * noise (model.py:276-287).  Inside StyledConv this is part of the conv epilogue;
    the standalone module is kept for API parity."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """modconv -> noise -> bias + leaky-ReLU (model.py:303-337), as ONE fused launch
    (two for the upsampling variant: MFMA transposed conv, then FIR + epilogue)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, batch=None, sd=None, rgb=None, want_y=True, ranged=False):
        """rgb = (w_rgb [3,C], s_rgb [B,C]) (no-grad path): also returns the ToRGB partial sums, see functional.rgb_fusable;
        want_y=False then skips storing the activation itself (last layer: nothing else reads it)."""
        if noise is None:   # fresh per-sample noise, model.py:283-285
            B = style.shape[0] if sd is None else sd[0].shape[0]
            r = input.shape[-1] * (2 if self.conv.upsample else 1)
            noise = torch.empty(B, 1, r, r, device=input.device, dtype=torch.float32).normal_()
        return self.conv.fused(input, style, noise=noise, noise_weight=self.noise.weight, bias=self.activate.bias,
                               activate=True, batch=batch, sd=sd, rgb=rgb, want_y=want_y, ranged=ranged)


class ToRGB(nn.Module):
    """model.py:340-359: 1x1 modconv (no demod) + bias + FIR-upsampled skip, one HBM-bound launch."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, sd=None):
        conv = self.conv
        s = sd[0] if sd is not None else conv.styles(style)[0]
        fir = None
        if skip is not None:
            up = getattr(self, 'upsample', None)
            if up is None or tuple(up.kernel.shape) != (4, 4) or up.pad != (2, 1):
                raise NotImplementedError('ToRGB skip path is built for the 4-tap 2x Upsample')
            fir = up.kernel
        if _needs_grad(input, s, skip, conv.weight, self.bias):
            return AG.ToRGBFn.apply(input, s, conv.weight, self.bias, skip, fir)
        return F_.torgb(input, conv.weight.view(3, conv.in_channel), s, bias=self.bias.view(3), skip=skip, fir=fir)

    def finish(self, part, skip=None, u8=None):
        """The rest of forward() when the 1x1 conv was accumulated in the feeding conv's epilogue (no-grad path):
        sum of the per-cout-tile partials + bias + upsampled skip (u8: as uint8 HWC, functional.U8Target)."""
        fir = None
        if skip is not None:
            up = getattr(self, 'upsample', None)
            if up is None or tuple(up.kernel.shape) != (4, 4) or up.pad != (2, 1):
                raise NotImplementedError('ToRGB skip path is built for the 4-tap 2x Upsample')
            fir = up.kernel
        return F_.torgb_finish(part, bias=self.bias.view(3), skip=skip, fir=fir, u8=u8)


class Generator(RangePlanMixin, GraphReplayMixin, PlannerMixin, nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        self.channel_multiplier = channel_multiplier
        self.style = nn.Sequential(
            PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu')
                           for _ in range(n_mlp)])
        self.channels = {
            4: 512, 8: 512, 16: 512, 32: 512,
            64: 256 * channel_multiplier, 128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
            512: 32 * channel_multiplier, 1024: 16 * channel_multiplier,
        }
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer('noise_{}'.format(layer_idx), torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True,
                                         blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self.overlap_rgb = True     # run the ToRGB chain on a side stream in no-grad forwards
        # functional.Config this generator runs under (arithmetic, range plan, chain / Winograd / fusion switches); None = whatever
        # is in force where it is called (functional.config(): a `using` block, else the process default)
        self.config = None

    def invalidate_packs(self):
        """Forget every cached weight re-pack and launch plan.  Needed only after in-place edits through `.data`
        (weight blending, EMA, `p.data.copy_()`), which PyTorch's version counter does not see, or after replacing a
        Parameter object by assignment; optimizer steps, `load_state_dict` (also `assign=True`), `.to()` / `.cuda()` are
        tracked or invalidate on their own."""
        for m in self.modules():
            if isinstance(m, ModulatedConv2d):
                m.invalidate_packs()
        self._chain_plans = {}
        self.__dict__.pop('_wino_cache', None)
        self._range_state = None
        self._plist = None
        self._stamp_list = None
        self._drop_graphs()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)       # .to() / .cuda() / .float(): parameters may be new objects
        self._chain_plans, self._range_state, self._plist, self._stamp_list = {}, None, None, None
        self._drop_graphs()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)  # called once per load_state_dict (assign=True swaps the Parameters)
        self._chain_plans, self._range_state, self._plist, self._stamp_list = {}, None, None, None
        self._drop_graphs()

    # ---- latent-side helpers (model.py:449-469)
    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)


    def _params(self):
        """The parameter list, walked once and cached (nn.Module.parameters() costs ~0.15 ms per call on this module tree:
        a seventh of a small-batch forward's host time); invalidate_packs() / .to() / load_state_dict drop it."""
        plist = getattr(self, '_plist', None)
        if plist is None:
            plist = self._plist = list(self.parameters())
        return plist

    def _weights_stamp(self):
        """Cheap identity of the current weights: storage of one weight + the sum of the version counters of all parameters
        and of the fixed noise maps (they enter the activation ranges too)."""
        tl = getattr(self, '_stamp_list', None)
        if tl is None:
            tl = self._stamp_list = self._params() + list(self.noises.buffers())
        w = tl[0]
        v = 0
        for p in tl:
            v += p._version
        return (w.data_ptr(), w.device, v)

    def __deepcopy__(self, memo):
        # (optimization.py:28 deep-copies G) the copy gets its own word and no tokens of the original
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        skip = ('_sat', '_sat_seen', '_sat_tokens', '_graphs', '_graph_calls', '_last_token', '_plist', '_stamp_list', '_all_mods')
        for k, v in self.__dict__.items():
            if k not in skip:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ('_sat', '_sat_seen', '_sat_tokens', '_graphs', '_graph_calls', '_last_token', '_plist', '_stamp_list', '_all_mods'):
            st.pop(k, None)
        return st

    # ---- the path itself (model.py:471-539)
    # A raw `G([w])` -- what the reference's scripts call (run_inference.py:125, utils_inference.py:88, optimization.py:50,
    # invert_images.py:103, extract_statistics.py:85) -- hands back VERIFIED frames: the forward is awaited and, had any fp16
    # operand left the range plan, re-rendered in bf16x3 first.  Throughput callers that check tokens themselves
    # (ReenactmentSession, functional.StreamPipeline loops, bench.py) pass verify_range=False; `G.verify_range_default = False`
    # or SGDFR_VERIFY_RANGE=0 changes the default for a generator / the process.
    verify_range_default = os.environ.get('SGDFR_VERIFY_RANGE', '1') != '0'
    # under autograd: the whole-synthesis Function (False / SGDFR_FUSED_BACKWARD=0: one Function per layer)
    fused_backward = os.environ.get('SGDFR_FUSED_BACKWARD', '1') != '0'

    def forward(self, styles, return_latents=False, return_features=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=False, image_out=None,
                verify_range=None, graph=None):
        """Reference signature (model.py:471-482) plus optional extensions (no-grad forwards only): image_out / verify_range
        (see _forward_impl; None = this generator's default, which is to verify) and graph (None: the default policy below,
        True: replay whenever possible, False: always eager).

        hipGraph replay.  A no-grad forward is ~65 dependent launches = ~1 ms of Python at any batch size.  From the third
        forward of one signature (input shape, flags, arithmetic) on, the launch sequence is captured once and REPLAYED: the
        latent (and truncation latent) are copied into the graph's static inputs, one graph launch runs the identical kernels
        on the identical arguments (bit-identical images), and the outputs are cloned out of the graph's static buffers.
        Default policy (measured, scripts/small_batch_time.py): a replay costs the device ~80 us more than the eager launches
        (B=8: 1.34 -> 1.43 ms), so it pays where the HOST is the bound -- small batches (B=1: 917 -> 1462 frames/s, B=4 +3 %),
        i.e. batch * (size/256)^2 <= GRAPH_MAX_WORK -- and for verified forwards, whose wait exposes the enqueue time at every
        batch size (generate_image at B=32: 7.8 k -> 8.3 k frames/s) -- verify_range=True passed explicitly; a raw `G([w])`, verified
        only by default, is replayed when it is host-bound or when the capture's pinned intermediates (~4.5 GB at B=64) are a
        small share of the device's memory (graph_runner._capture_fits: <= 3 % of total, <= 25 % of what is free), and runs
        eagerly otherwise (graph_runner.MAX_BIG_GRAPHS).  Weight changes (tracked like the weight packs; after
        `.data` edits call invalidate_packs()), a change of arithmetic / range plan, hooks, style mixing, caller-supplied
        noise and randomize_noise run eagerly.  Switch off: `G.use_graphs = False` or SGDFR_GRAPHS=0."""
        cfg = getattr(self, 'config', None)
        if cfg is not None and cfg is not F_.config():      # this generator's own configuration: forward (and, through the autograd
            with F_.using(cfg):                             # Functions, backward) run under it whatever the caller's ambient one is
                return self.forward(styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                    input_is_latent, noise, randomize_noise, image_out, verify_range, graph)
        verify_explicit = verify_range is not None
        if verify_range is None:
            verify_range = bool(self.verify_range_default)
        key = self._graph_key(styles, return_latents, inject_index, truncation, truncation_latent, input_is_latent, noise,
                              randomize_noise, image_out, verify_range, graph, verify_explicit)
        if key is None:
            return self._forward_impl(styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                      input_is_latent, noise, randomize_noise, image_out, verify_range)
        return self._replay_or_run(key, styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                   input_is_latent, image_out, verify_range)

    def _forward_impl(self, styles, return_latents=False, return_features=False, inject_index=None, truncation=1,
                      truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=False, image_out=None,
                      verify_range=False, depth=0):
        """Reference signature (model.py:471-482) plus two optional extensions (no-grad forwards only):
        image_out = functional.U8Target returns the image as uint8 HWC frames (the reference's tensor_to_image scaling), written
        by the last ToRGB launch itself when that ToRGB is fused into its conv (otherwise converted by one extra launch);
        verify_range=True waits for this forward and, if any fp16 operand left the range plan, re-renders the batch in the
        bf16x3 arithmetic before returning (what generate_image and ReenactmentSession use: a clamped frame is never handed
        back).  Without it the forward returns at once and leaves a RangeToken (take_range_token / range_ok); unchecked
        tokens are polled without blocking by the following forwards."""
        styles_in = styles                             # (a clamped batch is rendered a second time from the caller's styles)
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else \
                [getattr(self.noises, 'noise_{}'.format(i)) for i in range(self.num_layers)]
            if randomize_noise and verify_range and not torch.is_grad_enabled():
                # a verified forward may have to render the batch a second time: draw the per-image noise HERE (what NoiseInjection
                # would draw, model.py:296-298) so that the second pass renders the same image
                nb = styles[0].shape[0]
                noise = [torch.randn(nb, 1, 2 ** (2 + (i + 1) // 2), 2 ** (2 + (i + 1) // 2), device=styles[0].device)
                         for i in range(self.num_layers)]
        trunc = truncation_latent if truncation < 1 else None
        if truncation < 1 and truncation_latent is None:
            raise RuntimeError('truncation < 1 needs truncation_latent')
        grad = _needs_grad(*styles, trunc, *self._params())

        def prepare(w, rows):
            if _needs_grad(w, trunc):
                return AG.LatentPrepareFn.apply(w, None, trunc, rows, 0, truncation)
            return F_.latent_prepare(w, rows, trunc=trunc, psi=truncation)

        if len(styles) < 2:
            # truncation (also of a full W+ code, as in the reference) + W -> W+ broadcast in one launch
            latent = prepare(styles[0], self.n_latent)
        else:   # style mixing (model.py:510-517); unused by the reenactment scripts
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([prepare(styles[0], inject_index), prepare(styles[1], self.n_latent - inject_index)], 1)

        order = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)]
        i = 1
        for conv1, conv2, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            order += [(conv1.conv, i), (conv2.conv, i + 1), (to_rgb.conv, i + 2)]
            i += 2
        layers = [self.conv1] + list(self.convs)           # StyledConvs in execution order: plain, (up, plain) x n
        to_rgbs = [self.to_rgb1] + list(self.to_rgbs)      # to_rgbs[k] follows layers[2k]
        self.__dict__['_last_token'] = None
        with F_.saturation_sink(self._sat_word()):         # every split launch below (and its backward) counts into OUR word
            if grad:
                frozen = not any(p.requires_grad for p in self._params())
                hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in layers + to_rgbs)
                if self.fused_backward and not hooked and not return_features and image_out is None:
                    # ONE Function for the whole synthesis network -- its backward walks every saved activation once and batches
                    # the per-layer glue (autograd.SynthesisFn): the direction trainer (frozen generator) and PTI (trained one)
                    image = AG.SynthesisFn.apply(latent, self, order, layers, to_rgbs, noise, *AG.synthesis_params(self, layers, to_rgbs))
                    return (image, latent) if return_latents else (image, None)
                if frozen:
                    # ... per-layer Functions (hooked modules): the two batched style launches, differentiable w.r.t. the latent only
                    flat = iter(AG.StylesBatchedFn.apply(latent, order))
                    sd = [(next(flat), next(flat) if (m.kernel_size == 3 and m.demodulate) else None) for m, _ in order]
                else:   # differentiable per-layer modulation (autograd routes dL/ds back into the latent rows and the weights)
                    sd = [m.styles(latent[:, li]) for m, li in order]
                # (the autograd forward plans every conv from the true max |x| of each image: it cannot saturate on finite data)
                return self._synthesis(latent, sd, layers, to_rgbs, noise, True, False, return_latents, image_out)
            # no-grad: every layer's s = A_l(w_l) and demodulation d_l in two launches (instead of 33)
            specs = [m.style_spec(li) for m, li in order]
            plans, arith = None, None                      # arith: this generator's fallback arithmetic, when it has one
            if F_.config().precision == 'fp16x3' and F_.config().range_plan is True:
                plans, arith = self._range_plans(latent, noise, specs, order, layers)
            if arith is not None and arith != F_.config().precision:    # saturation / non-finite fallback: run this forward in `arith`
                with F_.precision(arith):
                    return self._synthesis(latent, F_.styles_batched(latent, specs), layers, to_rgbs, noise, False, False,
                                           return_latents, image_out)
            out = self._synthesis(latent, F_.styles_batched(latent, specs, plans), layers, to_rgbs, noise, False,
                                  plans is not None, return_latents, image_out)
            if plans is None:
                return out
            self._settle_oldest_if_full()
            tok = self._snapshot()
            if not verify_range or tok is None:
                self.__dict__['_last_token'] = tok
                return out
            if self.range_ok(tok):
                return out
            # This batch clamped operands: it is rendered again before anything is handed back.  range_ok() has switched the
            # generator to bf16x3 and -- while the budget of automatic widenings lasts -- asked for a recalibration: the second
            # pass below measures THIS batch, widens the plan and renders in fp16x3 again (verified like the first); without
            # budget it renders in bf16 terms (fp32 exponent range, nothing to verify).
            self.__dict__['_rerendered'] = self.__dict__.get('_rerendered', 0) + 1
            # (rendered again from the caller's styles with THIS pass's resolved inject_index and noise tensors -- the same image,
            #  not a new random draw; a token that is merely suspect / stale asked for no recalibration: straight to bf16 terms)
            if depth <= self.AUTO_RECALIBRATIONS and (getattr(self, '_range_state', None) or {}).get('recal'):
                return self._forward_impl(styles_in, return_latents, return_features, inject_index, truncation, truncation_latent,
                                          input_is_latent, noise, False, image_out, True, depth + 1)
            with F_.precision('bf16x3'):
                return self._synthesis(latent, F_.styles_batched(latent, specs), layers, to_rgbs, noise, False, False,
                                       return_latents, image_out)

    def _synthesis(self, latent, sd, layers, to_rgbs, noise, grad, ranged, return_latents, image_out=None):
        """conv1 ... convs / to_rgbs with the (s, d) pairs of `sd` (one per entry of conv1, to_rgb1, (up, plain, to_rgb)*)."""
        batch = latent.shape[0]
        sd_of_layer = [0] + [2 + 3 * (i // 2) + (i % 2) for i in range(len(self.convs))]
        sd_of_rgb = [1] + [4 + 3 * k for k in range(len(self.to_rgbs))]
        # inference on the split kernels: layers are launched through functional.styled_conv_split, not through their
        # modules -- unless somebody hooked a layer's forward (per-layer probes), then every module really runs
        hooked = any(m._forward_hooks or m._forward_pre_hooks for m in layers + to_rgbs)
        chain = (not grad) and (not hooked) and F_.config().precision in ('fp16x3', 'bf16x3') and \
            not (F_.config().precision == 'fp16x3' and F_.config().range_plan == 'exact')       # 'exact': fp32 hand-over, max |x| measured per layer

        # The RGB branch: a fused ToRGB (partial sums from the conv epilogue) is finished by a small launch on the main
        # stream; an unfused one (HBM-bound kernel re-reading the activation) runs on a side HIP stream next to the conv chain.
        side = _side_stream(latent.device) if (self.overlap_rgb and not grad) else None
        main = torch.cuda.current_stream() if side is not None else None
        on_side = [False]                          # is the latest `skip` being produced on the side stream?

        if image_out is not None and grad:
            raise RuntimeError('image_out (uint8 frames) is an inference output: call the generator under torch.no_grad()')
        wrote_u8 = [False]

        def rgb(layer, x, part_in, skip_in, sdl, last=False):
            if part_in is not None:
                if on_side[0]:
                    main.wait_stream(side)
                    skip_in.record_stream(main)
                    on_side[0] = False
                if last and image_out is not None:
                    wrote_u8[0] = True
                    return layer.finish(part_in, skip_in, u8=image_out)
                return layer.finish(part_in, skip_in)
            run = lambda: layer(x, None, skip_in, sd=sdl)
            if side is None:
                return run()
            on_side[0] = True
            side.wait_stream(main)                 # x (and this forward's styles) are ready
            x.record_stream(side)
            if skip_in is not None:                # may come from a main-stream finish(): keep its block until the side kernel read it
                skip_in.record_stream(side)
            with torch.cuda.stream(side):
                return run()

        plan = self._chain_plan(batch, chain, noise, layers)

        # ConstantInput is broadcast inside the kernel (batch stride 0) instead of repeated
        x, skip = self.input.input, None
        for li, layer in enumerate(layers):
            c = layer.conv
            up = c.upsample
            sdl = sd[sd_of_layer[li]]
            nz = noise[li]
            first = li == 0
            use_chain, fuse, to_next, want_y, wino_next, arith_next, xs_arith = plan[li]
            k = li // 2
            if not use_chain:
                if isinstance(x, F_.SplitAct):
                    raise RuntimeError('internal: a split activation reached a layer that cannot take it')
                out, part = layer(x, None, noise=nz, batch=batch if first else None, sd=sdl, ranged=ranged), None
            else:
                rgb_arg = (to_rgbs[k].conv.weight.view(3, c.out_channel), sd[sd_of_rgb[k]][0]) if fuse else None
                wino_in = x.wino if isinstance(x, F_.SplitAct) else 0
                out, part = F_.styled_conv_split(
                    x, c.packed_wsplit(arith=x.arith, f=wino_in) if wino_in else c.packed_split(arith=getattr(x, 'arith', None)),
                    sdl[0], sdl[1], c.out_channel, upsample=up,
                    fir=c.blur.kernel if up else None, noise=nz, noise_weight=layer.noise.weight, bias=layer.activate.bias,
                    batch=batch if first else None, s_next=sd[sd_of_layer[li + 1]][0] if to_next else None, rgb=rgb_arg,
                    want_y=want_y, wino_next=wino_next, arith_next=arith_next, xs_arith=xs_arith)
            if not up:
                skip = rgb(to_rgbs[k], out, part, skip, sd[sd_of_rgb[k]], last=li == len(layers) - 1)
            x = out
        if side is not None and on_side[0]:
            main.wait_stream(side)
            skip.record_stream(main)
        image = skip
        if image_out is not None and not wrote_u8[0]:       # last ToRGB not fused (tiny batches / odd shapes): one extra launch
            from .reenact import images_to_uint8, grid_frames_uint8
            if image_out.frames is None and not image_out.swap_rb:
                image = images_to_uint8(image)
            else:
                B, _, H, W = image.shape
                frames = image_out.frames if image_out.frames is not None else torch.empty(B, H, W, 3, device=image.device, dtype=torch.uint8)
                K = frames.shape[2] // W
                grid_frames_uint8([image if k == image_out.panel else None for k in range(K)], swap_rb=image_out.swap_rb, out=frames)
                image = frames
        return (image, latent) if return_latents else (image, None)
