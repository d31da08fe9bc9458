"""e4e W+ producer: the step *before* the generator path in the inference flow.

`Encoder4Editing(50, 'ir_se', 256)` maps aligned faces `[B,3,256,256]` in [-1,1] to W+ codes `[B,14,512]`
(reference: libs/gan/encoder4editing/psp_encoders.py:122-199 on the IR-SE-50 trunk of helpers.py:25-112; built at
run_inference.py:74-79 and invert_images.py:63-70).  The state_dict key set is the reference's
(`input_layer.*`, `body.N.{shortcut_layer,res_layer}.*`, `styles.N.{convs.M,linear}.*`, `latlayer{1,2}.*`) so an
e4e checkpoint loads unchanged.

Convolutions run on MIOpen through PyTorch-ROCm (stock 3x3/1x1 convs; nothing here is a modulated conv); what is
specific to this build:

  * inference plan (`no_grad` + `eval()`): every conv->BatchNorm pair (second conv of each unit, shortcut
    projection, stem) is folded into one conv with bias, activations are kept channels-last for MIOpen's NHWC
    kernels, and the plan is cached until a parameter changes;
  * the 14 style heads are evaluated as 3 groups (coarse / middle / fine share their input feature map): the
    first conv of a group is one conv with the heads' filters concatenated, the following stride-2 convs (per-head
    filters on 8x8 ... 1x1 maps) are an unfold + one batched GEMM per depth, and the 14 `EqualLinear`s run on the HIP
    linear kernel (`sgdfr_linear_f32`);
  * like the reference (psp_encoders.py:185-199) no `latent_avg` is added; the generator applies truncation.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .model import EqualLinear

# (depth, units) of the four trunk stages; the first unit of a stage has stride 2 (helpers.py:29-36, num_layers=50)
_TRUNK = {50: ((64, 3), (128, 4), (256, 14), (512, 3)),
          100: ((64, 3), (128, 13), (256, 30), (512, 3)),
          152: ((64, 3), (128, 8), (256, 36), (512, 3))}


class SEModule(nn.Module):
    """Channel gate x * sigmoid(fc2(relu(fc1(mean_hw x)))) (helpers.py:57-74)."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, 1, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, 1, bias=False)

    def forward(self, x):
        g = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(F.relu(self.fc1(g))))


class ResidualUnit(nn.Module):
    """BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN [-> SE]  +  shortcut (helpers.py:77-121).
    The shortcut is a strided subsample when in == depth (the reference's MaxPool2d(1, stride)), else 1x1 conv + BN."""

    def __init__(self, in_channel, depth, stride, se):
        super().__init__()
        self.stride = stride
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, 1, stride, bias=False),
                                                nn.BatchNorm2d(depth))
        layers = [nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, 3, 1, 1, bias=False), nn.PReLU(depth),
                  nn.Conv2d(depth, depth, 3, stride, 1, bias=False), nn.BatchNorm2d(depth)]
        if se:
            layers.append(SEModule(depth, 16))
        self.res_layer = nn.Sequential(*layers)

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)


class GradualStyleBlock(nn.Module):
    """log2(spatial) stride-2 conv3x3 + LeakyReLU(0.01) down to 1x1, then EqualLinear (psp_encoders.py:33-53)."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        mods, c = [], in_c
        for _ in range(int(math.log2(spatial))):
            mods += [nn.Conv2d(c, out_c, 3, 2, 1), nn.LeakyReLU()]
            c = out_c
        self.convs = nn.Sequential(*mods)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        return self.linear(self.convs(x).view(-1, self.out_c))


def _fold(conv, bn):
    """conv (no bias) followed by eval-mode BatchNorm == conv with scaled filters and a bias."""
    g = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return (conv.weight * g.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last), bn.bias - bn.running_mean * g


class Encoder4Editing(nn.Module):
    def __init__(self, num_layers, mode='ir', image_resolution=256):
        super().__init__()
        if num_layers not in _TRUNK:
            raise ValueError('num_layers should be 50, 100 or 152, got %r' % (num_layers,))
        if mode not in ('ir', 'ir_se'):
            raise ValueError("mode should be 'ir' or 'ir_se', got %r" % (mode,))
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        units, c = [], 64
        for depth, n in _TRUNK[num_layers]:
            for u in range(n):
                units.append(ResidualUnit(c, depth, 2 if u == 0 else 1, mode == 'ir_se'))
                c = depth
        self.body = nn.Sequential(*units)
        # feature taps after these trunk units (psp_encoders.py:171-177): 128ch @ /4, 256ch @ /8, 512ch @ /16
        self._taps = (6, 20, 23)
        self.style_count = 2 * int(math.log2(image_resolution)) - 2
        self.coarse_ind, self.middle_ind = 3, 7
        self.styles = nn.ModuleList(
            GradualStyleBlock(512, 512, 16 if i < self.coarse_ind else 32 if i < self.middle_ind else 64)
            for i in range(self.style_count))
        self.latlayer1 = nn.Conv2d(256, 512, 1)
        self.latlayer2 = nn.Conv2d(128, 512, 1)
        self._plan_key, self._plan = None, None

    # ------------------------------------------------------------------ reference-shaped (autograd-capable) forward
    def _features(self, x):
        x = self.input_layer(x)
        taps = []
        for i, unit in enumerate(self.body):
            x = unit(x)
            if i in self._taps:
                taps.append(x)
        c1, c2, c3 = taps
        p2 = F.interpolate(c3, size=c2.shape[2:], mode='bilinear', align_corners=True) + self.latlayer1(c2)
        p1 = F.interpolate(p2, size=c1.shape[2:], mode='bilinear', align_corners=True) + self.latlayer2(c1)
        return c3, p2, p1

    def _groups(self):
        n = self.style_count
        return ((0, min(self.coarse_ind, n)), (min(self.coarse_ind, n), min(self.middle_ind, n)),
                (min(self.middle_ind, n), n))

    def forward(self, x):
        """[B,3,R,R] -> W+ [B,style_count,512]: w0 from the coarse map for every row, plus one delta per row > 0
        (the reference's `Inference` progressive stage, psp_encoders.py:185-199)."""
        if not (self.training or torch.is_grad_enabled()):
            return self._forward_planned(x)
        feats = self._features(x)
        rows = []
        for (lo, hi), f in zip(self._groups(), feats):
            rows += [self.styles[j](f) for j in range(lo, hi)]
        w0 = rows[0]
        return torch.stack([w0] + [w0 + d for d in rows[1:]], dim=1)

    # ------------------------------------------------------------------ inference plan
    def _state_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def invalidate_packs(self):
        """Drop the folded inference plan (needed only after in-place `.data` edits, which the version counters behind
        `_state_key` do not see)."""
        self._plan_key, self._plan = None, None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packs()
        return out

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_packs()

    def _build_plan(self):
        cl = torch.channels_last
        plan = {'stem': _fold(self.input_layer[0], self.input_layer[1]), 'units': [], 'heads': []}
        for unit in self.body:
            r = unit.res_layer
            bn1 = r[0]
            g1 = bn1.weight * torch.rsqrt(bn1.running_var + bn1.eps)
            entry = {'bn1': (g1.view(1, -1, 1, 1), (bn1.bias - bn1.running_mean * g1).view(1, -1, 1, 1)),
                     'w1': r[1].weight.contiguous(memory_format=cl), 'prelu': r[2].weight,
                     'c2': _fold(r[3], r[4]), 'stride': unit.stride,
                     'se': (r[5].fc1.weight, r[5].fc2.weight) if len(r) > 5 else None,
                     'sc': _fold(unit.shortcut_layer[0], unit.shortcut_layer[1])
                     if isinstance(unit.shortcut_layer, nn.Sequential) else None}
            plan['units'].append(entry)
        for lo, hi in self._groups():
            heads = [self.styles[j] for j in range(lo, hi)]
            if not heads:
                plan['heads'].append(None)
                continue
            depth = len(heads[0].convs) // 2
            # first conv of the group: one conv with the heads' filters concatenated (they share the input map)
            w0 = torch.cat([h.convs[0].weight for h in heads], 0).contiguous(memory_format=cl)
            b0 = torch.cat([h.convs[0].bias for h in heads], 0)
            # deeper stride-2 convs: per-head filters on tiny maps (8x8 ... 1x1) -> unfold + ONE batched GEMM per depth
            # (MIOpen only has its naive kernel for grouped fp32 NHWC convs: measured 100x slower than this)
            deeper = []
            for k in range(1, depth):
                wk = torch.stack([h.convs[2 * k].weight.reshape(h.out_c, -1).t() for h in heads], 0).contiguous()   # [G, Cin*9, Cout]
                bk = torch.stack([h.convs[2 * k].bias for h in heads], 0).unsqueeze(1)                              # [G, 1, Cout]
                deeper.append((wk, bk))
            plan['heads'].append(((w0, b0), deeper, heads))
        return plan

    @torch.no_grad()
    def _forward_planned(self, x):
        key = self._state_key()
        if key != self._plan_key:
            self._plan, self._plan_key = self._build_plan(), key
        plan = self._plan
        x = x.contiguous(memory_format=torch.channels_last)
        x = F.prelu(F.conv2d(x, plan['stem'][0], plan['stem'][1], 1, 1), self.input_layer[2].weight)
        taps = []
        for i, u in enumerate(plan['units']):
            s = u['stride']
            short = x[:, :, ::s, ::s] if u['sc'] is None else F.conv2d(x, u['sc'][0], u['sc'][1], s)
            y = F.conv2d(torch.addcmul(u['bn1'][1], x, u['bn1'][0]), u['w1'], None, 1, 1)
            y = F.conv2d(F.prelu(y, u['prelu']), u['c2'][0], u['c2'][1], s, 1)
            if u['se'] is not None:
                g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(y.mean((2, 3), keepdim=True), u['se'][0])), u['se'][1]))
                x = torch.addcmul(short, y, g)
            else:
                x = y + short
            if i in self._taps:
                taps.append(x)
        c1, c2, c3 = taps
        p2 = F.interpolate(c3, size=c2.shape[2:], mode='bilinear', align_corners=True) + self.latlayer1(c2)
        p1 = F.interpolate(p2, size=c1.shape[2:], mode='bilinear', align_corners=True) + self.latlayer2(c1)
        rows = []
        for f, grp in zip((c3, p2, p1), plan['heads']):
            if grp is None:
                continue
            (w0, b0), deeper, heads = grp
            G, C, B = len(heads), heads[0].out_c, f.shape[0]
            h = F.leaky_relu(F.conv2d(f, w0, b0, 2, 1), 0.01)                                  # [B, G*C, H, W]
            for wk, bk in deeper:
                Hc, Wc = h.shape[2], h.shape[3]
                Ho, Wo = (Hc + 1) // 2, (Wc + 1) // 2
                cols = F.unfold(h.reshape(B * G, C, Hc, Wc), 3, padding=1, stride=2)           # [B*G, C*9, Ho*Wo]
                cols = cols.view(B, G, C * 9, Ho * Wo).permute(1, 0, 3, 2).reshape(G, B * Ho * Wo, C * 9)
                h = F.leaky_relu(torch.baddbmm(bk, cols, wk), 0.01)                            # [G, B*Ho*Wo, C]
                h = h.view(G, B, Ho, Wo, C).permute(1, 0, 4, 2, 3).reshape(B, G * C, Ho, Wo)
            if h.shape[2] != 1 or h.shape[3] != 1:
                raise RuntimeError('style heads expect a %dx%d feature map' % (heads[0].spatial, heads[0].spatial))
            h = h.reshape(B, G, C)                                                             # spatial is 1x1 here
            rows += [head.linear(h[:, j].contiguous()) for j, head in enumerate(heads)]
        w0 = rows[0]
        return torch.stack([w0] + [w0 + d for d in rows[1:]], dim=1)
