"""ORACLE (test infrastructure, never shipped): CPU restatement of the e4e W+ producer.

Functional, state_dict-driven restatement of `Encoder4Editing(50, 'ir_se', R).forward` in its `Inference` stage
(/root/reference/libs/gan/encoder4editing/psp_encoders.py:122-199; trunk units helpers.py:57-121; style heads
psp_encoders.py:33-53; FPN merge helpers.py:124-140).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Pinned by oracle/make_golden_e4e.py, which runs the real reference on CPU and
asserts this file agrees before writing tests/golden/kat6_e4e.npz.
"""
import math

import torch
import torch.nn.functional as F

TAPS = (6, 20, 23)          # psp_encoders.py:171-177
COARSE, MIDDLE = 3, 7       # psp_encoders.py:144-145


def _bn(P, k, x, eps=1e-5):
    return F.batch_norm(x, P[k + '.running_mean'], P[k + '.running_var'], P[k + '.weight'], P[k + '.bias'], False, 0.0, eps)


def residual_unit(P, k, x):
    """helpers.py:96-121 (bottleneck_IR_SE); without `res_layer.5.*` keys it is bottleneck_IR (helpers.py:77-94)."""
    w2 = P[k + '.res_layer.3.weight']
    stride = 1
    if (k + '.shortcut_layer.0.weight') in P:
        sw = P[k + '.shortcut_layer.0.weight']
        # stride is not in the state_dict: the projection exists exactly for the first unit of a stage (stride 2) and
        # for the stem-width first stage; both have stride 2 in get_blocks (helpers.py:25-27)
        stride = 2
        shortcut = _bn(P, k + '.shortcut_layer.1', F.conv2d(x, sw, None, stride))
    else:
        stride = P['__stride__'][k]
        shortcut = x[:, :, ::stride, ::stride]                           # MaxPool2d(1, stride)
    y = _bn(P, k + '.res_layer.0', x)
    y = F.prelu(F.conv2d(y, P[k + '.res_layer.1.weight'], None, 1, 1), P[k + '.res_layer.2.weight'])
    y = _bn(P, k + '.res_layer.4', F.conv2d(y, w2, None, stride, 1))
    if (k + '.res_layer.5.fc1.weight') in P:
        g = F.adaptive_avg_pool2d(y, 1)
        g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, P[k + '.res_layer.5.fc1.weight'])), P[k + '.res_layer.5.fc2.weight']))
        y = y * g
    return y + shortcut


def style_head(P, k, x):
    """psp_encoders.py:33-53: stride-2 convs + LeakyReLU(0.01) to 1x1, then EqualLinear(lr_mul=1) without activation
    (StyleGAN2 model.py:148-157: F.linear(x, W/sqrt(in), bias))."""
    i = 0
    while (k + '.convs.%d.weight' % i) in P:
        x = F.leaky_relu(F.conv2d(x, P[k + '.convs.%d.weight' % i], P[k + '.convs.%d.bias' % i], 2, 1), 0.01)
        i += 2
    x = x.reshape(x.shape[0], -1)
    w = P[k + '.linear.weight']
    return F.linear(x, w * (1.0 / math.sqrt(w.shape[1])), P[k + '.linear.bias'])


def unit_strides(P):
    """Stride of every trunk unit: 2 for the first unit of each stage (helpers.py:25-27), which is the unit whose
    input width differs from its depth -- or, for the 64->64 first stage, unit 0."""
    strides, n = {}, 0
    while ('body.%d.res_layer.1.weight' % n) in P:
        w = P['body.%d.res_layer.1.weight' % n]
        strides['body.%d' % n] = 2 if (n == 0 or w.shape[0] != w.shape[1]) else 1
        n += 1
    return strides, n


def encoder_forward(P, x):
    """[B,3,R,R] -> W+ [B, n_styles, 512] (psp_encoders.py:167-199, progressive stage = Inference)."""
    P = dict(P)
    P['__stride__'], n_units = unit_strides(P)
    x = F.prelu(_bn(P, 'input_layer.1', F.conv2d(x, P['input_layer.0.weight'], None, 1, 1)), P['input_layer.2.weight'])
    taps = []
    for i in range(n_units):
        x = residual_unit(P, 'body.%d' % i, x)
        if i in TAPS:
            taps.append(x)
    c1, c2, c3 = taps
    n_styles = 0
    while ('styles.%d.linear.weight' % n_styles) in P:
        n_styles += 1
    w0 = style_head(P, 'styles.0', c3)
    rows = [w0]
    feat = c3
    for i in range(1, n_styles):
        if i == COARSE:
            feat = p2 = F.interpolate(c3, size=c2.shape[2:], mode='bilinear', align_corners=True) + \
                F.conv2d(c2, P['latlayer1.weight'], P['latlayer1.bias'])
        elif i == MIDDLE:
            feat = F.interpolate(p2, size=c1.shape[2:], mode='bilinear', align_corners=True) + \
                F.conv2d(c1, P['latlayer2.weight'], P['latlayer2.bias'])
        rows.append(w0 + style_head(P, 'styles.%d' % i, feat))
    return torch.stack(rows, dim=1)
