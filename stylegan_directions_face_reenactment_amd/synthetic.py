"""Deterministic synthetic generator parameters (no checkpoints exist offline).

The reference ships no weights (``pretrained_models/`` is git-ignored,
/root/reference/.gitignore:1-2), so parity tests, golden fixtures and the bench
all run on synthetic parameters.  The 99 MB state cannot be committed, so it is
regenerated bit-identically wherever it is needed from a counter-based
generator: ``value = f(seed, crc32(key), element_index)``.

Only integer hashing and exact IEEE add/mul are used (no libm transcendental),
so the same float32 values come out on any host CPU / numpy build:

  * 64-bit splitmix finaliser over ``seed ^ key_hash ^ (index+1)*golden``
  * the four 16-bit fields of the hash are summed (Irwin-Hall, n=4) and
    rescaled to zero mean / unit variance -> approximately N(0,1), |x| <= 3.47

Distributions follow the reference's default initialisation
(model.py:135,218-220,294,421: randn weights, randn/lr_mul for the mapping MLP)
except that parameters the reference initialises to *zero* (noise strengths
model.py:280, activation biases fused_act.py:77, ToRGB biases model.py:348) get
small non-zero values so every term of the forward pass is exercised.
"""
import zlib

import numpy as np
import torch

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(x):
    x = x.copy()
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def counter_normal(seed, key, n):
    """n approximately-N(0,1) float64 values for (seed, key); exact on any host."""
    kh = np.uint64(zlib.crc32(key.encode()) & 0xFFFFFFFF)
    base = (np.uint64(seed) << np.uint64(32)) ^ kh
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        h = _mix64(_mix64(idx * _GOLDEN + base) ^ (base * _M2))
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m)
         + (h >> np.uint64(48))).astype(np.float64)
    # four U{0..65535}: mean 2*65535, var 4*(65536^2-1)/12
    return (s - 2.0 * 65535.0) * (1.0 / np.sqrt((65536.0 ** 2 - 1.0) / 3.0))


def counter_tensor(seed, key, shape, mean=0.0, std=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = counter_normal(seed, key, n) * std + mean
    return torch.from_numpy(v.astype(np.float32).reshape(tuple(shape)))


def _spec(key, lr_mlp):
    """(mean, std) for a Generator state_dict key; None -> keep constructor value."""
    if key.endswith('.kernel'):                       # FIR buffers: constructor values
        return None
    if key.startswith('style.'):
        if key.endswith('.weight'):
            return 0.0, 1.0 / lr_mlp                  # randn / lr_mul (model.py:135)
        return 0.0, 0.1 / lr_mlp                      # bias (zero in the reference) -> b*lr_mul ~ 0.1
    if key.startswith('noises.'):
        return 0.0, 1.0
    if key == 'input.input':
        return 0.0, 1.0
    if key.endswith('modulation.weight'):
        return 0.0, 1.0
    if key.endswith('modulation.bias'):
        return 1.0, 0.1                               # bias_init=1 (model.py:222)
    if key.endswith('conv.weight'):
        return 0.0, 1.0
    if key.endswith('noise.weight'):
        return 0.1, 0.05                              # zero in the reference
    if key.endswith('activate.bias'):
        return 0.0, 0.1                               # zero in the reference
    if key.endswith('.bias'):                         # to_rgb*.bias
        return 0.0, 0.1
    raise KeyError(key)


def synthetic_state_dict(template, seed=0, lr_mlp=0.01):
    """Fill a Generator-shaped state_dict (key -> tensor) with synthetic values.

    `template` is any mapping key -> tensor with the reference's key set
    (SURVEY.md §8 a14); returned tensors are fresh CPU float32.
    """
    out = {}
    for key, t in template.items():
        spec = _spec(key, lr_mlp)
        if spec is None:
            out[key] = t.detach().clone().float().cpu()
        else:
            out[key] = counter_tensor(seed, key, tuple(t.shape), *spec)
    return out


def trained_like_state_dict(template, seed=0, outlier_fraction=0.01, outlier_gain=30.0, bias_sigma=0.5):
    """synthetic_state_dict reshaped towards what a TRAINED g_ema looks like: heavy-tailed weights instead of
    N(0,1) everywhere --
      * every 3x3 conv weight: `outlier_fraction` of its INPUT channels (at least one) multiplied by `outlier_gain` (outlier
        channels that dominate the contraction; a gain on output channels would be undone by the demodulation);
      * every modulation bias of a 3x3 conv (constructor value 1, model.py:216): log-normal, exp(N(0, bias_sigma^2)) -- per-channel
        style magnitudes spread over several binades;
      * ToRGB weights (no demodulation) keep their scale.
    The fp16x3 range plan is calibrated on the first batch; this is the weight set its re-render rate is quoted on
    (tests/test_gpu_generator.py::test_range_plan_on_trained_like_weights, bench.py `range_plan_stress`)."""
    import torch
    out = synthetic_state_dict(template, seed)
    for key, t in out.items():
        if key.endswith('conv.weight') and t.dim() == 5 and t.shape[-1] == 3:
            cin = t.shape[2]
            n = max(1, int(round(outlier_fraction * cin)))
            pick = torch.from_numpy(counter_normal(seed, key + '.outliers', cin)).argsort()[:n]
            t[:, :, pick] *= outlier_gain
        elif key.endswith('conv.modulation.bias') and ('to_rgb' not in key):
            t.copy_(torch.exp(counter_tensor(seed, key + '.lognormal', tuple(t.shape), 0.0, bias_sigma)))
    return out


def synthetic_latents(seed, batch, n_latent=14, style_dim=512, key='wplus', std=1.0):
    """Random W+ codes [B, n_latent, 512] (bench config 2: 'random w+')."""
    return counter_tensor(seed, key, (batch, n_latent, style_dim), 0.0, std)


def synthetic_z(seed, batch, style_dim=512, key='z'):
    return counter_tensor(seed, key, (batch, style_dim))


def synthetic_direction_state(seed, input_dim=15, out_dim=512, num_layers=8, w_plus=True):
    """DirectionMatrix.linear parameters: N(0,0.03) weights (direction_matrix.py:31-32)."""
    rows = out_dim * num_layers if w_plus else out_dim
    return {
        'linear.weight': counter_tensor(seed, 'A.linear.weight', (rows, input_dim), 0.0, 0.03),
        'linear.bias': counter_tensor(seed, 'A.linear.bias', (rows,), 0.0, 0.02),
    }


def synthetic_shape_params(seed, key, n):
    """3DMM parameters shaped like the reference's `calculate_shapemodel` output (libs/utilities/generic.py:22-34):
    (angles [n,3] yaw/pitch/roll in degrees, {'pose': [n,6] (jaw = column 3), 'alpha_exp': [n,50]}) -- the inputs of the
    shift-vector construction; DECA itself is not available offline."""
    ang = counter_tensor(seed, key + '.ang', (n, 3), 0.0, 12.0)
    pose = counter_tensor(seed, key + '.pose', (n, 6), 0.03, 0.08)
    exp = counter_tensor(seed, key + '.exp', (n, 50), 0.1, 0.7)
    return ang, {'pose': pose, 'alpha_exp': exp}


def synthetic_encoder_state(template, seed=0):
    """Fill an Encoder4Editing-shaped state_dict (psp_encoders.py:122-160 key set) with synthetic values:
    conv filters ~ N(0, gain/fan_in) so activations stay O(1) through the 24 residual units and the style heads, BatchNorm statistics near
    (0, 1) with positive variances, PReLU slopes near 0.25, randn EqualLinear weights (lr_mul=1)."""
    out = {}
    for key, t in template.items():
        shape = tuple(t.shape)
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith('running_var'):
            v = counter_normal(seed, key, int(np.prod(shape))) * 0.2 + 1.0
            out[key] = torch.from_numpy(np.maximum(v, 0.3).astype(np.float32).reshape(shape))
        elif key.endswith('running_mean'):
            out[key] = counter_tensor(seed, key, shape, 0.0, 0.1)
        elif len(shape) == 4:                                     # conv filters [Cout,Cin/groups,kh,kw]
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 2.0 if key.startswith('styles.') else 0.7     # heads: plain LeakyReLU chains; trunk: residual sums
            out[key] = counter_tensor(seed, key, shape, 0.0, float(np.sqrt(gain / fan_in)))
        elif len(shape) == 2:                                     # EqualLinear weight
            out[key] = counter_tensor(seed, key, shape, 0.0, 1.0)
        elif '.linear.' in key:                                   # EqualLinear bias
            out[key] = counter_tensor(seed, key, shape, 0.0, 0.1)
        elif key.endswith('.bias'):                               # BatchNorm shift / conv bias
            out[key] = counter_tensor(seed, key, shape, 0.0, 0.05)
        elif key.endswith('.weight'):                             # 1-D: BatchNorm scale or PReLU slope
            is_prelu = key.endswith('input_layer.2.weight') or key.endswith('res_layer.2.weight')
            out[key] = counter_tensor(seed, key, shape, 0.25 if is_prelu else 1.0, 0.05 if is_prelu else 0.1)
        else:
            raise KeyError(key)
    return out
