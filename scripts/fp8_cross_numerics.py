"""CPU simulation for DESIGN 4.10's "what comes next (0)": the split product with its two cross terms kept in fp8.
    a*b ~= a_hi*b_hi (fp16 x fp16)  +  fp8(a_lo)*fp8(b_hi) + fp8(a_hi)*fp8(b_lo)
One modulated-conv-sized contraction (K = 512 channels x 9 taps), leaky-ReLU shaped activations whose loudness varies 2^8 between
pixels (rows), against fp64; the same for today's fp16x3 / bf16x3 and for dropping cross terms.  fp8 scales: ONE power of two per
"image" (64 rows share it, as the range plan's per-image exponents would give) -- small rows then live in e4m3's lower binades.
Companion of scripts/fp8_cross_probe.hip (the speed half of the question).   python scripts/fp8_cross_numerics.py"""
import torch
torch.manual_seed(0)
K, M, N = 4608, 256, 512
loud = 2.0 ** (torch.rand(N, 1, dtype=torch.float64) * 8 - 4)
x = torch.randn(N, K, dtype=torch.float64) * loud
x = torch.where(x > 0, x, 0.2 * x)
w = torch.randn(M, K, dtype=torch.float64) / K ** 0.5
ref = x @ w.T


def r(v, dt):
    return v.to(torch.float32).to(dt).to(torch.float64)


def split(v, dt):
    h = r(v, dt)
    return h, r(v - h, dt)


def f8(v, dt, group):
    """fp8 with one power-of-two scale per `group` rows (largest element just under the format's top binade)"""
    top = 256.0 if dt == torch.float8_e4m3fn else 32768.0
    g = v.reshape(-1, group * v.shape[1]).abs().amax(dim=1, keepdim=True)
    s = (2.0 ** torch.ceil(torch.log2(g / top + 1e-300))).repeat_interleave(group, 0)
    return (v / s).to(torch.float32).to(dt).to(torch.float64) * s


res = {}
for name, dt in (('fp16x3 (default today)', torch.float16), ('bf16x3', torch.bfloat16)):
    xh, xl = split(x, dt)
    wh, wl = split(w, dt)
    res[name] = xh @ wh.T + xl @ wh.T + xh @ wl.T
xh, xl = split(x, torch.float16)
wh, wl = split(w, torch.float16)
res['fp16, main term only (1 product)'] = xh @ wh.T
res['fp16, main + activation lo (2 products)'] = xh @ wh.T + xl @ wh.T
for fmt, dt in (('e4m3', torch.float8_e4m3fn), ('e5m2', torch.float8_e5m2)):
    for group, gname in ((1, 'per-pixel scale'), (64, 'per-image scale')):
        xl8, xh8 = f8(xl, dt, group), f8(xh, dt, group)
        wl8, wh8 = f8(wl, dt, M), f8(wh, dt, M)
        res['fp16 main + %s cross terms, %s' % (fmt, gname)] = xh @ wh.T + xl8 @ wh8.T + xh8 @ wl8.T
for k, v in res.items():
    e = (v - ref).abs()
    rel_row = (e.amax(dim=1) / ref.abs().amax(dim=1)).max()
    print('%-52s max err / max|y| = %.2e   worst pixel: err / that pixel\'s max|y| = %.2e   rms = %.2e' % (
        k, e.max() / ref.abs().max(), rel_row, (e.pow(2).mean() / ref.pow(2).mean()).sqrt()))
