"""Same-process A/B of the split blur (parity planes -> next conv's split input) on the big levels at B=64:
    python scripts/blur_ab.py VAR=a,b [VAR2=a,b]      (environment switches read per call by upfirdn2d.hip)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_
variants = [{}]
for v in sys.argv[1:]:
    name, vals = v.split('=')
    variants = [dict(b, **{name: x}) for b in variants for x in vals.split(',')]
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fir = torch.tensor([[1., 3, 3, 1]], device='cuda'); fir = (fir.t() @ fir); fir = fir / fir.sum() * 4
for C, h in [(512, 16), (256, 32), (128, 64), (64, 128)]:
    B = 64
    ps = ((h + 1) * (h + 1) + 31) // 32 * 32
    planes = torch.randn(B, C, 4, ps, device='cuda')
    nz = torch.randn(1, 1, 2 * h, 2 * h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(C, device='cuda')
    sn = torch.randn(B, C, device='cuda')
    fn = lambda: F_.blur_bias_act_split(planes, fir, h, h, sn, nz, nw, bias, True, arith='fp16x3', plane_stride=ps)
    gb = (planes.numel() * 4 + B * C * 4 * h * h * 4) / 1e9
    best, same, ref = [1e30] * len(variants), [], None
    for var in variants:
        os.environ.update(var)
        out = fn().clone()
        same.append(True if ref is None else bool(torch.equal(out, ref)))
        ref = out if ref is None else ref
        for _ in range(3): fn()
    for _ in range(3):
        for i, var in enumerate(variants):
            os.environ.update(var)
            fn()
            best[i] = min(best[i], timed(fn))
    print('C%d %d->%d | ' % (C, h, 2 * h) + ' | '.join('%s %.0f us %.2f TB/s%s' % (','.join('%s=%s' % kv for kv in var.items()), t, gb / t * 1e3,
                                                                                 '' if ok else ' DIFFERENT') for var, t, ok in zip(variants, best, same)), flush=True)
