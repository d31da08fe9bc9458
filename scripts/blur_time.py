"""Times the split blur (parity planes -> next conv's split input) on the four big levels at B=64: python scripts/blur_time.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from stylegan_directions_face_reenactment_amd import functional as F_
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fir = torch.tensor([[1., 3, 3, 1]], device='cuda'); fir = (fir.t() @ fir); fir = fir / fir.sum() * 4
out = []
for C, h in [(512, 4), (512, 8), (512, 16), (256, 32), (128, 64), (64, 128)]:
    B = 64
    ps = ((h + 1) * (h + 1) + 31) // 32 * 32
    planes = torch.randn(B, C, 4, ps, device='cuda')
    nz = torch.randn(1, 1, 2 * h, 2 * h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(C, device='cuda')
    sn = torch.randn(B, C, device='cuda')
    t = bench(lambda: F_.blur_bias_act_split(planes, fir, h, h, sn, nz, nw, bias, True, arith='fp16x3', plane_stride=ps))
    gb = (planes.numel() * 4 + B * C * 4 * h * h * 4) / 1e9
    out.append('C%d %d->%d %.0f us (%.2f TB/s)' % (C, h, 2 * h, t, gb / t * 1e6 / 1e3))
print(' | '.join(out), flush=True)
