"""Pre-split ("XS") input staged by DMA vs fp32 input converted in the kernel: same bits, how much faster?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 64
for cin, cout, h, up in [(512, 512, 32, 0), (256, 256, 64, 0), (128, 128, 128, 0), (64, 64, 256, 0), (512, 256, 32, 1), (256, 128, 64, 1), (128, 64, 128, 1)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda')
    s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
    wsp = F_.prepack_split(w, 'fp16x3')
    mode = N.MODE_UP3 if up else N.MODE_PLAIN3
    xs = F_.to_split(x, s, 'fp16x3')
    a = F_.modconv_split(x, wsp, s, d, cout, arith='fp16x3', mode=mode)
    b = F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=mode, x_split=(B, cin, h, h), batch=B)
    t0 = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, arith='fp16x3', mode=mode))
    t1 = bench(lambda: F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=mode, x_split=(B, cin, h, h), batch=B))
    t2 = bench(lambda: F_.to_split(x, s, 'fp16x3'))
    print('%s %d->%d @%d: fp32 in %.0f us | pre-split in %.0f us (%.0f%%) | conversion pass alone %.0f us | identical %s' %
          ('up   ' if up else 'plain', cin, cout, h, t0, t1, 100 * t1 / t0, t2, bool(torch.equal(a, b))), flush=True)

# producer side: the plain conv's epilogue emitting the next layer's split input == to_split(y, s_next)
for cin, cout, h in [(64, 128, 64), (128, 64, 128), (32, 256, 32)]:
    Bp = 48
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(Bp, cin, h, h, device='cuda')
    s = torch.randn(Bp, cin, device='cuda'); d = torch.rand(Bp, cout, device='cuda') + 0.5
    nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
    sn = torch.randn(Bp, cout, device='cuda')
    wsp = F_.prepack_split(w, 'fp16x3')
    y, part, xs_out = F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3', s_next=sn)
    ref = F_.to_split(y, sn, 'fp16x3')
    y2, _, xs2 = F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3', s_next=sn, want_y=False)
    print('producer %d->%d @%d: xs_out == to_split(y, s_next): %s ; without y: %s (y is None: %s)' %
          (cin, cout, h, bool(torch.equal(xs_out, ref)), bool(torch.equal(xs2, ref)), y2 is None))

# producer side 2: blur writing the split form
fir = torch.tensor([[1., 3, 3, 1]], device='cuda'); fir = (fir.t() @ fir); fir = fir / fir.sum() * 4
for C, h in [(512, 8), (256, 32), (128, 64), (64, 128)]:
    Bp = 64
    planes = torch.randn(Bp, C, 4, h + 1, h + 1, device='cuda')
    nz = torch.randn(1, 1, 2 * h, 2 * h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(C, device='cuda')
    sn = torch.randn(Bp, C, device='cuda')
    y = F_.blur_bias_act(planes, fir, h, h, nz, nw, bias, True)
    ref = F_.to_split(y, sn, 'fp16x3')
    xs = F_.blur_bias_act_split(planes, fir, h, h, sn, nz, nw, bias, True, arith='fp16x3')
    t0 = bench(lambda: F_.blur_bias_act(planes, fir, h, h, nz, nw, bias, True))
    t1 = bench(lambda: F_.blur_bias_act_split(planes, fir, h, h, sn, nz, nw, bias, True, arith='fp16x3'))
    print('blur C=%d %d->%d: fp32 out %.0f us | split out %.0f us | identical %s' % (C, h, 2 * h, t0, t1, bool(torch.equal(xs, ref))))
