import sys, torch
sys.path.insert(0, '/root/repo')
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
for cin, cout, h in [(64, 64, 256), (128, 128, 128), (256, 256, 64)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda')
    s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
    nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
    wr = torch.randn(3, cout, device='cuda'); sr = torch.randn(B, cout, device='cuda')
    wsp = F_.prepack_split(w, 'fp16x3')
    t0 = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3'))
    t1 = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3', rgb=(wr, sr)))
    t2 = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3', rgb=(wr, sr), want_y=False))
    print('%d->%d @%d: plain %.0f us | +rgb %.0f us | rgb only (no y store) %.0f us' % (cin, cout, h, t0, t1, t2))
