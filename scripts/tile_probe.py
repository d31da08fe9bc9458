"""Where does a tile's time go?  Times the hot pre-split-input layers in the generator's own launch configuration (B=64)
under the kernel's probe switch SGDFR_SPLIT_DBG (1: one channel block only, 2: no epilogue, 4: no K loop)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(os.environ.get('B', 64))
out = []
# (cin, cout, h, up, rgb fused, emits xs, stores y)
for cin, cout, h, up, rgb, exs, wy in [(512, 512, 32, 0, 1, 1, 0), (256, 256, 64, 0, 1, 1, 0), (128, 128, 128, 0, 1, 1, 0),
                                         (64, 64, 256, 0, 1, 0, 0), (512, 256, 32, 1, 0, 0, 1), (256, 128, 64, 1, 0, 0, 1), (128, 64, 128, 1, 0, 0, 1)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda')
    s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
    nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
    wsp = F_.prepack_split(w, 'fp16x3')
    xs = F_.to_split(x, s, 'fp16x3')
    del x
    if up:
        ps = ((h + 1) * (h + 1) + 31) // 32 * 32 if os.environ.get('DENSE_PLANES', '0') == '0' else 0      # the chain's padded planes
        buf = torch.empty((B, cout, 4, ps) if ps else (B, cout, 4, h + 1, h + 1), device='cuda')
        fn = lambda: F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=N.MODE_UP3, x_split=(B, cin, h, h), batch=B, out=buf,
                                      plane_stride=ps)
    else:
        rw = torch.randn(3, cout, device='cuda'); rs = torch.randn(B, cout, device='cuda'); sn = torch.randn(B, cout, device='cuda')
        fn = lambda: F_.modconv_split(xs, wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3', x_split=(B, cin, h, h), batch=B,
                                      rgb=(rw, rs) if rgb else None, s_next=sn if exs else None, want_y=bool(wy))
    out.append('%s%d->%d@%d %.0f' % ('up' if up else 'pl', cin, cout, h, bench(fn)))
print('dbg=%s: ' % os.environ.get('SGDFR_SPLIT_DBG', '0') + ' | '.join(out), flush=True)
