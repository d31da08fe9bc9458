"""Per-layer timing of the conv kernels (direct MFMA vs Winograd) outside the generator: python scripts/conv_microbench.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N

def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def main():
    B = 64
    shapes = [(512, 512, 16), (512, 512, 32), (256, 256, 64), (128, 128, 128), (64, 64, 256)]
    which = sys.argv[1:] or ['direct', 'wino']
    if 'up' in which:
        for cin, cout, h in [(512, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
            w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda')
            s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
            wp, q, qt = F_.prepack(w); fl = B * F_.conv_flops(cin, cout, h, h)
            t = bench(lambda: F_.modconv_raw(x, wp, s, d, cout, N.MODE_UP3, h, h))
            wsp = F_.prepack_split(w, 'fp16x3')
            t2 = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, mode=N.MODE_UP3, arith='fp16x3'))
            print('up %4d->%4d @%3d: fp32 %7.1f us %6.1f TF | fp16x3 %7.1f us %6.1f TF-eq' % (cin, cout, h, t * 1e6, fl / t / 1e12, t2 * 1e6, fl / t2 / 1e12), flush=True)
        return
    for cin, cout, h in shapes:
        w = torch.randn(1, cout, cin, 3, 3, device='cuda')
        x = torch.randn(B, cin, h, h, device='cuda')
        s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
        nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
        wp, q, qt = F_.prepack(w); u = F_.prepack_wino(w)
        fl = B * F_.conv_flops(cin, cout, h, h)
        out = []
        if 'direct' in which:
            t = bench(lambda: F_.modconv_raw(x, wp, s, d, cout, N.MODE_PLAIN3, h, h, nz, nw, bias, True))
            out.append('direct %7.1f us %6.1f TF' % (t * 1e6, fl / t / 1e12))
        if 'wino' in which:
            t = bench(lambda: F_.modconv_wino(x, u, s, d, cout, nz, nw, bias, True))
            out.append('wino %7.1f us %6.1f TF' % (t * 1e6, fl / t / 1e12))
        if 'split' in which:
            for ar in ('bf16x3', 'fp16x3'):
                wsp = F_.prepack_split(w, ar)
                t = bench(lambda: F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith=ar))
                out.append('%s %7.1f us %6.1f TF-eq' % (ar, t * 1e6, fl / t / 1e12))
        print('%4d->%4d @%3d: %s' % (cin, cout, h, ' | '.join(out)), flush=True)

if __name__ == '__main__':
    main()
