"""Same-process A/B of the plain split conv against its 1-D Winograd F(2,3) form (csrc/wsplit.hip) on the generator's plain layer
shapes in their chain form (pre-split / pre-transformed input, xs_out + fused ToRGB, no y):  python scripts/wsplit_ab.py [--batch 64]"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_      # noqa: E402

LAYERS = [(512, 512, 16), (512, 512, 32), (256, 256, 64), (128, 128, 128)]


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--y', type=int, default=0, help='1: write y only (no xs_out / ToRGB)')
    args = ap.parse_args()
    B = args.batch
    for cin, cout, h in LAYERS:
        w = torch.randn(1, cout, cin, 3, 3, device='cuda')
        x = torch.randn(B, cin, h, h, device='cuda')
        s = torch.randn(B, cin, device='cuda')
        d = torch.rand(B, cout, device='cuda') + 0.5
        nz = torch.randn(1, 1, h, h, device='cuda')
        nw = torch.full((1,), 0.1, device='cuda')
        bias = torch.randn(cout, device='cuda')
        sn = torch.randn(B, cout, device='cuda')
        rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda'))
        wsp, wws, wws4 = F_.prepack_split(w, 'fp16x3'), F_.prepack_wsplit(w, 'fp16x3'), F_.prepack_wsplit(w, 'fp16x3', f=4)
        xs, vs, vs4 = F_.to_split(x, s, 'fp16x3'), F_.to_wsplit(x, s, 'fp16x3'), F_.to_wsplit(x, s, 'fp16x3', f=4)
        kw = dict(rgb=None, s_next=None, want_y=True) if args.y else dict(rgb=rgb, s_next=sn, want_y=False)
        plain = lambda: F_.modconv_split(xs, wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3', x_split=(B, cin, h, h), batch=B, **kw)
        wino = lambda: F_.modconv_wsplit(vs, (B, cin, h, h), wws, d, cout, nz, nw, bias, True, arith='fp16x3', **kw)
        wino4 = lambda: F_.modconv_wsplit(vs4, (B, cin, h, h), wws4, d, cout, nz, nw, bias, True, arith='fp16x3', f=4, **kw)
        ra, rb = plain(), wino()
        ra = ra if isinstance(ra, tuple) else (ra,)
        rb = rb if isinstance(rb, tuple) else (rb,)
        diff = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ra, rb) if a is not None and a.dtype == torch.float32)
        rc = wino4()
        rc = rc if isinstance(rc, tuple) else (rc,)
        diff4 = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ra, rc) if a is not None and a.dtype == torch.float32)
        best = [1e9, 1e9, 1e9]
        for _ in range(args.rounds):
            for k, fn in enumerate((plain, wino, wino4)):
                fn()
                best[k] = min(best[k], timed(fn, args.reps))
        fl = B * F_.conv_flops(cin, cout, h, h)
        print('plain %d->%d@%d | direct %.0f us %.0f TF | F(2,3) %.0f us %.0f TF %.3fx diff %.2e | F(4,3) %.0f us %.0f TF %.3fx diff %.2e'
              % (cin, cout, h, best[0], fl / best[0] / 1e6, best[1], fl / best[1] / 1e6, best[0] / best[1], diff,
                 best[2], fl / best[2] / 1e6, best[0] / best[2], diff4), flush=True)
        del w, x, xs, vs, vs4


if __name__ == '__main__':
    main()
