"""Same-process A/B of tiling plans on the generator's layer shapes in their CHAIN form (pre-split input; plain layers emit the
next layer's split input and / or the fused ToRGB partial sums like Generator.forward does; transposed convs write padded parity
planes).  Environment switches read per call by split.hip (SGDFR_SPLIT_UP4, SGDFR_SPLIT_P4) are flipped between the timing
loops, so both variants run on the same box in the same thermal state, interleaved.

    python scripts/layer_ab.py [--batch 64] [--reps 20] [--rounds 3] VAR=a,b [VAR2=a,b ...]
"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N      # noqa: E402

LAYERS = [  # (cin, cout, h, up, emits_xs, fuses_rgb, wants_y)  -- Generator(256, cm=1) at B=64, as _synthesis plans them
    (512, 512, 16, 1, 0, 0, 1), (512, 512, 16, 0, 1, 1, 0),
    (512, 256, 32, 1, 0, 0, 1), (512, 512, 32, 0, 1, 1, 0),
    (256, 128, 64, 1, 0, 0, 1), (256, 256, 64, 0, 1, 1, 0),
    (128, 64, 128, 1, 0, 0, 1), (128, 128, 128, 0, 1, 1, 0),
    (64, 64, 256, 0, 0, 1, 0),
]


def make(B, cin, cout, h, up, emit, fuse, want_y):
    w = torch.randn(1, cout, cin, 3, 3, device='cuda')
    x = torch.randn(B, cin, h, h, device='cuda')
    s = torch.randn(B, cin, device='cuda')
    d = torch.rand(B, cout, device='cuda') + 0.5
    wsp = F_.prepack_split(w, 'fp16x3')
    xs = F_.to_split(x, s, 'fp16x3')
    del x, w
    nz = torch.randn(1, 1, h, h, device='cuda')
    nw = torch.full((1,), 0.1, device='cuda')
    bias = torch.randn(cout, device='cuda')
    sn = torch.randn(B, cout, device='cuda') if emit else None
    rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda')) if fuse else None
    if up:
        ps = ((h + 1) * (h + 1) + 31) // 32 * 32
        out = torch.empty(B, cout, 4, ps, device='cuda')
        return lambda: F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=N.MODE_UP3, x_split=(B, cin, h, h), batch=B,
                                        plane_stride=ps, out=out)
    return lambda: F_.modconv_split(xs, wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3', x_split=(B, cin, h, h), batch=B,
                                    rgb=rgb, s_next=sn, want_y=bool(want_y) and sn is None)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def results(fn):
    r = fn()
    r = r if isinstance(r, tuple) else (r,)
    return [t.clone() for t in r if t is not None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('vars', nargs='+')
    args = ap.parse_args()
    variants = [{}]
    for v in args.vars:
        name, vals = v.split('=')
        variants = [dict(base, **{name: val}) for base in variants for val in vals.split(',')]

    def select(var):
        for k, v in var.items():
            os.environ[k] = v
        F_._shape_query.cache_clear()

    for L in LAYERS:
        cin, cout, h, up = L[:4]
        tag = '%s %d->%d@%d' % ('up   ' if up else 'plain', cin, cout, h)
        if args.only and args.only not in tag.replace(' ', ''):
            continue
        fn = make(args.batch, *L)
        best = [1e30] * len(variants)
        ref = None
        same = []
        for i, var in enumerate(variants):
            select(var)
            out = results(fn)
            if ref is None:
                ref = out
                same.append(True)
            else:
                same.append(all(torch.equal(a, b) for a, b in zip(ref, out)))
            for _ in range(3):
                fn()
        for _ in range(args.rounds):
            for i, var in enumerate(variants):
                select(var)
                fn()
                best[i] = min(best[i], timed(fn, args.reps))
        flops = 2.0 * 9 * cin * cout * h * h * args.batch
        print(tag + ' | ' + ' | '.join('%s %.0f us %.0f TF%s' % (','.join('%s=%s' % kv for kv in var.items()), t, flops / t / 1e6,
                                                                  '' if ok else ' DIFFERENT') for var, t, ok in zip(variants, best, same)),
              flush=True)
        del fn, ref
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
