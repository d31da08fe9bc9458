"""Race screen of csrc/wswide.hip (counted vmcnt waits, rings that run across tiles): every generator shape of the F(4,3) layers at
B = 64 and B = 32, chain form, launched REPS times alone and beside an HBM-heavy neighbour on a second stream; every launch's
hand-over and ToRGB partial sums must equal the first launch's bits AND the 64-tile kernel's.
    python scripts/wswide_soak.py [reps]"""
import os
import sys
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
side = torch.cuda.Stream()
junk_a = torch.randn(64 * 1024 * 1024, device='cuda')
junk_b = torch.empty_like(junk_a)
bad = 0
for B in (64, 32):
    for cin, cout, h in ((512, 512, 32), (256, 256, 64), (128, 128, 128)):
        torch.manual_seed(cin + B)
        w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda'); s = torch.randn(B, cin, device='cuda')
        d = torch.rand(B, cout, device='cuda') + 0.5; nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda')
        bias = torch.randn(cout, device='cuda'); sn = torch.randn(B, cout, device='cuda')
        rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda'))
        wws = F_.prepack_wsplit(w, 'fp16x3', f=4); vs = F_.to_wsplit(x, s, 'fp16x3', f=4)
        del x, w
        fn = lambda: F_.modconv_wsplit(vs, (B, cin, h, h), wws, d, cout, nz, nw, bias, True, arith='fp16x3', f=4, rgb=rgb, s_next=sn, want_y=False)
        os.environ['SGDFR_WSPLIT_WIDE_NOW'] = '0'
        _, part0, xs0 = fn()
        torch.cuda.synchronize()
        os.environ['SGDFR_WSPLIT_WIDE_NOW'] = '2'
        n_bad = 0
        for it in range(reps):
            if it % 2:
                with torch.cuda.stream(side):      # an HBM-bound neighbour: operand DMAs and stores see a loaded memory system
                    junk_b.copy_(junk_a)
            _, part, xs = fn()
            if not (torch.equal(xs, xs0) and torch.equal(part, part0)):
                n_bad += 1
        torch.cuda.synchronize()
        bad += n_bad
        print('B=%d %d->%d@%d: %d launches, %d differ from the 64-tile kernel' % (B, cin, cout, h, reps, n_bad), flush=True)
        del vs, wws
        torch.cuda.empty_cache()
print('SOAK', 'FAILED' if bad else 'OK')
sys.exit(1 if bad else 0)
