"""A/B of environment switches of wsplit.hip (read per launch) on the generator's F(4,3) layer shapes, chain form, same process:
    python scripts/wsplit_env_ab.py SGDFR_WSPLIT_DESYNC=0,50,100"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_      # noqa: E402


def timed(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


name, vals = sys.argv[1].split('=')
vals = vals.split(',')
B = int(os.environ.get('BATCH', '64'))
for cin, cout, h in ((512, 512, 16), (512, 512, 32), (256, 256, 64), (128, 128, 128)):
    w = torch.randn(1, cout, cin, 3, 3, device='cuda'); x = torch.randn(B, cin, h, h, device='cuda'); s = torch.randn(B, cin, device='cuda')
    d = torch.rand(B, cout, device='cuda') + 0.5; nz = torch.randn(1, 1, h, h, device='cuda'); nw = torch.full((1,), 0.1, device='cuda')
    bias = torch.randn(cout, device='cuda'); sn = torch.randn(B, cout, device='cuda')
    rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda'))
    wws = F_.prepack_wsplit(w, 'fp16x3', f=4); vs = F_.to_wsplit(x, s, 'fp16x3', f=4)
    fn = lambda: F_.modconv_wsplit(vs, (B, cin, h, h), wws, d, cout, nz, nw, bias, True, arith='fp16x3', f=4, rgb=rgb, s_next=sn, want_y=False)
    best = {v: 1e9 for v in vals}
    tot = {v: 0.0 for v in vals}
    R = int(os.environ.get('ROUNDS', '8'))
    for _ in range(R):
        for v in vals:
            os.environ[name] = v
            t = timed(fn)
            best[v] = min(best[v], t)
            tot[v] += t
    print('%d->%d@%d | ' % (cin, cout, h) + ' | '.join('%s=%s best %.0f mean %.0f us' % (name, v, best[v], tot[v] / R) for v in vals), flush=True)
