"""Randomised bit-identity fuzz of the tail-round launch (split.hip launch_deep_tail): random shapes of the deep-plan transposed conv
(pre-split and fp32 input, dense and padded + interleaved planes) and of its adjoint (DOWN3) whose tile count leaves a tail, one launch
of full tiles (SGDFR_SPLIT_UP_TAIL=0) against whole rounds + half tiles (=100000).   python scripts/fuzz_tail.py [n] [seed]"""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import functional as F_, synthetic as S
N = F_.N
n_cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)


def both(fn):
    out = {}
    for flag in ('0', '100000'):
        os.environ['SGDFR_SPLIT_UP_TAIL'] = flag
        out[flag] = fn()
    return out['0'], out['100000']


done = bad = skipped = 0
while done < n_cases:
    H = rng.choice([8, 16, 16, 32, 32, 64, 128])
    cin, cout = rng.choice([16, 32, 64, 128, 256]), rng.choice([64, 128, 192, 256])
    rp = (H + 1) * (H + 1)
    ps = (rp + 31) // 32 * 32
    # a batch whose tile count leaves a tail of <= 128 tiles after at least one whole round
    cands = [B for B in range(1, 400) if any(t > 256 and 0 < t % 256 <= 128 for t in (-(-B * n // 256) * (cout // 64) for n in (rp, ps)))
             and B * cout * 4 * ps * 4 < 3e9]
    if not cands:
        continue
    B = rng.choice(cands)
    if not F_._shape_query('sgdfr_modconv2d_split_f8_ok', B, cin, cout, H, H, N.MODE_UP3) or not F_.xin_ok(B, cin, cout, H, H, N.MODE_UP3):
        skipped += 1
        continue
    arith = rng.choice(['fp16x3', 'fp16x3', 'bf16x3'])
    x = S.counter_tensor(seed, 'x%d' % done, (B, cin, H, H)).cuda()
    w = S.counter_tensor(seed, 'w%d' % done, (1, cout, cin, 3, 3)).cuda()
    s = S.counter_tensor(seed, 's%d' % done, (B, cin), 1.0, 0.3).cuda()
    d = S.counter_tensor(seed, 'd%d' % done, (B, cout), 1.0, 0.2).cuda()
    wsp, xs = F_.prepack_split(w, arith), F_.to_split(x, s, arith)
    a0, a1 = both(lambda: F_.modconv_split(xs, wsp, None, d, cout, arith=arith, mode=N.MODE_UP3, x_split=tuple(x.shape), batch=B))
    b0, b1 = both(lambda: F_.modconv_split(xs, wsp, None, d, cout, arith=arith, mode=N.MODE_UP3, x_split=tuple(x.shape), batch=B, plane_stride=ps))
    c0, c1 = both(lambda: F_.modconv_split(x, wsp, s, d, cout, arith=arith, mode=N.MODE_UP3))
    valid = lambda il: il.view(B, cout, ps, 4)[:, :, :rp]
    ok = torch.equal(a0, a1) and torch.equal(valid(b0), valid(b1)) and torch.equal(c0, c1) and torch.equal(a0, c0)
    # the adjoint of the same layer (cout planes -> cin channels), when its own tile count has a tail
    td = -(-B * rp // 256) * (cin // 128) if cin % 128 == 0 else 0
    if td > 256 and 0 < td % 256 <= 128 and F_.split_ok(B, cout, cin, H, H, N.MODE_DOWN3):
        gT = S.counter_tensor(seed, 'g%d' % done, (B, cout, 4, H + 1, H + 1)).cuda()
        gxs, wd = F_.planes_to_split(gT, d, arith), F_.prepack_split(w, arith, adjoint='down')
        e0, e1 = both(lambda: F_.modconv_split(gxs, wd, None, None, cin, mode=N.MODE_DOWN3, arith=arith, x_split=(B, cout, H, H), batch=B))
        ok = ok and torch.equal(e0, e1)
    done += 1
    if not ok:
        bad += 1
        print('DIFFERENT BITS: B=%d %d->%d @%d^2 %s' % (B, cin, cout, H, arith))
    del x, w, s, d, wsp, xs, a0, a1, b0, b1, c0, c1
print('%d random shapes with a tail round (%d candidates not on the deep plan skipped): %d with different bits' % (done, skipped, bad))
sys.exit(1 if bad else 0)
