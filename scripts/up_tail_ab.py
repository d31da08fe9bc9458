"""Same-process A/B of the tail-round launch of the deep-plan transposed conv (split.hip launch_up_deep_tail, SGDFR_SPLIT_UP_TAIL
is read per call): the generator's four big transposed layers at B = 64 / 32 / 16, padded + interleaved planes (the chain's layout),
us per launch with one launch of full tiles and with the tail round on half tiles.   python scripts/up_tail_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import functional as F_, synthetic as S
N = F_.N

def bench(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for B in (64, 32, 16):
    for cin, cout, H in ((512, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)):
        if not F_._shape_query('sgdfr_modconv2d_split_f8_ok', B, cin, cout, H, H, N.MODE_UP3):
            print('B=%2d %3d->%3d @%3d^2: not on the deep plan' % (B, cin, cout, H))
            continue
        x = S.counter_tensor(3, 'x', (B, cin, H, H)).cuda()
        w = S.counter_tensor(3, 'w', (1, cout, cin, 3, 3)).cuda()
        s = S.counter_tensor(3, 's', (B, cin), 1.0, 0.3).cuda()
        d = S.counter_tensor(3, 'd', (B, cout), 1.0, 0.2).cuda()
        wsp, xs = F_.prepack_split(w, 'fp16x3'), F_.to_split(x, s, 'fp16x3')
        ps = ((H + 1) * (H + 1) + 31) // 32 * 32
        out = torch.empty(B, cout, 4, ps, device='cuda')
        tiles = -(-B * ps // 256) * (cout // 64)
        run = lambda: F_.modconv_split(xs, wsp, None, d, cout, arith='fp16x3', mode=N.MODE_UP3, x_split=tuple(x.shape), batch=B, plane_stride=ps, out=out)
        t = {}
        for rep in range(2):
            for flag in ('0', '100000'):
                os.environ['SGDFR_SPLIT_UP_TAIL'] = flag
                t.setdefault(flag, []).append(bench(run))
        print('B=%2d %3d->%3d @%3d^2: %5d tiles = %5.2f rounds | one launch %6.1f us | tail on half tiles %6.1f us (%+.1f %%)'
              % (B, cin, cout, H, tiles, tiles / 256, min(t['0']), min(t['100000']), 100 * (min(t['100000']) / min(t['0']) - 1)))


# the adjoint (dL/d(x*s) of the transposed conv, mode DOWN3: 128 x 256 tiles, tail on 128 x 128) at the trainer's per-rank batch
print('--- DOWN3 (backward of the transposed conv)')
for B in (16, 32, 64):
    for cin, cout, H in ((512, 256, 32), (256, 128, 64), (128, 64, 128)):       # forward shapes: the adjoint maps cout planes -> cin
        if not F_.split_ok(B, cout, cin, H, H, N.MODE_DOWN3):
            continue
        w = S.counter_tensor(3, 'wd', (1, cout, cin, 3, 3)).cuda()
        gT = S.counter_tensor(3, 'gT', (B, cout, 4, H + 1, H + 1)).cuda()
        d = S.counter_tensor(3, 'dd', (B, cout), 1.0, 0.2).cuda()
        gxs = F_.planes_to_split(gT, d, 'fp16x3')
        wsp = F_.prepack_split(w, 'fp16x3', adjoint='down')
        tiles = -(-B * (H + 1) * (H + 1) // 256) * (cin // 128)
        run = lambda: F_.modconv_split(gxs, wsp, None, None, cin, mode=N.MODE_DOWN3, arith='fp16x3', x_split=(B, cout, H, H), batch=B)
        t, outs = {}, {}
        for rep in range(2):
            for flag in ('0', '100000'):
                os.environ['SGDFR_SPLIT_UP_TAIL'] = flag
                t.setdefault(flag, []).append(bench(run))
                outs[flag] = run()
        same = torch.equal(outs['0'], outs['100000'])
        print('B=%2d %3d->%3d @%3d^2: %5d tiles = %5.2f rounds | one launch %6.1f us | tail on half tiles %6.1f us (%+.1f %%) %s'
              % (B, cout, cin, H, tiles, tiles / 256, min(t['0']), min(t['100000']), 100 * (min(t['100000']) / min(t['0']) - 1),
                 'same bits' if same else 'DIFFERENT BITS'))
