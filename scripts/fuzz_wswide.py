"""Randomised bit-identity fuzz of csrc/wswide.hip against wsplit_kernel<., 6> (GPU box): random shapes inside the wide launcher's
constraints (Cin % 16, Cin >= 64, Cout % 128, H % 16, W % 32), random batch, random persistent grid, every epilogue combination the
host wrapper can ask for (y / split hand-over / fused ToRGB partial sums, shared / per-sample / no noise, activation on and off), both
arithmetics, inputs loud enough in some cases to clamp (the saturation count must agree too).  tests/test_gpu_wsplit.py holds the fixed
cases; this is the wider net.    python scripts/fuzz_wswide.py [cases] [seed]"""
import os, sys, random, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
torch.manual_seed(seed)
bad = 0
for case in range(n_cases):
    cin = rng.choice([64, 80, 96, 128, 160, 256, 512])
    cout = rng.choice([128, 128, 256, 384, 512])
    H = 16 * rng.randint(1, 8)
    W = 32 * rng.choice([1, 1, 2, 2, 3, 4, 8])
    B = rng.randint(1, 6)
    while B * cin * H * W > 48 * 1024 * 1024 or B * cout * H * W > 48 * 1024 * 1024:
        B = max(1, B - 1)
        if B == 1:
            H = max(16, H // 2)
    tiles = B * (H // 16) * (W // 32) * (cout // 128)
    persist = rng.choice([0, 0, 8, 16, 24, 256])
    arith = rng.choice(['fp16x3', 'fp16x3', 'bf16x3'])
    loud = rng.random() < 0.25
    act = rng.random() < 0.8
    noise_kind = rng.choice(['shared', 'sample', 'none'])
    x = torch.randn(B, cin, H, W, device='cuda') * (rng.choice([3e4, 1e6, 1e8]) if loud else 1.0)
    w = torch.randn(1, cout, cin, 3, 3, device='cuda')
    s = torch.randn(B, cin, device='cuda') * 0.3 + 1.0
    d = torch.rand(B, cout, device='cuda') + 0.5
    bias = torch.randn(cout, device='cuda') * 0.1
    nw = torch.full((1,), 0.1, device='cuda')
    sn = torch.randn(B, cout, device='cuda') * 0.3 + 1.0
    rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda') * 0.3 + 1.0)
    noise = None if noise_kind == 'none' else torch.randn(B if noise_kind == 'sample' else 1, 1, H, W, device='cuda')
    vs = F_.to_wsplit(x, s, arith, f=4)
    wsp = F_.prepack_wsplit(w, arith, f=4)
    if persist:
        os.environ['SGDFR_WSPLIT_PERSIST'] = str(persist)
    else:
        os.environ.pop('SGDFR_WSPLIT_PERSIST', None)
    variants = [{}, {'s_next': sn}, {'s_next': sn, 'want_y': False}, {'rgb': rgb}, {'rgb': rgb, 'want_y': False},
                {'s_next': sn, 'rgb': rgb}, {'s_next': sn, 'rgb': rgb, 'want_y': False}]
    for kw in rng.sample(variants, 3):
        out = {}
        for wide in ('0', '2'):
            os.environ['SGDFR_WSPLIT_WIDE_NOW'] = wide
            word = F_.new_saturation_word(x.device)
            with F_.saturation_sink(word):
                r = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw if noise is not None else None, bias, act, arith=arith, f=4, **kw)
            torch.cuda.synchronize()
            r = r if isinstance(r, tuple) else (r,)
            out[wide] = [t.clone() if t is not None else None for t in r] + [int(word.item())]
        ok = True
        for a, b in zip(out['0'], out['2']):
            if isinstance(a, torch.Tensor):
                ok = ok and torch.equal(a, b)
            else:
                ok = ok and a == b
        tag = 'ok ' if ok else 'BAD'
        bad += not ok
        print('%s case %3d: %3d->%3d @%3dx%3d B=%d tiles=%4d persist=%3d %s act=%d noise=%-6s loud=%d sat=%d outputs=%s' % (
            tag, case, cin, cout, H, W, B, tiles, persist, arith, act, noise_kind, loud, out['2'][-1], '+'.join(sorted(kw)) or 'y'), flush=True)
    del x, w, vs, wsp
print('fuzz_wswide: %d cases x 3 variants, %d mismatches' % (n_cases, bad))
sys.exit(1 if bad else 0)
