"""Random-shape cross-checks of the split conv kernels (not a test: a wider net than tests/test_gpu_split.py):
plain / transposed / strided-adjoint modes, fp32-input vs pre-split-input launches (bit-equal), padded vs dense planes
(bit-equal), all against the fp32 MFMA kernel of the same mode.  python scripts/fuzz_split.py [n] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N

def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    torch.manual_seed(0)
    done = {'plain': 0, 'up': 0, 'down': 0}
    worst = {'plain': 0.0, 'up': 0.0, 'down': 0.0}
    while sum(done.values()) < n:
        kind = rng.choice(['plain', 'up', 'down'])
        H = rng.choice([4, 8, 16, 32, 64, 128])
        cin = 16 * rng.randint(1, 16 if H >= 64 else 32)
        cout = 64 * rng.randint(1, 4 if H >= 64 else 8)
        B = rng.randint(1, max(1, min(70, (1 << 27) // (max(cin, cout) * H * H * (4 if kind != 'plain' else 1)))))
        x = torch.randn(B, cin, H, H, device='cuda'); w = torch.randn(1, cout, cin, 3, 3, device='cuda')
        s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
        if kind == 'plain':
            if not F_.split_ok(B, cin, cout, H, H):
                continue
            nz = torch.randn(1, 1, H, H, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
            wp = F_.prepack(w)[0]
            ref = F_.modconv_raw(x, wp, s, d, cout, N.MODE_PLAIN3, H, H, nz, nw, bias, True)
            wsp = F_.prepack_split(w, 'fp16x3')
            a = F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3')
            e = rel(a, ref)
            if F_.xin_ok(B, cin, cout, H, H):
                b = F_.modconv_split(F_.to_split(x, s, 'fp16x3'), wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3',
                                     x_split=tuple(x.shape), batch=B)
                assert torch.equal(a, b), ('plain xin', B, cin, cout, H)
        elif kind == 'up':
            if not F_.split_ok(B, cin, cout, H, H, N.MODE_UP3):
                continue
            wp = F_.prepack(w)[0]
            ref = F_.modconv_raw(x, wp, s, d, cout, N.MODE_UP3, H, H)
            wsp = F_.prepack_split(w, 'fp16x3')
            a = F_.modconv_split(x, wsp, s, d, cout, arith='fp16x3', mode=N.MODE_UP3)
            e = rel(a, ref)
            xin = F_.xin_ok(B, cin, cout, H, H, N.MODE_UP3)
            if xin:
                b = F_.modconv_split(F_.to_split(x, s, 'fp16x3'), wsp, None, d, cout, arith='fp16x3', mode=N.MODE_UP3,
                                     x_split=tuple(x.shape), batch=B)
                assert torch.equal(a, b), ('up xin', B, cin, cout, H)
            if F_._shape_query('sgdfr_modconv2d_split_ksplit_hint', B, cin, cout, H, H, N.MODE_UP3) == 1:
                rp = (H + 1) * (H + 1); ps = (rp + 31) // 32 * 32
                c = F_.modconv_split(x, wsp, s, d, cout, arith='fp16x3', mode=N.MODE_UP3, plane_stride=ps)
                # (padded planes are INTERLEAVED since round 4: [B, cout, ps, px, py] -> [B, cout, phase 2*py+px, position])
                planar = c.view(B, cout, ps, 2, 2).permute(0, 1, 4, 3, 2).reshape(B, cout, 4, ps)[..., :rp]
                assert torch.equal(planar, a.view(B, cout, 4, rp)), ('up padded', B, cin, cout, H)
        else:
            C, co = cin, cout          # planes have C channels, dL/dx has co
            if co % 128 or not F_.split_ok(B, C, co, H, H, N.MODE_DOWN3):
                continue
            wd = torch.randn(1, C, co, 3, 3, device='cuda')
            gT = torch.randn(B, C, 4, H + 1, H + 1, device='cuda')
            dd = torch.rand(B, C, device='cuda') + 0.5
            ref = F_.modconv_raw(gT, F_.prepack_t(wd, flip=False), dd, None, co, N.MODE_DOWN3, H, H)
            a = F_.modconv_split(F_.planes_to_split(gT, dd, 'bf16x3'), F_.prepack_split(wd, 'bf16x3', adjoint='down'), None, None, co,
                                 mode=N.MODE_DOWN3, arith='bf16x3', x_split=(B, C, H, H), batch=B)
            e = rel(a, ref)
        tol = 3e-4 if kind == 'down' else 2e-5
        assert e <= tol, (kind, B, cin, cout, H, e)
        worst[kind] = max(worst[kind], e); done[kind] += 1
    # the persistent-block launches (>= 12 tiles per CU): big images, few channels, ragged batch sizes
    big = 0
    for _ in range(6 if n >= 30 else 0):
        H = rng.choice([128, 256])
        cout = rng.choice([64, 128]); cin = 16 * rng.randint(1, 4)
        tiles_per_img = H * H // (256 if cout % 128 == 0 else 512)
        B = (3072 + tiles_per_img - 1) // tiles_per_img + rng.randint(0, 6)
        x = torch.randn(B, cin, H, H, device='cuda'); w = torch.randn(1, cout, cin, 3, 3, device='cuda')
        s = torch.randn(B, cin, device='cuda'); d = torch.rand(B, cout, device='cuda') + 0.5
        nz = torch.randn(1, 1, H, H, device='cuda'); nw = torch.full((1,), 0.1, device='cuda'); bias = torch.randn(cout, device='cuda')
        sn = torch.randn(B, cout, device='cuda')
        wsp = F_.prepack_split(w, 'fp16x3')
        ya, _, xa = F_.modconv_split(x, wsp, s, d, cout, nz, nw, bias, True, arith='fp16x3', s_next=sn)
        yb, _, xb = F_.modconv_split(F_.to_split(x, s, 'fp16x3'), wsp, None, d, cout, nz, nw, bias, True, arith='fp16x3', s_next=sn,
                                     x_split=tuple(x.shape), batch=B)
        assert torch.equal(ya, yb) and torch.equal(xa, xb), ('persistent', B, cin, cout, H)
        big += 1
        del x, ya, yb, xa, xb
    print('fuzz ok:', done, 'persistent launches', big, 'worst rel err', {k: '%.1e' % v for k, v in worst.items()})

if __name__ == '__main__':
    main()
