"""1-D Winograd F(2,3) form of the plain split conv (csrc/wsplit.hip): the input transform B^T(x*s) and the weight transform
G W are applied in fp32 BEFORE the hi/lo split, so the kernel is held to the same per-layer bound as the direct split kernels
(2e-5 * scale vs the fp64 oracle in fp16x3, 1e-4 in bf16x3) -- reference ModulatedConv2d.forward model.py:232-273."""
import pytest
import torch

from util import S, maxabs

pytestmark = pytest.mark.gpu

CASES = [  # cin, cout, H, B
    (32, 128, 16, 3),      # 8 tile columns: one image per patch
    (64, 128, 32, 2),      # 16 tile columns = whole rows, 4 patches per image
    (48, 256, 64, 1),      # 2 patches across, 3 channel blocks, 2 cout tiles
    (16, 128, 128, 2),     # 4 patches across
    (16, 128, 256, 1),
    (128, 384, 32, 5),
    (32, 256, 64, 40),     # 1280 blocks: persistent blocks, 5 tiles each, every tile but the last stages its successor
    (48, 128, 64, 40),     # odd number of channel blocks: persistent without staging ahead (the ring does not come round)
    (16, 128, 128, 36),    # 2304 blocks: 9 tiles per block, first-round start spread
]
TOL = {'fp16x3': 2e-5, 'bf16x3': 1e-4}


def _oracle(x, w, s, d, noise, nw, bias):
    x, w, s, d = x.double().cpu(), w.double().cpu(), s.double().cpu(), d.double().cpu()
    cin = x.shape[1]
    y = torch.nn.functional.conv2d(x * s[:, :, None, None], w[0] / (cin * 9) ** 0.5, padding=1) * d[:, :, None, None]
    y = y + nw.double().cpu() * noise.double().cpu() + bias.double().cpu().view(1, -1, 1, 1)
    return torch.nn.functional.leaky_relu(y, 0.2) * 2 ** 0.5


def _inputs(cin, cout, H, B, tag='wsplit'):
    key = '%s.%d.%d.%d.%d' % (tag, cin, cout, H, B)
    w = S.counter_tensor(5, key + '.w', (1, cout, cin, 3, 3)).cuda()
    x = S.counter_tensor(5, key + '.x', (B, cin, H, H)).cuda()
    s = S.counter_tensor(5, key + '.s', (B, cin), 1.0, 0.3).cuda()
    d = S.counter_tensor(5, key + '.d', (B, cout), 1.0, 0.2).cuda()
    noise = S.counter_tensor(5, key + '.n', (1, 1, H, H)).cuda()
    nw = torch.full((1,), 0.1).cuda()
    bias = S.counter_tensor(5, key + '.b', (cout,), 0.0, 0.1).cuda()
    return w, x, s, d, noise, nw, bias


# F(4,3) multiplies by interpolation-point powers up to 8 and subtracts: its rounding error is ~3x F(2,3)'s (measured below);
# the per-layer bound is the one the fp32 Winograd F(2x2,3x3) kernel of wino.hip is held to
TOL4 = {'fp16x3': 6e-5, 'bf16x3': 4e-4}


@pytest.mark.parametrize('f', [2, 4])
@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3'])
@pytest.mark.parametrize('cin,cout,H,B', CASES)
def test_wsplit_conv_matches_fp64_oracle(cin, cout, H, B, arith, f):
    from stylegan_directions_face_reenactment_amd import functional as F_
    w, x, s, d, noise, nw, bias = _inputs(cin, cout, H, B)
    assert F_.N.load().sgdfr_modconv2d_wsplit_supported(B, cin, cout, H, H, f)
    vs = F_.to_wsplit(x, s, arith, f=f)
    y = F_.modconv_wsplit(vs, (B, cin, H, H), F_.prepack_wsplit(w, arith, f=f), d, cout, noise, nw, bias, True, arith=arith, f=f)
    ref = _oracle(x, w, s, d, noise, nw, bias)
    err = maxabs(y, ref)
    print('F(%d,3) %s %d->%d @%d: %.2e of scale' % (f, arith, cin, cout, H, err / max(1.0, float(ref.abs().max()))))
    assert err <= (TOL if f == 2 else TOL4)[arith] * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize('cin,cout,H,W,B,persist', [(64, 128, 16, 32, 2, 0), (128, 256, 32, 64, 3, 0), (256, 256, 64, 64, 5, 8), (96, 128, 32, 256, 2, 8)])
def test_fp8_cross_terms_on_the_wide_kernel(cin, cout, H, W, B, persist, monkeypatch):
    """SGDFR_SPLIT_FP16F8 (fp16 main term, both cross terms in e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4): the layer against the fp64
    oracle -- 1.2e-4 of max|y| (measured 4.6-5.6e-5; three fp16 products: 4e-6, bound 6e-5) with inputs as loud as the generator's
    range plan leaves them (calibrated maximum near 2^10 in the fp16 domain, single images up to 2^4 quieter) -- the same fused
    outputs as the fp16 form (hand-over in fp16 pairs, ToRGB partial sums), one tile per block and persistent blocks whose
    channel-block pairs run across tiles; and a shape the wide kernel cannot take fails loudly instead of reading fp8 chunks as fp16."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    key = 'f8.%d.%d.%d.%d.%d' % (cin, cout, H, W, B)
    w = S.counter_tensor(5, key + '.w', (1, cout, cin, 3, 3)).cuda()
    loud = 2.0 ** (12 - 4 * torch.arange(B, dtype=torch.float32) / max(B - 1, 1)).view(B, 1, 1, 1).cuda()
    x = S.counter_tensor(5, key + '.x', (B, cin, H, W)).cuda() * loud
    s = S.counter_tensor(5, key + '.s', (B, cin), 1.0, 0.3).cuda()
    d = (S.counter_tensor(5, key + '.d', (B, cout), 1.0, 0.2).cuda() / loud.view(B, 1)).contiguous()
    nw = torch.full((1,), 0.1).cuda()
    bias = S.counter_tensor(5, key + '.b', (cout,), 0.0, 0.1).cuda()
    sn = S.counter_tensor(5, key + '.sn', (B, cout), 1.0, 0.3).cuda()
    rgb = (S.counter_tensor(5, key + '.rw', (3, cout)).cuda(), S.counter_tensor(5, key + '.rs', (B, cout), 1.0, 0.3).cuda())
    noise = S.counter_tensor(5, key + '.n', (1, 1, H, W)).cuda()
    if persist:
        monkeypatch.setenv('SGDFR_WSPLIT_PERSIST', str(persist))
    if cin % 32:
        vs = F_.to_wsplit(x, s, 'fp16f8', f=4)
        wsp = F_.prepack_wsplit(w, 'fp16f8', f=4)
        with pytest.raises(RuntimeError, match='wide-tile kernel only'):
            F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith='fp16f8', f=4)
        return
    ref = _oracle(x, w, s, d, noise, nw, bias)
    out = {}
    for arith in ('fp16x3', 'fp16f8'):
        vs = F_.to_wsplit(x, s, arith, f=4)
        wsp = F_.prepack_wsplit(w, arith, f=4)
        monkeypatch.setenv('SGDFR_WSPLIT_WIDE_NOW', '2')
        y, part, xs = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith=arith, f=4, rgb=rgb, s_next=sn, want_y=True)
        torch.cuda.synchronize()
        out[arith] = (y, part, xs)
        err = ((y.double().cpu() - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3))).max().item()
        print(arith, 'worst image: max err / max|y| = %.2e' % err)
        assert err <= (6e-5 if arith == 'fp16x3' else 1.2e-4)
    y3, p3, x3 = out['fp16x3']
    y8, p8, x8 = out['fp16f8']
    scale = y3.abs().amax(dim=(1, 2, 3), keepdim=True)
    assert ((y8 - y3).abs() / scale).max().item() <= 1.2e-4 and not torch.equal(y8, y3)
    assert ((p8 - p3).abs().amax(dim=(1, 2, 3)) / p3.abs().amax(dim=(1, 2, 3))).max().item() <= 5e-4
    # the hand-over leaves as fp16 pairs in both (the next conv is a transposed one on the three-product kernels): same hi terms
    # wherever y agrees to the last fp16 bit -- compare the decoded values instead
    def decode(xs_):
        v = xs_.view(torch.float16).float()
        return v[:, :, 0] + v[:, :, 1]
    a, b = decode(x3), decode(x8)
    assert ((a - b).abs().amax(dim=(1, 2, 3)) / a.abs().amax(dim=(1, 2, 3))).max().item() <= 2e-4
    # SGDFR_SPLIT_HANDOVER_F8 (xs_arith='fp16f8'): the hand-over carries the fp8 cross-term operands of the NEXT conv -- the same
    # bits as to_split(y * s_next, 'fp16f8'), from either input arithmetic; y and the ToRGB sums do not change
    for arith in ('fp16x3', 'fp16f8'):
        vs = F_.to_wsplit(x, s, arith, f=4)
        wsp = F_.prepack_wsplit(w, arith, f=4)
        y, part, xs = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith=arith, f=4, rgb=rgb, s_next=sn, want_y=True,
                                        xs_arith='fp16f8')
        assert torch.equal(y, out[arith][0]) and torch.equal(part, out[arith][1])
        assert torch.equal(xs[:, :, 0], out[arith][2][:, :, 0])
        assert torch.equal(xs, F_.to_split(y, sn, 'fp16f8'))
    # ... and from the 64-tile kernel (the layers whose tile count keeps them there: 512 @ 16^2 at B=64)
    monkeypatch.setenv('SGDFR_WSPLIT_WIDE_NOW', '0')
    vs = F_.to_wsplit(x, s, 'fp16x3', f=4)
    wsp = F_.prepack_wsplit(w, 'fp16x3', f=4)
    y, part, xs = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith='fp16x3', f=4, rgb=rgb, s_next=sn, want_y=True,
                                    xs_arith='fp16f8')
    assert torch.equal(y, out['fp16x3'][0]) and torch.equal(xs, F_.to_split(y, sn, 'fp16f8'))


WIDE_CASES = [  # cin, cout, H, W, B, persist
    (64, 128, 16, 32, 1, 0),       # one patch per image, one tile per block, 4 channel blocks (the shortest ring wrap)
    (128, 256, 32, 32, 3, 0),      # two patches per image, two cout tiles
    (80, 128, 64, 64, 2, 0),       # odd number of channel blocks, 4 x 2 patches
    (128, 128, 128, 128, 2, 16),   # 64 tiles on 16 persistent blocks: four tiles per block, the ring runs across tiles
    (64, 384, 48, 96, 5, 8),       # ragged everything: 3 cout tiles, 3 x 3 patches, 135 tiles on 8 blocks (16 or 17 each)
    (256, 256, 64, 64, 40, 256),   # 640 tiles: the persistent grid of the bench shapes (2 or 3 tiles per block), start spread off
    (128, 128, 32, 256, 3, 8),     # 256-wide rows (ffhq-256's last plain layer): 8 patches across, 48 tiles on 8 blocks (a persistent
                                   # grid deals tiles to the 8 XCDs: it must have at least 8 blocks -- the product's has 256)
]


@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3'])
@pytest.mark.parametrize('cin,cout,H,W,B,persist', WIDE_CASES)
def test_wide_tile_kernel_writes_the_same_bits_as_the_64_tile_kernel(cin, cout, H, W, B, persist, arith, monkeypatch):
    """csrc/wswide.hip (128 couts x 128 tiles per block, position-outer K loop, output transform folded into the loop) against
    wsplit_kernel<., 6> on the same WS input and pack: y, the split hand-over and the fused ToRGB partial sums must be IDENTICAL
    -- every accumulator sums (channel block, kernel row, product term) in the same order and A^T M keeps its expression tree --
    with shared and per-sample noise, image borders inside and between patches (the zero rows come from the buffer load's bounds
    check), one tile per block and persistent blocks whose ring runs across tiles; and y within the F(4,3) bound of the oracle."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    key = 'wide.%d.%d.%d.%d.%d' % (cin, cout, H, W, B)
    w = S.counter_tensor(5, key + '.w', (1, cout, cin, 3, 3)).cuda()
    x = S.counter_tensor(5, key + '.x', (B, cin, H, W)).cuda()
    s = S.counter_tensor(5, key + '.s', (B, cin), 1.0, 0.3).cuda()
    d = S.counter_tensor(5, key + '.d', (B, cout), 1.0, 0.2).cuda()
    nw = torch.full((1,), 0.1).cuda()
    bias = S.counter_tensor(5, key + '.b', (cout,), 0.0, 0.1).cuda()
    sn = S.counter_tensor(5, key + '.sn', (B, cout), 1.0, 0.3).cuda()
    rgb = (S.counter_tensor(5, key + '.rw', (3, cout)).cuda(), S.counter_tensor(5, key + '.rs', (B, cout), 1.0, 0.3).cuda())
    vs = F_.to_wsplit(x, s, arith, f=4)
    wsp = F_.prepack_wsplit(w, arith, f=4)
    if persist:
        monkeypatch.setenv('SGDFR_WSPLIT_PERSIST', str(persist))
    for per_sample in (False, True):
        noise = S.counter_tensor(5, key + '.n%d' % per_sample, (B if per_sample else 1, 1, H, W)).cuda()
        for kw in ({}, {'s_next': sn, 'rgb': rgb, 'want_y': False}, {'s_next': sn, 'rgb': rgb, 'want_y': True}, {'rgb': rgb}):
            out = {}
            for wide in ('0', '2'):
                monkeypatch.setenv('SGDFR_WSPLIT_WIDE_NOW', wide)
                word = F_.new_saturation_word(x.device)
                with F_.saturation_sink(word):
                    r = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith=arith, f=4, **kw)
                torch.cuda.synchronize()
                r = r if isinstance(r, tuple) else (r,)
                out[wide] = [t.clone() if t is not None else None for t in r] + [int(word.item())]
            for a, b in zip(out['0'], out['2']):
                if isinstance(a, torch.Tensor):
                    same = torch.equal(a, b)
                    if not same:
                        bad = (a != b).nonzero()
                        print('mismatch', tuple(a.shape), a.dtype, 'first', bad[:6].tolist(), 'count', bad.shape[0], 'of', a.numel())
                    assert same, (kw.keys(), per_sample)
                else:
                    assert a == b
    monkeypatch.setenv('SGDFR_WSPLIT_WIDE_NOW', '2')
    noise = S.counter_tensor(5, key + '.n0', (1, 1, H, W)).cuda()
    y = F_.modconv_wsplit(vs, (B, cin, H, W), wsp, d, cout, noise, nw, bias, True, arith=arith, f=4)
    xd, wd = x.double().cpu() * s.double().cpu()[:, :, None, None], w[0].double().cpu() / (cin * 9) ** 0.5
    ref = torch.nn.functional.conv2d(xd, wd, padding=1) * d.double().cpu()[:, :, None, None]
    ref = torch.nn.functional.leaky_relu(ref + 0.1 * noise.double().cpu() + bias.double().cpu().view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    assert maxabs(y, ref) <= TOL4[arith] * max(1.0, float(ref.abs().max()))


def _decode_split(xs, arith):
    """[B, C/8, 2, HW, 8] int16 hi/lo -> fp32 [B, C, HW] (hi + lo)."""
    dt = torch.float16 if arith == 'fp16x3' else torch.bfloat16
    v = xs.view(dt).float()
    v = v[:, :, 0] + v[:, :, 1]                       # [B, C/8, HW, 8]
    B, G, HW, _ = v.shape
    return v.permute(0, 1, 3, 2).reshape(B, G * 8, HW)


@pytest.mark.parametrize('f', [2, 4])
@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3'])
def test_wsplit_chain_outputs_match_the_direct_split_kernel(arith, f):
    """xs_out (the next conv's split input) and the fused ToRGB partial sums, without y: same contract as modconv_split."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    cin, cout, H, B = 64, 256, 32, 3
    w, x, s, d, noise, nw, bias = _inputs(cin, cout, H, B, 'wsplit.chain')
    s_next = S.counter_tensor(5, 'wsplit.chain.sn', (B, cout), 1.0, 0.3).cuda()
    rgb_w = S.counter_tensor(5, 'wsplit.chain.rw', (3, cout)).cuda()
    rgb_s = S.counter_tensor(5, 'wsplit.chain.rs', (B, cout), 1.0, 0.3).cuda()
    tol = (TOL if f == 2 else TOL4)[arith]
    vs = F_.to_wsplit(x, s, arith, f=f)
    wsp = F_.prepack_wsplit(w, arith, f=f)
    y, part, xs = F_.modconv_wsplit(vs, (B, cin, H, H), wsp, d, cout, noise, nw, bias, True, arith=arith,
                                    rgb=(rgb_w, rgb_s), s_next=s_next, want_y=False, f=f)
    assert y is None
    ref = _oracle(x, w, s, d, noise, nw, bias)                                   # [B, cout, H, H] fp64
    scale = max(1.0, float(ref.abs().max()))
    xscale = 0.0625 if arith == 'fp16x3' else 1.0
    got = _decode_split(xs, arith).view(B, cout, H, H).cpu().double() / xscale
    want = ref * s_next.double().cpu()[:, :, None, None]
    assert maxabs(got, want) <= 2 * tol * max(1.0, float(want.abs().max()))
    rgb = part.view(B, cout // 128, 3, H, H).sum(1).cpu().double()
    want_rgb = torch.einsum('bchw,jc,bc->bjhw', ref, rgb_w.double().cpu(), rgb_s.double().cpu()) / cout ** 0.5
    assert maxabs(rgb, want_rgb) <= 4 * tol * max(1.0, float(want_rgb.abs().max()), scale)
    # y together with the other outputs is the same y
    y2, part2, xs2 = F_.modconv_wsplit(vs, (B, cin, H, H), wsp, d, cout, noise, nw, bias, True, arith=arith,
                                       rgb=(rgb_w, rgb_s), s_next=s_next, want_y=True, f=f)
    assert maxabs(y2, ref) <= tol * scale
    assert torch.equal(xs2, xs) and torch.equal(part2, part)


def test_wsplit_rejects_unsupported_shapes():
    from stylegan_directions_face_reenactment_amd import functional as F_
    lib = F_.N.load()
    for f in (2, 4):
        assert not lib.sgdfr_modconv2d_wsplit_supported(2, 64, 64, 32, 32, f)        # Cout % 128
        assert not lib.sgdfr_modconv2d_wsplit_supported(2, 64, 128, 8, 8, f)         # too narrow
        assert not lib.sgdfr_modconv2d_wsplit_supported(2, 24, 128, 32, 32, f)       # Cin % 16
        assert lib.sgdfr_modconv2d_wsplit_supported(64, 512, 512, 32, 32, f)
    assert not lib.sgdfr_modconv2d_wsplit_supported(64, 512, 512, 32, 32, 3)


@pytest.mark.parametrize('f', [2, 4])
@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3', 'fp16f8'])
@pytest.mark.parametrize('C,H,B', [(16, 8, 5), (64, 16, 3), (24, 32, 2), (16, 64, 2), (16, 128, 2)])
def test_blur_winograd_handover_is_bit_identical_to_transforming_the_fp32_result(C, H, B, arith, f):
    """sgdfr_blur_bias_act_split_f32(wino=f) == sgdfr_to_wsplit_f32(sgdfr_blur_bias_act_f32(...), s_next, f), dense and padded planes
    (fp16f8: the F(4,3) form whose lo chunks hold the e4m3 cross-term operands)."""
    if arith == 'fp16f8' and f != 4:
        pytest.skip('fp8 cross terms exist for the F(4,3) form only')
    from stylegan_directions_face_reenactment_amd import functional as F_
    key = 'wblur.%d.%d.%d' % (C, H, B)
    fir = torch.tensor([[1., 3., 3., 1.]]).cuda()
    fir = fir.t() @ fir
    fir = fir / fir.sum() * 4
    planes = S.counter_tensor(6, key + '.t', (B, C, 4, H + 1, H + 1)).cuda()
    noise = S.counter_tensor(6, key + '.n', (1, 1, 2 * H, 2 * H)).cuda()
    nw = torch.full((1,), 0.3).cuda()
    bias = S.counter_tensor(6, key + '.b', (C,), 0.0, 0.1).cuda()
    sn = S.counter_tensor(6, key + '.s', (B, C), 1.0, 0.3).cuda()
    if H == 128 and f != 4:
        pytest.skip('rows of two column tiles (W = 128): F(4,3) hand-over on interleaved planes only')
    y = F_.blur_bias_act(planes, fir, H, H, noise, nw, bias, True)
    want = F_.to_wsplit(y, sn, arith, f=f)
    if H < 128:
        got = F_.blur_bias_act_split(planes, fir, H, H, sn, noise, nw, bias, True, arith=arith, wino=f)
        assert got.shape == want.shape and torch.equal(got, want)
    ps = ((H + 1) * (H + 1) + 31) // 32 * 32
    # the padded form is interleaved: [B, C, positions, px, py] (include/sgdfr.h, plane_stride)
    il = torch.zeros(B, C, ps, 2, 2, device='cuda')
    il[:, :, :(H + 1) * (H + 1)] = planes.view(B, C, 2, 2, -1).permute(0, 1, 4, 3, 2)      # [.., py, px, pos] -> [.., pos, px, py]
    padded = il.view(B, C, 4, ps)
    got2 = F_.blur_bias_act_split(padded, fir, H, H, sn, noise, nw, bias, True, arith=arith, plane_stride=ps, wino=f)
    assert torch.equal(got2, want)
