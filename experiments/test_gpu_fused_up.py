"""GPU (run explicitly: `python experiments/build.py && python -m pytest experiments -m gpu`): the two shelved kernels against the
product path.  The upsampling StyledConv as ONE launch (experiments/csrc/upfir.hip: transposed conv with the FIR blur, noise, bias, leaky-ReLU and the
split hand-over in its epilogue; reference model.py:246-257 + 303-337) against the two-pass form it replaces and against the
fp64 oracle."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import O, S, maxabs      # noqa: E402
import fused_up as X               # noqa: E402

pytestmark = pytest.mark.gpu


def _fir():
    k = torch.tensor([1., 3., 3., 1.])
    k = torch.outer(k, k)
    return (k / k.sum() * 4).cuda()


def _inputs(tag, B, cin, cout, H, W, per_sample_noise=False):
    x = S.counter_tensor(21, tag + '.x', (B, cin, H, W)).cuda()
    w = S.counter_tensor(21, tag + '.w', (1, cout, cin, 3, 3)).cuda()
    s = S.counter_tensor(21, tag + '.s', (B, cin), 1.0, 0.3).cuda()
    d = S.counter_tensor(21, tag + '.d', (B, cout), 1.0, 0.2).abs().cuda() + 0.1
    sn = S.counter_tensor(21, tag + '.sn', (B, cout), 1.0, 0.3).cuda()
    nz = S.counter_tensor(21, tag + '.n', (B if per_sample_noise else 1, 1, 2 * H, 2 * W)).cuda()
    nw = torch.full((1,), 0.3).cuda()
    bias = S.counter_tensor(21, tag + '.b', (cout,), 0.0, 0.1).cuda()
    return x, w, s, d, sn, nz, nw, bias


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 32, 64, 16, 16), (3, 64, 128, 20, 12), (1, 128, 64, 128, 128), (2, 16, 64, 37, 5),
                                            (5, 32, 64, 8, 64)])
@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3'])
def test_fused_up_conv_equals_the_two_pass_form_bit_for_bit(B, cin, cout, H, W, arith):
    """Same MFMA sums per super-pixel (the K loop is the deep transposed plan's), same 16-tap order in the FIR, same epilogue
    arithmetic: the int16 hand-over buffers must be identical, for square / ragged shapes, shared and per-sample noise, and
    patches that hang over the right and bottom edges."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if not X.upfir_ok(B, cin, cout, H, W) or not F_.split_ok(B, cin, cout, H, W, F_.N.MODE_UP3):
        pytest.skip('shape not supported by one of the two forms')
    for per_sample in (False, True):
        x, w, s, d, sn, nz, nw, bias = _inputs('uf%d' % per_sample, B, cin, cout, H, W, per_sample)
        wsp = F_.prepack_split(w, arith)
        xs = F_.to_split(x, s, arith)
        with F_.using(F_.config().replace(use_splitk=False)):          # (K slices add their partial sums in another order)
            if F_.xin_ok(B, cin, cout, H, W, F_.N.MODE_UP3):
                planes = F_.modconv_split(xs, wsp, None, d, cout, mode=F_.N.MODE_UP3, x_split=(B, cin, H, W), arith=arith)
            else:           # tiling plans without a pre-split variant convert x*s in the kernel: the same bits
                planes = F_.modconv_split(x, wsp, s, d, cout, mode=F_.N.MODE_UP3, arith=arith)
        want = F_.blur_bias_act_split(planes, _fir(), H, W, sn, nz, nw, bias, True, arith=arith)
        word = F_.new_saturation_word(x.device)
        with F_.saturation_sink(word):
            got = X.modconv_upfir_split(xs, (B, cin, H, W), wsp, d, cout, _fir(), sn, nz, nw, bias, True, arith=arith)
        torch.cuda.synchronize()
        assert got.shape == want.shape and int(word.item()) == 0
        same = torch.equal(got, want)
        if not same:
            bad = (got != want).nonzero()
            print('first mismatches (b, group, part, pixel, ch):', bad[:8].tolist(), 'count', bad.shape[0], 'of', got.numel())
        assert same


def test_fused_up_conv_matches_fp64_oracle():
    """The layer against the oracle's StyledConv (upsample) in fp64: the same 2e-5 relative bound as the split kernels."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    B, cin, cout, H = 2, 64, 64, 32
    x, w, s, d_unused, sn, nz, nw, bias = _inputs('ufo', B, cin, cout, H, H)
    # the oracle's own layer (model.py:232-273 restated) in fp64, with a modulation that maps the "style" s to itself
    sd = s.double().cpu()
    eye = torch.eye(cin, dtype=torch.float64) * cin ** 0.5
    y = O.modulated_conv2d(x.double().cpu(), sd, w.double().cpu(), eye, torch.zeros(cin, dtype=torch.float64), demodulate=True, upsample=True)
    ref = O.fused_leaky_relu(y + float(nw) * nz.double().cpu(), bias.double().cpu())
    wc = (w[0] / (cin * 9) ** 0.5).double().cpu()
    dd = torch.rsqrt(torch.einsum('bi,oi->bo', sd ** 2, (wc ** 2).sum((2, 3))) + 1e-8)
    for arith, tol in (('fp16x3', 2e-5), ('bf16x3', 1e-4)):
        wsp = F_.prepack_split(w, arith)
        s_r, d_r = (F_.split_range(s, dd.float().cuda(), F_.absmax(x)) if arith == 'fp16x3' else (s, dd.float().cuda()))
        xs = F_.to_split(x, s_r, arith)
        ones = torch.ones(B, cout).cuda()
        got = X.modconv_upfir_split(xs, (B, cin, H, H), wsp, d_r, cout, _fir(), ones, nz, nw, bias, True, arith=arith)
        # decode the hand-over: hi + lo terms (fp16: times 2^4, the static activation pre-scale of the split form)
        dt = torch.float16 if arith == 'fp16x3' else torch.bfloat16
        val = got.view(dt).float()
        val = (val[:, :, 0] + val[:, :, 1]) * (16.0 if arith == 'fp16x3' else 1.0)            # [B, C/8, OHW, 8]
        y = val.permute(0, 1, 3, 2).reshape(B, cout, 2 * H, 2 * H)
        err = maxabs(y, ref) / float(ref.abs().max())
        print('%s fused up conv vs fp64: rel %.2e' % (arith, err))
        assert err <= tol


@pytest.mark.parametrize('B,cin,cout,H,W', [(2, 32, 64, 16, 16), (3, 64, 128, 20, 12), (1, 128, 64, 128, 128), (5, 32, 64, 8, 64),
                                            (7, 48, 192, 33, 17), (64, 64, 64, 32, 32)])
@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x3'])
def test_role_swapping_transposed_conv_writes_the_same_planes(B, cin, cout, H, W, arith):
    """experiments/csrc/uppp.hip (two wave groups per block: one runs a tile's MFMAs while the other DMAs its next operands and stores its own
    finished tile) against split.hip's transposed conv on the same pre-split input: identical interleaved planes, for tiles that
    straddle images, ragged last tiles, blocks with an odd number of tiles and grids smaller than the tile count."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    ps = ((H + 1) * (W + 1) + 31) // 32 * 32
    if not X.up_pp_ok(B, cin, cout, H, W, ps) or not F_.xin_ok(B, cin, cout, H, W, F_.N.MODE_UP3):
        pytest.skip('shape outside one of the two kernels')
    x, w, s, d, sn, nz, nw, bias = _inputs('pp', B, cin, cout, H, W)
    wsp = F_.prepack_split(w, arith)
    xs = F_.to_split(x, s, arith)
    with F_.using(F_.config().replace(use_splitk=False)):
        want = F_.modconv_split(xs, wsp, None, d, cout, mode=F_.N.MODE_UP3, x_split=(B, cin, H, W), arith=arith, plane_stride=ps)
    got = X.modconv_up_pp(xs, (B, cin, H, W), wsp, d, cout, ps, arith=arith)
    torch.cuda.synchronize()
    rp = (H + 1) * (W + 1)
    a, b = got.view(B, cout, ps, 4)[:, :, :rp], want.view(B, cout, ps, 4)[:, :, :rp]
    same = torch.equal(a, b)
    if not same:
        bad = (a != b).nonzero()
        print('mismatches (b, cout, pos, phase):', bad[:8].tolist(), bad.shape[0], 'of', a.numel())
    assert same
