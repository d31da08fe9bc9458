"""GPU parity of the whole generator path against (i) golden vectors captured from the real reference
(tests/golden/kat3..5) and (ii) the CPU oracle on full tensors.  Bar from BASELINE.json: max-abs <= 1e-3 on
fp32 images whose magnitude reaches ~10 with the synthetic weights; asserted tighter (2e-4) here."""
import numpy as np
import pytest
import torch

from util import O, S, SEED, golden, hip_generator, image_digest, maxabs, synthetic_state, t


@pytest.fixture(autouse=True, params=['fp16x3', 'fp32'])
def _both_arithmetics(request):
    """Every generator-level check runs twice: default split-fp16 conv kernels and the fp32 MFMA / Winograd kernels."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    with F_.precision(request.param):
        yield

pytestmark = pytest.mark.gpu

IMG_TOL = 2e-4      # |image| <= ~10, 14 layers of K<=4608 fp32 accumulations; north_star allows 1e-3


def _check_digest(img, g, prefix, tol=IMG_TOL):
    d = image_digest(img)
    assert np.abs(d['sub'] - g[prefix + '.sub']).max() <= tol
    # a row/col sum adds 256 values: allow sqrt(256)*tol-ish drift
    assert np.abs(d['rowsum'] - g[prefix + '.rowsum']).max() <= 40 * tol
    assert np.abs(d['colsum'] - g[prefix + '.colsum']).max() <= 40 * tol


@pytest.mark.parametrize('size', [32, 64])
def test_small_generators_all_layers(size):
    """KAT-3: every StyledConv / ToRGB output of Generator(32) and Generator(64) vs the reference's probes."""
    g = golden('kat3_small_generators.npz')
    G = hip_generator(size, 1)
    feats = {}
    from stylegan_directions_face_reenactment_amd.model import StyledConv, ToRGB
    hooks = [m.register_forward_hook(lambda mod, i, o, nm=nm: feats.__setitem__(nm, o.detach()))
             for nm, m in G.named_modules() if isinstance(m, (StyledConv, ToRGB))]
    w = t(g['g%d.w' % size]).cuda()
    with torch.no_grad():
        img, lat = G([w], input_is_latent=True)
    for h in hooks:
        h.remove()
    assert lat is None
    assert maxabs(img, t(g['g%d.image' % size])) <= IMG_TOL
    assert len(feats) == G.num_layers + G.log_size - 1
    for nm, f in feats.items():
        idx = torch.from_numpy(g['g%d.%s.probe_idx' % (size, nm)])
        got = f.reshape(-1).cpu()[idx]
        assert maxabs(got, t(g['g%d.%s.probe' % (size, nm)])) <= 1e-4, nm
        stats = g['g%d.%s.stats' % (size, nm)]
        assert abs(float(f.double().mean()) - stats[0]) <= 1e-5
        assert abs(float(f.double().abs().mean()) - stats[1]) <= 1e-5


@pytest.mark.parametrize('cm', [1, 2])
def test_generator256_golden_and_oracle(cm):
    """KAT-4 (+KAT-7): z path and W+ path with psi=0.7 and an explicit truncation latent, and the
    synthesis-only (psi=1) W+ path of bench config 2; channel_multiplier 1 (voxceleb) and 2 (ffhq)."""
    g = golden('kat4_generator256.npz')
    G = hip_generator(256, cm)
    assert sum(p.numel() for p in G.parameters()) == int(g['n_params_cm%d' % cm])
    z = S.synthetic_z(SEED, 2, key='kat4.z').cuda()
    ztr = S.synthetic_z(SEED, 64, key='kat4.ztrunc').cuda()
    w = S.synthetic_latents(SEED, 2, key='kat4.w').cuda()
    with torch.no_grad():
        trunc = G.style(ztr).mean(0, keepdim=True)                 # mean_latent with an injected z batch
        assert maxabs(trunc, t(g['cm%d.trunc' % cm])) <= 1e-5
        img_z, lat_z = G([z], return_latents=True, truncation=0.7, truncation_latent=trunc)
        assert lat_z.shape == (2, 14, 512)
        assert maxabs(lat_z, t(g['cm%d.lat_z' % cm])) <= 1e-5
        _check_digest(img_z, g, 'cm%d.z' % cm)
        img_w, _ = G([w], truncation=0.7, truncation_latent=trunc, input_is_latent=True)
        _check_digest(img_w, g, 'cm%d.w' % cm)
        img_p, _ = G([w], input_is_latent=True)
        _check_digest(img_p, g, 'cm%d.p' % cm)
    if cm == 1:
        assert maxabs(img_p[0], t(g['cm1.p.full0'])) <= IMG_TOL     # one full image from the real reference
    # full tensors vs the oracle, both channel multipliers, synthesis-only and the truncated W+ path
    ref, _ = O.generator_forward(synthetic_state(256, cm), [w.cpu()], input_is_latent=True)
    assert maxabs(img_p, ref) <= IMG_TOL
    ref_w, _ = O.generator_forward(synthetic_state(256, cm), [w.cpu()], input_is_latent=True, truncation=0.7,
                                   truncation_latent=trunc.cpu())
    assert maxabs(img_w, ref_w) <= IMG_TOL


def test_generate_image_with_direction_shift():
    """KAT-5: generate_image + DirectionMatrix shift, z path and W+ path, W-space shift over 8 layers."""
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    g4, g = golden('kat4_generator256.npz'), golden('kat5_generate_image.npz')
    G = hip_generator(256, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda()
    trunc = t(g4['cm1.trunc']).cuda()
    z = S.synthetic_z(SEED, 2, key='kat4.z').cuda()
    w = S.synthetic_latents(SEED, 2, key='kat4.w').cuda()
    sv = t(g['sv']).cuda()
    with torch.no_grad():
        for path, code, is_lat in (('z', z, False), ('w', w, True)):
            img, lat = generate_image(G, code, 0.7, trunc, shift_code=A(sv), input_is_latent=is_lat,
                                      return_latents=True)
            assert maxabs(lat, t(g['%s.latent' % path])) <= 2e-5
            _check_digest(img, g, path)
        img = generate_image(G, w, 0.7, trunc, w_plus=False, num_layers_shift=8, shift_code=t(g['shift_w']).cuda(),
                             input_is_latent=True)
        _check_digest(img, g, 'wshift')
        assert not isinstance(img, tuple)


def test_batch_independence_at_bench_size():
    """Size-independent property at the bench configuration (B=64, 256x256, cm=1): every image depends only on
    its own latent row, so image i of a 64-batch equals image i computed alone / in a 2-batch."""
    G = hip_generator(256, 1)
    w = S.synthetic_latents(11, 64, key='prop.w').cuda()
    with torch.no_grad():
        big, _ = G([w], input_is_latent=True)
        assert big.shape == (64, 3, 256, 256) and torch.isfinite(big).all()
        for sl in (slice(0, 2), slice(31, 33), slice(63, 64)):
            small, _ = G([w[sl].contiguous()], input_is_latent=True)
            # not bitwise: the dispatcher may pick the Winograd kernel for the big batch and the direct one for the
            # small batch (block-count heuristic); both are fp32 and agree to rounding
            assert maxabs(big[sl], small) <= 1e-4
        ref, _ = O.generator_forward(synthetic_state(256, 1), [w[62:64].cpu()], input_is_latent=True)
        assert maxabs(big[62:64], ref) <= IMG_TOL


def test_fp8_cross_terms_stay_inside_the_contract_at_bench_size():
    """Config.cross_terms='fp8' (SGDFR_SPLIT_FP16F8: at B=64 the three F(4,3) layers that take the wide-tile kernel, the four
    transposed convs after F(4,3) layers and the last direct plain layer keep their two cross terms in e4m3, 2 MFMA units per product instead of
    3): the images stay within the north-star bar of 1e-3 of the oracle (measured 2.3e-4 at |image| <= 8, against 1.4e-5 for three fp16 products) and within 5e-4 of
    the default arithmetic; the plan really takes the fp8 form for those layers, nothing clamps, and a small batch (no wide-tile
    launches) is untouched."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if F_.config().precision != 'fp16x3':
        pytest.skip('cross_terms only exists for the fp16x3 arithmetic')
    G = hip_generator(256, 1)
    w = S.synthetic_latents(11, 64, key='prop.w').cuda()
    layers = [G.conv1] + list(G.convs)
    with torch.no_grad():
        base, _ = G([w], input_is_latent=True)
        with F_.using(F_.config().replace(cross_terms='fp8')):
            plan = G._chain_plan(64, True, [object()] * len(layers), layers)
            assert [p[5] for p in plan].count('fp16f8') == 4 and [p[6] for p in plan].count('fp16f8') == 4
            word = F_.new_saturation_word(w.device)
            with F_.saturation_sink(word):
                img, _ = G([w], input_is_latent=True)
            torch.cuda.synchronize()
            assert int(word.item()) == 0
            small, _ = G([w[:2].contiguous()], input_is_latent=True)
        assert torch.isfinite(img).all()
        assert 1e-6 < maxabs(img, base) <= 5e-4                      # it IS another arithmetic, and a close one
        assert maxabs(small, base[:2]) <= 1e-4                       # B=2: no layer takes the fp8 form
        ref, _ = O.generator_forward(synthetic_state(256, 1), [w[62:64].cpu()], input_is_latent=True)
        assert maxabs(img[62:64], ref) <= 5e-4                       # (north-star bar: 1e-3)
        assert maxabs(base[62:64], ref) <= IMG_TOL
        # A batch louder than the fp8 operands' range but NOT than the fp16 terms' (one noise map x 2^11: that layer's input lands between
        # 2^12.8 and 2^16 of the fp16 domain): the clamped cross-term halves are counted like clamped fp16 pairs, so the (verifying)
        # call measures the batch, widens the plan and renders again -- instead of handing back single-fp16 cross terms.
        G2 = hip_generator(256, 1)
        with F_.using(F_.config().replace(cross_terms='fp8')):
            G2([w], input_is_latent=True)
            plan0 = list(G2._range_state['x_log2'])
            loud = [getattr(G2.noises, 'noise_%d' % i) * (2.0 ** 11 if i == 8 else 1.0) for i in range(G2.num_layers)]
            got, _ = G2([w], input_is_latent=True, noise=loud)
            assert G2.saturated_pairs() > 0 and G2.range_mode() == 'fp16x3'
            plan1 = list(G2._range_state['x_log2'])
            assert max(b - a for a, b in zip(plan0, plan1)) >= 2 and all(b >= a for a, b in zip(plan0, plan1))
        ref_loud, _ = O.generator_forward(synthetic_state(256, 1), [w[62:64].cpu()], input_is_latent=True, noise=[n.cpu() for n in loud])
        scale = max(1.0, float(ref_loud.abs().max()) / 8)
        print('loud fp8 batch: %.2e vs the oracle (|image| <= %.1f), plan widened by %d binades' % (
            maxabs(got[62:64], ref_loud), float(ref_loud.abs().max()), max(b - a for a, b in zip(plan0, plan1))))
        assert maxabs(got[62:64], ref_loud) <= 5e-4 * scale


def test_noise_modes_and_truncation_quirks():
    G = hip_generator(64, 1)
    w = S.synthetic_latents(12, 4, n_latent=G.n_latent, key='nz.w').cuda()
    P = synthetic_state(64, 1)
    with torch.no_grad():
        a, _ = G([w], input_is_latent=True)
        b, _ = G([w], input_is_latent=True)
        assert maxabs(a, b) == 0.0                                   # fixed noise buffers by default (model.py:488-492)
        c, _ = G([w], input_is_latent=True, randomize_noise=True)
        assert c.shape == a.shape and torch.isfinite(c).all() and maxabs(a, c) > 1e-3
        ns = G.make_noise()
        assert [tuple(n.shape) for n in ns] == [tuple(getattr(G.noises, 'noise_%d' % i).shape) for i in range(G.num_layers)]
        d, _ = G([w], input_is_latent=True, noise=ns)
        ref, _ = O.generator_forward(P, [w.cpu()], input_is_latent=True, noise=[n.cpu() for n in ns])
        assert maxabs(d, ref) <= 2e-4
        # truncation is applied to a full W+ code too (model.py:494-500), by broadcasting trunc [1,512]
        tr = S.counter_tensor(12, 'nz.t', (1, 512)).cuda()
        e, lat = G([w], input_is_latent=True, truncation=0.5, truncation_latent=tr, return_latents=True)
        assert maxabs(lat, tr + 0.5 * (w - tr)) <= 1e-6
        ref, _ = O.generator_forward(P, [w.cpu()], input_is_latent=True, truncation=0.5, truncation_latent=tr.cpu())
        assert maxabs(e, ref) <= 2e-4
        # a batch of one and an empty-noise list element
        f, _ = G([w[:1].contiguous()], input_is_latent=True)
        assert maxabs(f, a[:1]) <= 1e-5


def test_module_protocol_on_gpu():
    """deepcopy / state_dict round trip / weight-version repack, as optimize_g and load_models use them."""
    import copy
    G = hip_generator(32, 1)
    w = S.synthetic_latents(13, 2, n_latent=G.n_latent, key='mp.w').cuda()
    with torch.no_grad():
        a, _ = G([w], input_is_latent=True)
        G2 = copy.deepcopy(G)
        b, _ = G2([w], input_is_latent=True)
        assert maxabs(a, b) == 0.0
        G2.convs[0].conv.weight.mul_(1.5)                         # in-place update must invalidate the packed copy
        c, _ = G2([w], input_is_latent=True)
        P = {k: v.cpu() for k, v in G2.state_dict().items()}
        ref, _ = O.generator_forward(P, [w.cpu()], input_is_latent=True)
        assert maxabs(c, ref) <= 2e-4


def test_batched_reenactment_equals_per_frame_loop():
    """N frames in batches == the reference's one-generate_image-per-frame loop (run_inference.py:170-181),
    and the GPU uint8 conversion == the reference's tensor_to_image scaling + uint8 cast."""
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession, images_to_uint8
    G = hip_generator(64, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda()
    src = S.synthetic_latents(41, 1, n_latent=G.n_latent, key='re.src').cuda()
    trunc = S.counter_tensor(41, 're.t', (1, 512)).cuda()
    sv = S.counter_tensor(41, 're.sv', (7, 15), 0.0, 3.0).cuda()
    sess = ReenactmentSession(G, A, src, 0.7, trunc, batch=3)
    out = sess.render(sv)
    assert out.shape == (7, 3, 64, 64)
    with torch.no_grad():
        for i in range(7):
            ref = generate_image(G, src, 0.7, trunc, shift_code=A(sv[i:i + 1]), input_is_latent=True)
            assert maxabs(out[i:i + 1], ref) <= 1e-4     # B=1 takes the K-sliced kernels, the batch does not
    u8 = sess.render(sv, as_uint8=True)
    assert u8.shape == (7, 64, 64, 3) and u8.dtype == torch.uint8
    x = out.cpu().clone()
    x.clamp_(-1, 1).add_(1).div_(2 + 1e-5)
    expect = x.mul(255.0).numpy().transpose(0, 2, 3, 1).astype('uint8')
    diff = (u8.cpu().numpy().astype(int) - expect.astype(int))
    assert abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3      # identical up to fp32 rounding at integer boundaries
    assert images_to_uint8(torch.full((1, 3, 2, 2), 5.0).cuda()).max() == 254 and \
        images_to_uint8(torch.full((1, 3, 2, 2), -5.0).cuda()).max() == 0


def test_split_chain_is_bit_identical_to_layer_by_layer():
    """Inference dataflow of the split kernels (activations handed over only in the next conv's split form, ToRGB fused,
    last activation never stored) == the same kernels run layer by layer through fp32 activations, bit for bit.  (With the
    Winograd form of the wide plain layers switched off: that one changes the arithmetic, see the next test.)"""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if F_.PRECISION == 'fp32':
        pytest.skip('the chain exists only for the split arithmetics')
    for size, B in ((64, 5), (256, 9)):
        G = hip_generator(size, 1)
        w = S.synthetic_latents(SEED, B, n_latent=G.n_latent, key='chain.w').cuda()
        tr = S.counter_tensor(SEED, 'chain.t', (1, 512)).cuda()
        with torch.no_grad():
            G.config = F_.config().replace(use_wsplit=False)
            a, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
            G.config = F_.config().replace(use_wsplit=False, use_split_chain=False)
            b, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
        assert torch.equal(a, b)
        assert G.saturated_pairs() == 0 or F_.PRECISION != 'fp16x3'     # nothing of this generator hit the fp16 clamp


def test_cm2_generator_without_plane_padding_takes_the_direct_kernel_at_256():
    """the Winograd hand-over of 256-wide rows exists only on interleaved padded planes; with plane padding off the
    128 -> 128 @ 256^2 layer of a cm=2 generator must stay on the direct kernel (it raised before) and give the same image
    within the arithmetic's bound."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if F_.PRECISION == 'fp32':
        pytest.skip('the chain exists only for the split arithmetics')
    G = hip_generator(256, 2)
    w = S.synthetic_latents(SEED, 8, n_latent=G.n_latent, key='nopad.w').cuda()
    with torch.no_grad():
        a, _ = G([w], input_is_latent=True)
        G.config = F_.config().replace(use_plane_padding=False)
        assert F_.wsplit_chain_f(8, 128, 128, 256, 256) == 4
        with F_.using(G.config):
            assert F_.wsplit_chain_f(8, 128, 128, 256, 256) == 0 and F_.wsplit_chain_f(8, 256, 256, 128, 128) == 4
        b, _ = G([w], input_is_latent=True)
    assert maxabs(a, b) <= IMG_TOL


@pytest.mark.timeout(1200)
def test_range_plan_on_trained_like_weights():
    """the fp16x3 range plan (calibrated on the first batch of a weight version) on what a trained g_ema looks
    like -- heavy-tailed conv weights (1 % of the input channels x 30), log-normal modulation biases (synthetic.trained_like_state_dict)
    -- with W+ codes from the e4e stand-in on 200 batches of random images: how many verified forwards had to be rendered twice, and
    every returned frame against the fp32-MFMA kernels of the same generator (no range plan, no fp16), three of them against the
    fp64 oracle.  Bar: re-render rate < 1 % after the first widening, every frame within 1e-3 (images scaled to |img| <= ~10)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
    from stylegan_directions_face_reenactment_amd.model import Generator
    if F_.PRECISION == 'fp32':
        pytest.skip('the range plan belongs to the fp16x3 arithmetic')
    state = S.trained_like_state_dict(O.template_state(256, 512, 8, 1), seed=SEED)
    G = Generator(256, 512, 8, channel_multiplier=1)
    G.load_state_dict(state, strict=True)
    G = G.eval().cuda()
    enc = Encoder4Editing(50, 'ir_se', 256).eval()
    enc.load_state_dict(S.synthetic_encoder_state(enc.state_dict(), seed=SEED), strict=True)
    enc = enc.cuda()
    tr = S.counter_tensor(SEED, 'rp.t', (1, 512)).cuda()
    n_batches, B = 200, 8
    worst, scale, keep = 0.0, 0.0, []
    first_widening_at = None
    with torch.no_grad():
        for i in range(n_batches):
            # (every tenth batch: 4x louder codes -- an outlier source; the e4e stand-in's codes are N(0, ~1) per row)
            x = S.counter_tensor(SEED, 'rp.x.%d' % i, (B, 3, 256, 256), 0.0, 0.5).clamp_(-1, 1).cuda()
            w = enc(x) * (4.0 if i % 10 == 9 else 1.0)
            img, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)       # the reference-shaped call: verified
            with F_.precision('fp32'):
                ref, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
            worst = max(worst, maxabs(img, ref))
            scale = max(scale, float(ref.abs().max()))
            st = G.range_stats()
            if first_widening_at is None and st['widenings']:
                first_widening_at = i
            if i in (0, 9, 199):
                keep.append((w[:1].cpu(), img[:1].cpu()))
    st = G.range_stats()
    after = st['rerendered'] - (1 if first_widening_at is not None else 0)
    print('trained-like weights, %d batches of %d e4e codes: %d re-rendered (first widening at batch %s, %d widenings), mode %s; every frame '
          'within %.2e of the fp32 kernels (max |img| %.1f)' % (n_batches, B, st['rerendered'], first_widening_at, st['widenings'], st['mode'],
                                                              worst, scale))
    assert after / n_batches < 0.01, st
    bar = 1e-3 * max(1.0, scale / 10.0)          # (BASELINE's 1e-3 is quoted on |img| <= ~10)
    assert worst <= bar
    P64 = O.cast_state(state, torch.float64)
    for w1, img1 in keep:
        ref64, _ = O.generator_forward(P64, [w1.double()], input_is_latent=True, truncation=0.7, truncation_latent=tr.cpu().double())
        assert maxabs(img1, ref64) <= bar


def test_winograd_chain_layers_stay_within_the_per_image_bound():
    """The wide plain layers of the chain run in 1-D Winograd form (F(4,3) by default, F(2,3) with functional.WSPLIT_F = 2;
    csrc/wsplit.hip, fed by the blur's transformed hand-over): a different summation order, so not bit-identical to the direct
    split kernels -- both forms are held to the fp64 oracle with the direct chain's image bound."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if F_.PRECISION == 'fp32':
        pytest.skip('the chain exists only for the split arithmetics')
    size, B = 256, 16                                     # (smaller batches K-slice these layers: no fused ToRGB, no chain form)
    G = hip_generator(size, 1)
    w = S.synthetic_latents(SEED, B, n_latent=G.n_latent, key='wchain.w').cuda()
    P64 = O.cast_state(synthetic_state(size, 1), torch.float64)
    with torch.no_grad():
        ref, _ = O.generator_forward(P64, [w[:2].double().cpu()], input_is_latent=True)
        assert F_.USE_WSPLIT and F_.WSPLIT_F == 4
        a, _ = G([w], input_is_latent=True)
        used = sorted(G._wino_inputs(B, [G.conv1] + list(G.convs)))
        assert set(G._wino_inputs(B, [G.conv1] + list(G.convs)).values()) == {4}
        with F_.using(F_.config().replace(wsplit_f=2)):
            a2, _ = G([w], input_is_latent=True)
            assert G._wino_inputs(B, [G.conv1] + list(G.convs)) == {6: 2, 8: 2}
        with F_.using(F_.config().replace(use_wsplit=False)):
            b, _ = G([w], input_is_latent=True)
    assert used == [6, 8, 10], used                      # 512 @ 32^2, 256 @ 64^2, 128 @ 128^2 (512 @ 16^2 joins from B = 48)
    assert not torch.equal(a, b)
    bound = 2e-4 if F_.PRECISION == 'fp16x3' else 5e-4
    ea, ea2, eb = maxabs(a[:2], ref), maxabs(a2[:2], ref), maxabs(b[:2], ref)
    print('256^2 images vs fp64 oracle: F(4,3) chain %.2e, F(2,3) chain %.2e, direct chain %.2e' % (ea, ea2, eb))
    assert ea <= bound and ea2 <= bound and eb <= bound and not torch.equal(a, a2)
    assert G.saturated_pairs() == 0 or F_.PRECISION != 'fp16x3'


def test_graph_replay_with_winograd_layers_is_bit_identical():
    """A captured forward that contains wsplit launches (B=16 at 256^2: the 32^2 ... 128^2 plain layers) replays the same bits."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    if F_.PRECISION == 'fp32':
        pytest.skip('the chain exists only for the split arithmetics')
    G = hip_generator(256, 1)
    ws = [S.synthetic_latents(SEED, 16, n_latent=G.n_latent, key='wgraph.w%d' % i).cuda() for i in range(4)]
    with torch.no_grad():
        assert G._wino_inputs(16, [G.conv1] + list(G.convs))
        eager = [G([w], input_is_latent=True, graph=False)[0] for w in ws]
        got = [G([w], input_is_latent=True, graph=True)[0] for w in ws]
        assert len(G._graphs) == 1
        for a, b in zip(eager, got):
            assert torch.equal(a, b)


def test_independent_batches_on_alternating_streams_give_the_same_images():
    """functional.StreamPipeline (bench.py --streams, ReenactmentSession(streams=2)): consecutive independent batches on two
    HIP streams overlap in time; every image is bit-identical to the one-stream rendering, for raw forwards and for the
    session's verified chunks (uint8 video frames included)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession
    G = hip_generator(64, 1)
    G.use_graphs = False
    ws = [S.synthetic_latents(SEED, 5, n_latent=G.n_latent, key='pipe.w%d' % i).cuda() for i in range(6)]
    with torch.no_grad():
        want = [G([w], input_is_latent=True)[0] for w in ws]
        pipe = F_.StreamPipeline(2)
        got = []
        for w in ws:
            with pipe.next():
                got.append(G([w], input_is_latent=True)[0])
        pipe.join(*got)
        for a, b in zip(want, got):
            assert torch.equal(a, b)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda().eval()
    src = S.synthetic_latents(SEED, 1, n_latent=G.n_latent, key='pipe.src').cuda()
    trunc = S.counter_tensor(SEED, 'pipe.t', (1, 512)).cuda()
    sv = S.counter_tensor(SEED, 'pipe.sv', (23, 15), 0.0, 3.0).cuda()
    one = ReenactmentSession(G, A, src, 0.7, trunc, batch=4, streams=1)
    two = ReenactmentSession(G, A, src, 0.7, trunc, batch=4, streams=2)
    two.pipeline_min_work = 0
    assert torch.equal(one.render(sv), two.render(sv)) and two._pipe is not None and one._pipe is None
    assert torch.equal(one.render(sv, as_uint8=True), two.render(sv, as_uint8=True))
    live, got = two.streaming(), []                       # chunk by chunk, state kept between calls
    for lo in range(0, 23, 4):
        prev = live.push(sv[lo:lo + 4])
        assert (prev is None) == (lo == 0)
        if prev is not None:
            got.append(prev)
    got.append(live.flush())
    assert live.flush() is None and torch.equal(torch.cat(got, 0), one.render(sv))
    src_img = S.counter_tensor(SEED, 'pipe.si', (1, 3, 64, 64), 0.0, 0.5).cuda()
    tgt_img = S.counter_tensor(SEED, 'pipe.ti', (23, 3, 64, 64), 0.0, 0.5).cuda()
    assert torch.equal(one.video_frames(src_img, tgt_img, sv), two.video_frames(src_img, tgt_img, sv))


def test_generator_forward_replays_a_hipgraph_by_default():
    """from the third no-grad forward of one signature on, Generator.forward replays a captured hipGraph --
    bit-identical to the eager launches for new inputs, through generate_image too; a weight change, a different batch size,
    caller-supplied noise, hooks and grad mode fall back to eager launches (and never to stale results)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    G = hip_generator(64, 1)
    tr = S.counter_tensor(SEED, 'graph.t', (1, 512)).cuda()
    ws = [S.synthetic_latents(SEED, 3, n_latent=G.n_latent, key='graph.w%d' % i).cuda() for i in range(5)]
    with torch.no_grad():
        G.use_graphs = False
        eager = [G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr, return_latents=True) for w in ws]
        G.use_graphs = True
        got = [G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr, return_latents=True) for w in ws]
        assert len(G._graphs) == 1                                  # calls 0, 1 eager, call 2 captured, calls 3, 4 replayed
        for (a, la), (b, lb) in zip(eager, got):
            assert torch.equal(a, b) and torch.equal(la, lb)
        assert got[3][0].data_ptr() != got[4][0].data_ptr()         # results are clones, not the graph's static buffer
        # the reference's glue on top (verified forwards): same bits
        imgs = [generate_image(G, w, 0.7, tr, input_is_latent=True) for w in ws]
        assert all(torch.equal(a[0], b) for a, b in zip(eager, imgs))
        # uint8 frames out of the last ToRGB launch, graphed
        u8 = [G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr, image_out=F_.U8Target())[0] for w in ws]
        G.use_graphs = False
        u8e = G([ws[4]], input_is_latent=True, truncation=0.7, truncation_latent=tr, image_out=F_.U8Target())[0]
        G.use_graphs = True
        assert u8[4].dtype == torch.uint8 and torch.equal(u8[4], u8e)
        # another batch size: its own signature; caller-supplied noise and hooks: eager
        n_graphs = len(G._graphs)
        G.use_graphs = False
        small_eager, _ = G([ws[0][:1]], input_is_latent=True, truncation=0.7, truncation_latent=tr)
        G.use_graphs = True
        small, _ = G([ws[0][:1]], input_is_latent=True, truncation=0.7, truncation_latent=tr)
        assert torch.equal(small, small_eager) and len(G._graphs) == n_graphs
        assert maxabs(small, eager[0][0][:1]) <= 1e-5                # (another batch size = another tiling: equal to rounding)
        noises = [getattr(G.noises, 'noise_%d' % i) * 1.0 for i in range(G.num_layers)]
        a, _ = G([ws[1]], input_is_latent=True, truncation=0.7, truncation_latent=tr, noise=noises)
        assert torch.equal(a, eager[1][0])
        # new weights: the graphs are dropped, the next forwards are right (and re-captured later)
        G.convs[3].conv.weight.mul_(1.5)
        G.use_graphs = False
        want, _ = G([ws[2]], input_is_latent=True, truncation=0.7, truncation_latent=tr)
        G.use_graphs = True
        for _ in range(4):
            b, _ = G([ws[2]], input_is_latent=True, truncation=0.7, truncation_latent=tr)
            assert torch.equal(b, want)
        assert not torch.equal(want, eager[2][0]) and len(G._graphs) == 1
    # grad mode never replays
    w = ws[0].clone().requires_grad_(True)
    img, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
    img.square().mean().backward()
    assert w.grad is not None and bool(torch.isfinite(w.grad).all())
    # deepcopy / pickle leave the graphs behind
    import copy
    G2 = copy.deepcopy(G)
    assert '_graphs' not in G2.__dict__
    with torch.no_grad():
        assert torch.equal(G2([ws[2]], input_is_latent=True, truncation=0.7, truncation_latent=tr)[0], want)


def test_graphed_reenactment_session_is_bit_identical():
    """hipGraph replay of the per-batch step (small-batch latency path) == eager launches, incl. a ragged tail."""
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession
    G = hip_generator(64, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda()
    src = S.synthetic_latents(43, 1, n_latent=G.n_latent, key='gr.src').cuda()
    trunc = S.counter_tensor(43, 'gr.t', (1, 512)).cuda()
    sv = S.counter_tensor(43, 'gr.sv', (7, 15), 0.0, 3.0).cuda()
    eager = ReenactmentSession(G, A, src, 0.7, trunc, batch=2).render(sv)
    sess = ReenactmentSession(G, A, src, 0.7, trunc, batch=2, graph=True)
    first, second = sess.render(sv), sess.render(sv.flip(0))
    # (frames 0 and 6 are a one-frame tail in one of the two runs: other launch shapes, not compared bit for bit)
    assert torch.equal(first, eager) and torch.equal(second.flip(0)[1:6], eager[1:6])
    assert sess._graph is not None


def test_video_grid_frames_match_reference_packing():
    """source|target|reenacted uint8 frames for a whole batch in one launch == the reference's per-frame
    generate_grid_image + tensor_to_image + cvtColor + np.uint8 (utils_inference.py:11-33, run_inference.py:188-194)."""
    from stylegan_directions_face_reenactment_amd.reenact import grid_frames_uint8
    src = S.counter_tensor(5, 'grid.src', (1, 3, 16, 24), 0.0, 0.8)
    tgt = S.counter_tensor(5, 'grid.tgt', (6, 3, 16, 24), 0.0, 0.8)
    ren = S.counter_tensor(5, 'grid.ren', (6, 3, 16, 24), 0.0, 0.8)
    for swap in (True, False):
        got = grid_frames_uint8([src.cuda(), tgt.cuda(), ren.cuda()], swap_rb=swap)
        assert got.shape == (6, 16, 72, 3) and got.dtype == torch.uint8
        expect = O.grid_video_frames(src, tgt, ren, swap_rb=swap)
        diff = got.cpu().numpy().astype(int) - expect.astype(int)
        assert abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3      # fp32 rounding at integer boundaries only
    one = grid_frames_uint8([tgt.cuda()])
    assert (one.cpu().numpy().astype(int) - O.tensor_to_uint8_hwc(tgt).astype(int)).__abs__().max() <= 1
    with pytest.raises(RuntimeError):
        grid_frames_uint8([src.cuda()] * 5)
    with pytest.raises(RuntimeError):
        grid_frames_uint8([src.cuda(), tgt[:, :, :8].cuda()])


def test_wide_layers_and_1024_generator():
    """Resolutions above 256 (the reference also ships ffhq-1024, libs/configs/config_models.py:16-20, pooled to 256 by
    generate_image): wide rows switch the conv staging to row segments; checked per layer and end to end."""
    from stylegan_directions_face_reenactment_amd.model import Generator, StyledConv
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    for cin, cout, h, up in ((8, 32, 512, False), (16, 64, 1024, False), (8, 8, 256, True), (8, 16, 512, True)):
        key = 'wide.%d.%d.%d.%d' % (cin, cout, h, up)
        m = StyledConv(cin, cout, 3, 64, upsample=up)
        sd = {k: S.counter_tensor(51, key + k, tuple(v.shape)) for k, v in m.state_dict().items() if 'kernel' not in k}
        sd['conv.modulation.bias'] = sd['conv.modulation.bias'] * 0.1 + 1.0
        if up:
            sd['conv.blur.kernel'] = m.conv.blur.kernel
        m.load_state_dict(sd)
        x = S.counter_tensor(51, key + 'x', (1, cin, h, h))
        st = S.counter_tensor(51, key + 's', (1, 64))
        r = 2 * h if up else h
        nz = S.counter_tensor(51, key + 'n', (1, 1, r, r))
        ref = O.styled_conv({'L.' + k: v for k, v in sd.items()}, 'L', x, st, nz, upsample=up)
        with torch.no_grad():
            y = m.cuda()(x.cuda(), st.cuda(), noise=nz.cuda())
        assert maxabs(y, ref) <= 3e-5, key
    G = Generator(1024, 512, 8, channel_multiplier=2)
    P = S.synthetic_state_dict(O.template_state(1024, 512, 8, 2), seed=SEED)
    G.load_state_dict(P)
    G = G.eval().cuda()
    assert G.n_latent == 18
    w = S.synthetic_latents(52, 1, n_latent=18, key='g1024.w')
    torch.set_num_threads(8)
    with torch.no_grad():
        img, _ = G([w.cuda()], input_is_latent=True)
        ref, _ = O.generator_forward(P, [w], input_is_latent=True)
        assert img.shape == (1, 3, 1024, 1024)
        assert maxabs(img, ref) <= IMG_TOL
        small = generate_image(G, w.cuda(), 1.0, None, input_is_latent=True)       # pooled to 256 like the reference
        assert small.shape == (1, 3, 256, 256)
        assert maxabs(small, torch.nn.functional.adaptive_avg_pool2d(ref, (256, 256))) <= IMG_TOL


def test_uint8_frames_from_the_last_torgb_launch():
    """SURVEY §8f-4 as worded: the [-1,1] -> uint8 HWC conversion is part of the final ToRGB launch (no fp32 image is
    stored and re-read).  Bit-identical to converting the fp32 image afterwards, as plain frames and as one panel of the
    source|target|reenacted video grid (with and without the channel swap), incl. the unfused fallback at tiny batches."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession, grid_frames_uint8, images_to_uint8
    for size, B in ((256, 6), (64, 16), (64, 1)):
        G = hip_generator(size, 1)
        w = S.synthetic_latents(SEED, B, n_latent=G.n_latent, key='u8.w').cuda()
        with torch.no_grad():
            img, _ = G([w], input_is_latent=True)
            u8, _ = G([w], input_is_latent=True, image_out=F_.U8Target())
            assert u8.dtype == torch.uint8 and u8.shape == (B, size, size, 3)
            assert torch.equal(u8, images_to_uint8(img))
            for swap in (False, True):
                grid = torch.zeros(B, size, 3 * size, 3, dtype=torch.uint8, device='cuda')
                out, _ = G([w], input_is_latent=True, image_out=F_.U8Target(grid, panel=1, swap_rb=swap))
                assert out.data_ptr() == grid.data_ptr()
                expect = grid_frames_uint8([img * 0 - 1, img, img * 0 - 1], swap_rb=swap)     # black | image | black
                assert torch.equal(grid, expect)
    with pytest.raises(RuntimeError):
        G([w.requires_grad_(True)], input_is_latent=True, image_out=F_.U8Target())
    # the session's video path: reenacted panel from the generator, the other two from one grid launch
    G = hip_generator(64, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda()
    src_code = S.synthetic_latents(45, 1, n_latent=G.n_latent, key='u8.src').cuda()
    trunc = S.counter_tensor(45, 'u8.t', (1, 512)).cuda()
    sv = S.counter_tensor(45, 'u8.sv', (21, 15), 0.0, 3.0).cuda()
    src_img = S.counter_tensor(45, 'u8.si', (1, 3, 64, 64), 0.0, 0.6).cuda()
    tgt_img = S.counter_tensor(45, 'u8.ti', (21, 3, 64, 64), 0.0, 0.6).cuda()
    sess = ReenactmentSession(G, A, src_code, 0.7, trunc, batch=16)
    video = sess.video_frames(src_img, tgt_img, sv)
    ren = sess.render(sv)
    assert torch.equal(video, grid_frames_uint8([src_img, tgt_img, ren], swap_rb=True))
    assert torch.equal(sess.render(sv, as_uint8=True), images_to_uint8(ren))
