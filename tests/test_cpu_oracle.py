"""CPU: the oracle restatement (oracle/sg2_oracle.py) against the golden vectors captured from the REAL
reference by oracle/make_golden.py (KAT-1..7).  This is what pins the oracle wherever the repo travels."""
import numpy as np
import torch

from util import O, S, SEED, golden, image_digest, maxabs, synthetic_state, t


def test_kat1_upfirdn2d_and_fused_act():
    g = golden('kat1_ops.npz')
    x, k = t(g['x']), t(g['kernel'])
    for ci, (up, down, p0, p1) in enumerate(g['cases']):
        xr = x.clone().requires_grad_(True)
        y = O.upfirdn2d(xr, k, up=int(up), down=int(down), pad=(int(p0), int(p1)))
        assert y.shape == g['y%d' % ci].shape
        assert maxabs(y, t(g['y%d' % ci])) <= 2e-6
        (y * t(g['g%d' % ci])).sum().backward()
        assert maxabs(xr.grad, t(g['gx%d' % ci])) <= 2e-6
    for name in ('a', 'b'):
        y = O.fused_leaky_relu(t(g['act_x_' + name]), t(g['act_b_' + name]))
        assert maxabs(y, t(g['act_y_' + name])) == 0.0


def test_kat2_modulated_conv_forward_and_grads():
    g = golden('kat2_modconv.npz')
    for name in g['names']:
        cin, cout, h, k, demod, up = [int(v) for v in g['%s.cfg' % name]]
        x = t(g['%s.x' % name]).requires_grad_(True)
        s = t(g['%s.s' % name]).requires_grad_(True)
        W = t(g['%s.w' % name]).requires_grad_(True)
        mw = t(g['%s.mw' % name]).requires_grad_(True)
        mb = t(g['%s.mb' % name]).requires_grad_(True)
        y = O.modulated_conv2d(x, s, W, mw, mb, demodulate=bool(demod), upsample=bool(up))
        assert maxabs(y, t(g['%s.y' % name])) <= 5e-6
        (y * t(g['%s.g' % name])).sum().backward()
        for ours, key in ((x, 'gx'), (s, 'gs'), (W, 'gw'), (mw, 'gmw'), (mb, 'gmb')):
            assert maxabs(ours.grad, t(g['%s.%s' % (name, key)])) <= 3e-5, (name, key)


def test_kat3_small_generators_every_layer():
    g = golden('kat3_small_generators.npz')
    for size in (32, 64):
        P = synthetic_state(size, 1)
        img, _, layers = O.generator_forward(P, [t(g['g%d.w' % size])], input_is_latent=True, return_layers=True)
        assert maxabs(img, t(g['g%d.image' % size])) <= 2e-5
        for nm, f in layers.items():
            idx = torch.from_numpy(g['g%d.%s.probe_idx' % (size, nm)])
            assert maxabs(f.reshape(-1)[idx], t(g['g%d.%s.probe' % (size, nm)])) <= 2e-5, nm


def _digest_ok(img, g, prefix, tol=5e-5):
    d = image_digest(img)
    assert np.abs(d['sub'] - g[prefix + '.sub']).max() <= tol
    assert np.abs(d['rowsum'] - g[prefix + '.rowsum']).max() <= 40 * tol
    assert np.abs(d['colsum'] - g[prefix + '.colsum']).max() <= 40 * tol


def test_kat4_generator256_cm1():
    g = golden('kat4_generator256.npz')
    P = synthetic_state(256, 1)
    assert sum(v.numel() for k, v in P.items() if not k.startswith('noises.') and not k.endswith('.kernel')) \
        == int(g['n_params_cm1']) == 24767458
    torch.set_num_threads(8)
    with torch.no_grad():
        trunc = O.mean_latent_from(P, S.synthetic_z(SEED, 64, key='kat4.ztrunc'))       # KAT-7
        assert maxabs(trunc, t(g['cm1.trunc'])) <= 1e-6
        img, lat = O.generator_forward(P, [S.synthetic_z(SEED, 2, key='kat4.z')], return_latents=True,
                                       truncation=0.7, truncation_latent=trunc)
        assert maxabs(lat, t(g['cm1.lat_z'])) <= 1e-6
        _digest_ok(img, g, 'cm1.z')
        w = S.synthetic_latents(SEED, 2, key='kat4.w')
        img, _ = O.generator_forward(P, [w], input_is_latent=True)
        _digest_ok(img, g, 'cm1.p')
        assert maxabs(img[0], t(g['cm1.p.full0'])) <= 5e-5


def test_kat5_generate_image_shift_and_dA():
    g4, g = golden('kat4_generator256.npz'), golden('kat5_generate_image.npz')
    P = synthetic_state(256, 1)
    trunc = t(g4['cm1.trunc'])
    sv = t(g['sv'])
    A = {k: v.clone().requires_grad_(True) for k, v in S.synthetic_direction_state(SEED).items()}
    shift = O.direction_matrix(A, sv)
    assert maxabs(shift, t(g['shift'])) <= 1e-6
    w = S.synthetic_latents(SEED, 2, key='kat4.w')
    torch.set_num_threads(8)
    img, lat = O.generate_image(P, w, 0.7, trunc, shift_code=shift, input_is_latent=True, return_latents=True)
    assert maxabs(lat, t(g['w.latent'])) <= 1e-6
    _digest_ok(img, g, 'w')
    (img ** 2).mean().backward()
    ref = t(g['w.gA'])
    assert maxabs(A['linear.weight'].grad, ref) <= 1e-4 * float(ref.abs().max())
    A2 = S.synthetic_direction_state(SEED, w_plus=False)
    assert maxabs(O.direction_matrix(A2, sv, w_plus=False), t(g['shift_w'])) <= 1e-6
    assert tuple(g['init_normal.shape']) == (4096, 15) and int(g['init_eye.nnz']) == 8 * 15


def test_state_layout_matches_reference_keys():
    shapes = O.generator_state_shapes(256, 512, 8, 1)
    assert len(shapes) == 135
    assert shapes['convs.10.conv.weight'] == (1, 64, 128, 3, 3)
    assert shapes['to_rgbs.5.conv.weight'] == (1, 3, 64, 1, 1)
    assert shapes['noises.noise_12'] == (1, 1, 256, 256)
    assert O.generator_state_shapes(256, 512, 8, 2)['convs.11.conv.weight'] == (1, 128, 128, 3, 3)


SHIFT_CASES = (('voxceleb', 15, 6), ('ffhq', 12, 6.0), ('voxceleb', 15, 4.5))


def shift_case(g, dataset, D, sc, B=8):
    """(tag, oracle config, source (angles, params), target (angles, params)) of one kat8 case, regenerated from the seed."""
    from oracle import shift_oracle as SO
    tag = '%s_%d_%s' % (dataset, D, str(sc).replace('.', 'p'))
    cfg = SO.initialize_directions(dataset, D, sc, g['ranges_' + dataset])
    return tag, cfg, S.synthetic_shape_params(SEED, tag + '.src', B), S.synthetic_shape_params(SEED, tag + '.tgt', B)


def test_kat8_shift_vectors_bit_exact():
    """oracle/shift_oracle.py == the real reference's make_shift / make_shift_vector(_50) (golden), bit for bit."""
    from oracle import shift_oracle as SO
    g = golden('kat8_shift.npz')
    for dataset, D, sc in SHIFT_CASES:
        tag, cfg, (ang_s, par_s), (ang_t, par_t) = shift_case(g, dataset, D, sc)
        coef = np.array([[cfg['a_jaw'], cfg['b_jaw']]] + [[d['a'], d['b']] for d in cfg['directions_exp']])
        assert (coef == g[tag + '.coef']).all()
        assert (SO.make_shift_vector(cfg, par_s, par_t, ang_s, ang_t).numpy() == g[tag + '.train']).all()
        sv50 = SO.make_shift_vector_50(cfg, par_s, par_t, ang_s, ang_t, g[tag + '.which'], t(g[tag + '.u']))
        assert (sv50.numpy() == g[tag + '.train50']).all()
        if tag + '.infer' in g.files:
            rows = [SO.make_shift(cfg, ang_s[0:1], ang_t[i:i + 1], {k: v[0:1] for k, v in par_s.items()},
                                  {k: v[i:i + 1] for k, v in par_t.items()}) for i in range(ang_t.shape[0])]
            assert (torch.cat(rows, 0).numpy() == g[tag + '.infer']).all()


def test_kat1b_upfirdn2d_per_axis_factors():
    g = golden('kat1b_upfirdn_mixed.npz')
    k = t(g['kernel'])
    for i in range(int(g['n'])):
        cfg = [int(v) for v in g['cfg%d' % i]]
        up, down, pad = tuple(cfg[4:6]), tuple(cfg[6:8]), tuple(cfg[8:12])
        x = t(g['x%d' % i]).requires_grad_(True)
        y = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
        assert y.shape == g['y%d' % i].shape and maxabs(y, t(g['y%d' % i])) <= 2e-6
        (y * t(g['g%d' % i])).sum().backward()
        assert maxabs(x.grad, t(g['gx%d' % i])) <= 4e-6
