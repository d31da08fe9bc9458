"""GPU: shift vectors built on the device (csrc/shift.hip through the C ABI) against the goldens captured from the real
reference's Inference.make_shift / Utilities_train.make_shift_vector(_50) (kat8, oracle/make_golden_shift.py).
Bar: BIT-exact (the kernel mirrors the reference's float64-scalar / float32-tensor arithmetic operation by operation)."""
import numpy as np
import pytest
import torch

from util import S, SEED, golden, t
from test_cpu_oracle import SHIFT_CASES, shift_case

pytestmark = pytest.mark.gpu


def _cuda(ang, par):
    return ang.cuda(), {k: v.cuda() for k, v in par.items()}


def _builder(g, dataset, D, sc):
    from stylegan_directions_face_reenactment_amd.shift import ShiftVectors
    return ShiftVectors(dataset, D, sc, ranges=g['ranges_' + dataset])


def test_direction_tables_match_reference_coefficients():
    g = golden('kat8_shift.npz')
    for dataset, D, sc in SHIFT_CASES:
        tag = shift_case(g, dataset, D, sc)[0]
        sv = _builder(g, dataset, D, sc)
        coef = np.array([[sv.a_jaw, sv.b_jaw]] + [[d['a'], d['b']] for d in sv.directions_exp])
        assert (coef == g[tag + '.coef']).all()
        assert sv.count_pose == (4 if dataset == 'voxceleb' else 3) and sv.num_expressions == D - sv.count_pose


def test_make_shift_batched_equals_reference_per_frame_loop():
    g = golden('kat8_shift.npz')
    for dataset, D, sc in SHIFT_CASES:
        tag, _, src, tgt = shift_case(g, dataset, D, sc)
        if tag + '.infer' not in g.files:
            continue
        sv = _builder(g, dataset, D, sc)
        (ang_s, par_s), (ang_t, par_t) = _cuda(*src), _cuda(*tgt)
        one = {k: v[0:1] for k, v in par_s.items()}
        out = sv.make_shift(ang_s[0:1], ang_t, one, par_t)                 # one source identity, 8 targets, ONE launch
        assert out.shape == (8, D) and out.is_cuda
        assert (out.cpu().numpy() == g[tag + '.infer']).all()
        rep = sv.make_shift(ang_s[0:1].expand(8, -1).contiguous(), ang_t, {k: v.expand(8, -1).contiguous() for k, v in one.items()}, par_t)
        assert torch.equal(rep, out)                                        # per-frame sources = the broadcast


def test_make_shift_vector_and_50_bit_exact():
    g = golden('kat8_shift.npz')
    for dataset, D, sc in SHIFT_CASES:
        tag, _, src, tgt = shift_case(g, dataset, D, sc)
        sv = _builder(g, dataset, D, sc)
        (ang_s, par_s), (ang_t, par_t) = _cuda(*src), _cuda(*tgt)
        out = sv.make_shift_vector(par_s, par_t, ang_s, ang_t)
        assert (out.cpu().numpy() == g[tag + '.train']).all()
        out50, which = sv.make_shift_vector_50(par_s, par_t, ang_s, ang_t, target_indices=g[tag + '.which'], u=t(g[tag + '.u']).cuda())
        assert (out50.cpu().numpy() == g[tag + '.train50']).all()
        assert (which.cpu().numpy() == g[tag + '.which']).all()


def test_make_shift_vector_50_device_draws():
    g = golden('kat8_shift.npz')
    dataset, D, sc = SHIFT_CASES[0]
    sv = _builder(g, dataset, D, sc)
    B = 64
    ang_s, par_s = _cuda(*S.synthetic_shape_params(SEED, 'draw.src', B))
    ang_t, par_t = _cuda(*S.synthetic_shape_params(SEED, 'draw.tgt', B))
    out, which = sv.make_shift_vector_50(par_s, par_t, ang_s, ang_t)
    full = sv.make_shift_vector(par_s, par_t, ang_s, ang_t)
    assert torch.equal(out[:B // 2], full[:B // 2])                         # first half = full reenactment shift
    second, which = out[B // 2:].cpu(), which.cpu().long()
    assert which.min() >= 0 and which.max() < D
    onehot = torch.zeros(B // 2, D, dtype=torch.bool)
    onehot[torch.arange(B // 2), which] = True
    assert (second[~onehot] == 0).all()                                     # one direction per sample
    # the draw spans [-shift_scale - start, shift_scale - start]: u = 0 gives the upper end, u = 1 the lower, 2*shift_scale apart
    hi, _ = sv.make_shift_vector_50(par_s, par_t, ang_s, ang_t, target_indices=which, u=torch.zeros(B // 2).cuda())
    lo, _ = sv.make_shift_vector_50(par_s, par_t, ang_s, ang_t, target_indices=which, u=torch.ones(B // 2).cuda())
    span = (hi[B // 2:].cpu() - lo[B // 2:].cpu())[onehot]
    assert (span - 2 * sc).abs().max() <= 1e-5
    moved = second[onehot]
    assert ((moved <= hi[B // 2:].cpu()[onehot]) & (moved >= lo[B // 2:].cpu()[onehot])).all()
    with pytest.raises(RuntimeError):
        sv.make_shift_vector_50({k: v[:3] for k, v in par_s.items()}, {k: v[:3] for k, v in par_t.items()}, ang_s[:3], ang_t[:3])


def test_errors_and_cpu_tensors_refused():
    g = golden('kat8_shift.npz')
    sv = _builder(g, 'voxceleb', 15, 6)
    ang_s, par_s = S.synthetic_shape_params(SEED, 'e.src', 4)
    ang_t, par_t = S.synthetic_shape_params(SEED, 'e.tgt', 4)
    with pytest.raises(RuntimeError):                                       # no CPU path
        sv.make_shift(ang_s, ang_t, par_s, par_t)
    (ang_s, par_s), (ang_t, par_t) = _cuda(ang_s, par_s), _cuda(ang_t, par_t)
    with pytest.raises(RuntimeError):                                       # 3 sources for 4 targets
        sv.make_shift(ang_s[:3], ang_t, {k: v[:3] for k, v in par_s.items()}, par_t)
    with pytest.raises(RuntimeError):                                       # alpha_exp too short for the table
        sv.make_shift(ang_s, ang_t, {'pose': par_s['pose'], 'alpha_exp': par_s['alpha_exp'][:, :5].contiguous()},
                      {'pose': par_t['pose'], 'alpha_exp': par_t['alpha_exp'][:, :5].contiguous()})
    assert sv.make_shift(ang_s[:0], ang_t[:0], {k: v[:0] for k, v in par_s.items()}, {k: v[:0] for k, v in par_t.items()}).shape == (0, 15)


def test_session_renders_from_shape_parameters():
    """ReenactmentSession.render_targets(params) == render(make_shift(params)) == the golden shift vectors rendered."""
    from util import hip_generator
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession
    g = golden('kat8_shift.npz')
    dataset, D, sc = SHIFT_CASES[0]
    tag, _, src, tgt = shift_case(g, dataset, D, sc)
    G = hip_generator(64, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(5))
    A = A.cuda()
    w = S.synthetic_latents(5, 1, n_latent=G.n_latent).cuda()
    trunc = S.counter_tensor(5, 'trunc', (1, 512)).cuda()
    sess = ReenactmentSession(G, A, w, 0.7, trunc, batch=3, shifts=_builder(g, dataset, D, sc))
    (ang_s, par_s), (ang_t, par_t) = _cuda(*src), _cuda(*tgt)
    one = {k: v[0:1] for k, v in par_s.items()}
    imgs = sess.render_targets(ang_s[0:1], one, ang_t, par_t)
    ref = sess.render(t(g[tag + '.infer']).cuda())
    assert imgs.shape == (8, 3, 64, 64) and torch.equal(imgs, ref)
