"""the e4e W+ producer in front of the generator path.

CPU: the oracle restatement (oracle/e4e_oracle.py) against the fixture produced by the real reference
(tests/golden/kat6_e4e.npz, oracle/make_golden_e4e.py) and the state_dict key contract.  GPU: the package's encoder
(MIOpen convs + HIP EqualLinear, folded inference plan and autograd path) against the same fixture, and the
encoder -> DirectionMatrix -> generator flow of run_inference.py:95-117,170-181.
"""
import zlib

import pytest
import torch

from util import O, S, SEED, golden, maxabs, t
from oracle import e4e_oracle as E

gpu = pytest.mark.gpu


def _encoder(res, seed):
    from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
    enc = Encoder4Editing(50, 'ir_se', res).eval()
    P = S.synthetic_encoder_state(enc.state_dict(), seed=seed)
    enc.load_state_dict(P, strict=True)
    return enc, P


def _inputs():
    x256 = S.counter_tensor(SEED, 'e4e.x', (2, 3, 256, 256), 0.0, 0.5).clamp_(-1, 1)
    x64 = S.counter_tensor(SEED, 'e4e.x64', (3, 3, 64, 64), 0.0, 0.5).clamp_(-1, 1)
    return x256, x64


def test_encoder_state_dict_contract_and_oracle_vs_reference_fixture():
    g = golden('kat6_e4e.npz')
    enc, P = _encoder(64, SEED + 1)
    keys = list(enc.state_dict().keys())
    from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
    keys256 = list(Encoder4Editing(50, 'ir_se', 256).state_dict().keys())
    assert len(keys256) == int(g['n_keys']) and zlib.crc32('\n'.join(keys256).encode()) == int(g['key_crc'][0])
    assert set(keys) < set(keys256)                        # res 64: 10 style heads instead of 14, same trunk
    _, x64 = _inputs()
    with torch.no_grad():
        w = E.encoder_forward(P, x64)
    assert w.shape == (3, 10, 512) and maxabs(w, t(g['w64'])) <= 2e-5 * float(abs(g['w64']).max())
    with pytest.raises(ValueError):
        Encoder4Editing(34, 'ir_se', 256)
    with pytest.raises(ValueError):
        Encoder4Editing(50, 'se', 256)


def test_encoder_has_no_cpu_fallback():
    enc, _ = _encoder(64, SEED + 1)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            enc(torch.zeros(1, 3, 64, 64))


@gpu
def test_encoder_matches_reference_fixture_on_gpu():
    g = golden('kat6_e4e.npz')
    x256, x64 = _inputs()
    for res, x, key, seed in ((64, x64, 'w64', SEED + 1), (256, x256, 'w256', SEED)):
        enc, P = _encoder(res, seed)
        enc = enc.cuda()
        ref = t(g[key])
        tol = 1e-3 * float(ref.abs().max())                # fp32 convs on MIOpen (algorithm-dependent summation order)
        with torch.no_grad():
            planned = enc(x.cuda())
        assert planned.shape == ref.shape and maxabs(planned, ref) <= tol
        trainable = enc(x.cuda())                          # grad mode: module-by-module path
        assert trainable.requires_grad and maxabs(trainable, ref) <= tol
        with torch.no_grad():                              # plan is rebuilt when a parameter changes
            enc.styles[0].linear.bias.add_(1.0)
            moved = enc(x.cuda())
        assert abs(maxabs(moved, planned) - 1.0) <= 1e-3


@gpu
def test_encode_then_reenact_flow():
    """run_inference.py:95-117 (source -> e4e -> W+) feeding :170-181 (shift + generator), batched."""
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession
    from util import hip_generator
    enc, P = _encoder(64, SEED + 1)
    enc = enc.cuda()
    G = hip_generator(64, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(SEED))
    A = A.cuda()
    _, x64 = _inputs()
    with torch.no_grad():
        src = enc(x64[:1].cuda())
    assert src.shape == (1, G.n_latent, 512)
    trunc = S.counter_tensor(7, 'flow.t', (1, 512)).cuda()
    sv = S.counter_tensor(7, 'flow.sv', (5, 15), 0.0, 2.0).cuda()
    out = ReenactmentSession(G, A, src, 0.7, trunc, batch=4).render(sv)
    # oracle: same flow on CPU
    PG = {k: v.cpu() for k, v in G.state_dict().items()}
    PA = {k: v.cpu() for k, v in A.state_dict().items()}
    with torch.no_grad():
        w_cpu = E.encoder_forward(P, x64[:1])
        shift = O.direction_matrix(PA, sv.cpu())
        ref = torch.cat([O.generate_image(PG, src.cpu(), 0.7, trunc.cpu(), shift_code=shift[i:i + 1],
                                          input_is_latent=True) for i in range(5)], 0)   # the per-frame loop
    assert maxabs(src, w_cpu) <= 1e-3 * float(w_cpu.abs().max())
    assert maxabs(out, ref) <= 1e-3


@gpu
def test_config3_joint_flow_at_256_batch32():
    """BASELINE configs[2] as one flow at its real size: e4e(256) source code -> shift vectors from 3DMM parameters (device)
    -> DirectionMatrix -> shift + truncation 0.7 -> Generator(256) at B=32 -> frames; three of the 32 frames are checked
    against the CPU oracle running the reference's per-frame loop (run_inference.py:170-181) on the oracle's own e4e code
    and the oracle's own shift vectors."""
    from oracle import shift_oracle as SO
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession
    from stylegan_directions_face_reenactment_amd.shift import ShiftVectors
    from util import golden, hip_generator, synthetic_state
    enc, P = _encoder(256, SEED)
    enc = enc.cuda()
    G = hip_generator(256, 1)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    PA = S.synthetic_direction_state(SEED)
    A.load_state_dict(PA)
    A = A.cuda()
    x256, _ = _inputs()
    ranges = golden('kat8_shift.npz')['ranges_voxceleb']
    ang_s, par_s = S.synthetic_shape_params(SEED, 'c3j.src', 1)
    ang_t, par_t = S.synthetic_shape_params(SEED, 'c3j.tgt', 32)
    trunc = S.counter_tensor(7, 'c3j.t', (1, 512))
    cuda = lambda d: {k: v.cuda() for k, v in d.items()}
    with torch.no_grad():
        src = enc(x256[:1].cuda())
        sess = ReenactmentSession(G, A, src, 0.7, trunc.cuda(), batch=32, shifts=ShiftVectors('voxceleb', 15, 6.0, ranges=ranges))
        out = sess.render_targets(ang_s.cuda(), cuda(par_s), ang_t.cuda(), cuda(par_t))
        assert out.shape == (32, 3, 256, 256)
        cfg = SO.initialize_directions('voxceleb', 15, 6.0, ranges)
        w_cpu = E.encoder_forward(P, x256[:1])
        PG = synthetic_state(256, 1)
        for i in (0, 13, 31):
            sv = SO.make_shift(cfg, ang_s, ang_t[i:i + 1], par_s, {k: v[i:i + 1] for k, v in par_t.items()})
            ref = O.generate_image(PG, w_cpu, 0.7, trunc, shift_code=O.direction_matrix(PA, sv), input_is_latent=True)
            assert maxabs(out[i:i + 1], ref) <= 1e-3, i      # north-star contract, end to end incl. the encoder
