"""CPU: host logic of the package -- module/state_dict contract, synthetic generator determinism, the C-ABI
library loads and exports every declared symbol, and the product path refuses CPU tensors (no fallback)."""
import copy
import ctypes
import os
import pickle
import re

import pytest
import torch

from util import O, ROOT, S, SEED, synthetic_state


def test_counter_rng_is_pinned_and_normal():
    v = S.counter_normal(1, 'abc', 200000)
    assert abs(v.mean()) < 0.01 and abs(v.std() - 1.0) < 0.01 and abs(v).max() < 3.5
    # fixed values: any change of the generator silently invalidates every golden fixture
    w = S.counter_tensor(SEED, 'conv1.conv.weight', (4,))
    assert [round(float(x), 6) for x in w] == [round(float(x), 6) for x in S.counter_tensor(SEED, 'conv1.conv.weight', (4,))]
    a = S.counter_normal(SEED, 'kat1.x', 3)
    b = S.counter_normal(SEED, 'kat1.x', 5)
    assert (a == b[:3]).all()                      # counter-based: prefix stable
    assert (S.counter_normal(SEED, 'kat1.y', 3) != a).all()


def test_synthetic_state_matches_golden_inputs():
    import numpy as np
    from util import golden
    g = golden('kat1_ops.npz')
    assert np.array_equal(S.counter_tensor(SEED, 'kat1.x', (2, 3, 9, 9)).numpy(), g['x'])   # same generator as make_golden


def test_generator_state_dict_contract():
    from stylegan_directions_face_reenactment_amd.model import Generator, EqualLinear
    for cm, nparams in ((1, 24767458), (2, 30034338)):
        G = Generator(256, 512, 8, channel_multiplier=cm)
        sd = G.state_dict()
        shapes = O.generator_state_shapes(256, 512, 8, cm)
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
        assert sum(p.numel() for p in G.parameters()) == nparams
    G = Generator(256, 512, 8, channel_multiplier=1)
    assert (G.n_latent, G.num_layers, G.log_size, G.size, G.style_dim) == (14, 13, 8, 256, 512)
    assert len(G.convs) == 12 and len(G.to_rgbs) == 6 and G.channels[256] == 64
    assert tuple(G.input.input.shape) == (1, 512, 4, 4)
    assert [tuple(getattr(G.noises, 'noise_%d' % i).shape)[-1] for i in range(13)] == \
        [4, 8, 8, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256]
    # strict load of a synthetic state, non-strict load of a checkpoint without buffers (run_inference.py:66-67)
    G.load_state_dict(synthetic_state(256, 1), strict=True)
    partial = {k: v for k, v in synthetic_state(256, 1).items() if 'noises.' not in k and '.kernel' not in k}
    missing = G.load_state_dict(partial, strict=False)
    assert all(('noises.' in k) or k.endswith('.kernel') for k in missing.missing_keys)
    # deepcopy / pickle (libs/optimization.py:28 deep-copies G) and optimizer parameter groups (:32-35)
    G2 = copy.deepcopy(G)
    assert torch.equal(G2.conv1.conv.weight, G.conv1.conv.weight)
    pickle.loads(pickle.dumps(G.convs[0]))
    assert len([p for i in range(4, 12) for p in G.convs[i].parameters()]) == 8 * 5
    assert isinstance(G.style[1], EqualLinear) and G.style[1].lr_mul == 0.01
    assert torch.allclose(G.convs[0].conv.blur.kernel.sum(), torch.tensor(4.0))
    assert G.convs[0].conv.blur.pad == (1, 1) and G.to_rgbs[0].upsample.pad == (2, 1)


def test_direction_matrix_contract():
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    assert tuple(A.linear.weight.shape) == (4096, 15) and A.input_dim == 15
    assert list(A.state_dict().keys()) == ['linear.weight', 'linear.bias']
    assert 0.02 < float(A.linear.weight.std()) < 0.04
    E = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, initialization='eye', verbose=False)
    assert int((E.linear.weight != 0).sum()) == 8 * 15
    assert torch.equal(E.linear.weight[512 * 3:512 * 3 + 15, :15], torch.eye(15))
    W = DirectionMatrix(512, verbose=False)                  # np.product-free default dims
    assert (W.input_dim, W.out_dim) == (512, 512)
    pickle.loads(pickle.dumps(A))


def test_library_loads_and_exports_every_declared_symbol():
    from stylegan_directions_face_reenactment_amd import _native
    header = open(os.path.join(ROOT, 'include', 'sgdfr.h')).read()
    declared = set(re.findall(r'\b(sgdfr_[a-z0-9_]+)\s*\(', header))
    assert declared >= set(_native.SIGNATURES) | {'sgdfr_abi_version', 'sgdfr_last_error'}
    lib = _native.load()                       # raises if the .so was not built
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sgdfr_abi_version() == _native.ABI_VERSION
    # argument validation happens before any launch, so error paths are testable without a GPU
    rc = lib.sgdfr_upfirdn2d_f32(None, None, None, 1, 4, 4, 1, 99, 4, 1, 1, 1, 1, 0, 0, 0, 0, None)
    assert rc != 0 and b'kernel' in lib.sgdfr_last_error()
    rc = lib.sgdfr_fused_bias_act_f32(None, None, None, None, 0, 1, 1, 1, 0, 0.2, 1.0, None)
    assert rc != 0 and b'act=3' in lib.sgdfr_last_error()
    rc = lib.sgdfr_modconv2d_fwd_f32(None, 0, None, None, None, None, 0, None, None, None, 1, 8, 8, 4, 4, 7, 0, 0.2,
                                     1.0, None)
    assert rc != 0 and b'mode' in lib.sgdfr_last_error()


def test_product_path_has_no_cpu_fallback():
    from stylegan_directions_face_reenactment_amd.model import Generator
    from stylegan_directions_face_reenactment_amd.op import fused_leaky_relu, upfirdn2d
    from stylegan_directions_face_reenactment_amd import functional as F_
    G = Generator(32, 512, 8, channel_multiplier=1)
    with pytest.raises(RuntimeError, match='no CPU path'):
        G([torch.randn(1, 512)])
    with pytest.raises(RuntimeError, match='no CPU path'):
        fused_leaky_relu(torch.randn(2, 3), torch.zeros(3))
    with pytest.raises(RuntimeError, match='no CPU path'):
        upfirdn2d(torch.randn(1, 1, 4, 4), torch.ones(4, 4))
    with pytest.raises(RuntimeError, match='no CPU path'):
        F_.linear(torch.randn(2, 4), torch.randn(3, 4))
    # and nothing in the shipped package imports the oracle
    pkg = os.path.join(ROOT, 'stylegan_directions_face_reenactment_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('no oracle import', ''), os.path.join(dirpath, f)


def test_compat_aliases():
    import sys
    from stylegan_directions_face_reenactment_amd import compat
    saved = {k: sys.modules.get(k) for k in compat.ALIASES}
    try:
        compat.install()
        from libs.gan.StyleGAN2.model import Generator, EqualLinear          # noqa: F401
        from libs.models.direction_matrix import DirectionMatrix             # noqa: F401
        from libs.gan.StyleGAN2.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d   # noqa: F401
        import stylegan_directions_face_reenactment_amd.model as M
        assert Generator is M.Generator
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_shard_range_and_flops_table():
    from stylegan_directions_face_reenactment_amd import distributed as D, functional as F_
    for total, world in ((512, 8), (128, 8), (10, 4), (3, 8)):
        spans = [D.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    # 29.75 GFLOP/img of 3x3 conv at cm=1
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64}
    fl = F_.conv_flops(512, 512, 4, 4)
    cin = 512
    for r in (8, 16, 32, 64, 128, 256):
        fl += F_.conv_flops(cin, ch[r], r // 2, r // 2, upsample=True) + F_.conv_flops(ch[r], ch[r], r, r)
        cin = ch[r]
    assert abs(fl / 1e9 - 29.746) < 0.01


def test_latent_store_roundtrip(tmp_path):
    """W+ latent store = one [14,512] fp32 .npy per frame (invert_images.py:119-125)."""
    import numpy as np
    from stylegan_directions_face_reenactment_amd.reenact import save_latent_codes, load_latent_codes
    lat = S.synthetic_latents(3, 5)
    names = ['%06d.png' % i for i in range(5)]
    paths = save_latent_codes(str(tmp_path / 'latent_codes'), names, lat)
    assert [p.split('/')[-1] for p in paths] == ['%06d.npy' % i for i in range(5)]
    one = np.load(paths[2])
    assert one.shape == (14, 512) and one.dtype == np.float32 and np.array_equal(one, lat[2].numpy())
    back = load_latent_codes(paths)
    assert back.shape == (5, 14, 512) and torch.equal(back, lat)
    with pytest.raises(RuntimeError):
        save_latent_codes(str(tmp_path / 'x'), names[:2], lat)
    np.save(str(tmp_path / 'bad.npy'), np.zeros((3, 512), np.float32))
    with pytest.raises(RuntimeError):
        load_latent_codes([paths[0], str(tmp_path / 'bad.npy')])


def test_direction_tables_on_cpu_match_reference_coefficients():
    """shift.initialize_directions (host logic, no GPU) == the reference's generic.initialize_directions as captured in kat8:
    same (a, b) lines bit for bit, same direction layout for both datasets."""
    import numpy as np
    from stylegan_directions_face_reenactment_amd import shift as SH
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'kat8_shift.npz'))
    for dataset, D, sc in (('voxceleb', 15, 6), ('ffhq', 12, 6.0), ('voxceleb', 15, 4.5)):
        tag = '%s_%d_%s' % (dataset, D, str(sc).replace('.', 'p'))
        count_pose, n_exp, dirs, jaw, scales, angle_dirs = SH.initialize_directions(dataset, D, sc, g['ranges_' + dataset])
        coef = np.array([[jaw['a'], jaw['b']]] + [[d['a'], d['b']] for d in dirs])
        assert (coef == g[tag + '.coef']).all()
        assert count_pose == (4 if dataset == 'voxceleb' else 3) and n_exp == D - count_pose
        assert [d['A_direction'] for d in dirs] == list(range(count_pose, D)) and list(scales) == [40.0, 20.0, 20.0]
    with pytest.raises(FileNotFoundError):
        SH.get_direction_ranges('/nonexistent/ranges.npy')
    with pytest.raises(RuntimeError):
        SH.ShiftVectors('voxceleb', 15, 6, ranges=g['ranges_voxceleb']).make_shift(
            torch.zeros(1, 3), torch.zeros(2, 3), {'pose': torch.zeros(1, 6), 'alpha_exp': torch.zeros(1, 50)},
            {'pose': torch.zeros(2, 6), 'alpha_exp': torch.zeros(2, 50)})      # CPU tensors: refused, no fallback


def test_winograd_layer_plan_of_the_bench_configs():
    """Which plain layers the no-grad chain runs in 1-D Winograd form is a pure function of (batch, layer shapes, switches) answered
    by host-only shape queries of the library: pinned here for the BASELINE configs (no GPU needed)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.model import Generator
    G = Generator(256, 512, 8, channel_multiplier=1)
    layers = [G.conv1] + list(G.convs)
    assert F_.USE_WSPLIT and F_.WSPLIT_F == 4 and F_.WSPLIT_MIN_CIN == 128
    # B=64 (configs[1]): 512 @ 16^2, 512 @ 32^2, 256 @ 64^2, 128 @ 128^2 in F(4,3); the 64 -> 64 @ 256^2 layer stays direct
    assert G._wino_inputs(64, layers) == {4: 4, 6: 4, 8: 4, 10: 4}
    # B=32 (configs[2]): the 16^2 layer is K-sliced (too few tiles to fill the chip), so it cannot fuse its ToRGB and stays direct
    assert G._wino_inputs(32, layers) == {6: 4, 8: 4, 10: 4}
    assert G._wino_inputs(2, layers) == {}
    with F_.using(F_.config().replace(wsplit_f=2)):
        assert G._wino_inputs(64, layers) == {4: 2, 6: 2, 8: 2}         # (F(2,3) pays from 256 input channels on)
    lib = F_.N.load()
    assert lib.sgdfr_modconv_prepack_wsplit_elems(512, 512, 4) == 512 * 512 * 3 * 6 * 2 + 8       # (+ the 16-byte trailer: max |w| of fp16f8 packs)
    assert lib.sgdfr_modconv_prepack_wsplit_elems(512, 512, 2) == 512 * 512 * 3 * 4 * 2 + 8
    # fp8 cross terms (Config.cross_terms='fp8', SGDFR_SPLIT_FP16F8): only for the layers whose F(4,3) launch takes the wide-tile
    # kernel by its tile count -- the three big ones at B=64, not the 16^2 layer (128 wide tiles < 192), nothing at small batches
    noise = [object()] * len(layers)
    assert [p[5] for p in G._chain_plan(64, True, noise, layers)] == [None] * len(layers)                 # default: three fp16 products
    with F_.using(F_.config().replace(cross_terms='fp8')):
        assert [p[5] for p in G._chain_plan(64, True, noise, layers)] == [None] * 5 + ['fp16f8', None, 'fp16f8', None, 'fp16f8', None, 'fp16f8', None]      # (the last one: plain form, for the direct 64 -> 64 @ 256^2 layer)
        # ... and those layers hand over to the transposed convs after them (deep plan) in the same form
        assert [p[6] for p in G._chain_plan(64, True, noise, layers)] == [None] * 4 + ['fp16f8', None, 'fp16f8', None, 'fp16f8', None, 'fp16f8', None, None]
        assert [p[5] for p in G._chain_plan(4, True, noise, layers)] == [None] * len(layers)
        # (B=4: only the 128 @ 128^2 layer is in F(4,3) form -- on the 64-tile kernel -- and the last transposed conv just reaches its deep plan)
        assert [p[6] for p in G._chain_plan(4, True, noise, layers)] == [None] * 10 + ['fp16f8', None, None]
        assert [p[6] for p in G._chain_plan(2, True, noise, layers)] == [None] * len(layers)
        with F_.precision('bf16x3'):                                                                      # the saturation fallback keeps its own arithmetic
            assert [p[5] for p in G._chain_plan(64, True, noise, layers)] == [None] * len(layers)
    assert lib.sgdfr_modconv2d_wsplit_wide(64, 256, 256, 64, 64) == 1 and lib.sgdfr_modconv2d_wsplit_wide(64, 512, 512, 16, 16) == 0
    assert lib.sgdfr_modconv2d_wsplit_wide(64, 80, 128, 64, 64) == 0                                      # Cin % 32: channel blocks pair up
    assert lib.sgdfr_modconv2d_split_f8_ok(64, 256, 128, 64, 64, F_.N.MODE_UP3) == 1 and lib.sgdfr_modconv2d_split_f8_ok(64, 512, 512, 4, 4, F_.N.MODE_UP3) == 0
    assert not lib.sgdfr_modconv2d_wsplit_supported(64, 64, 64, 256, 256, 4)          # Cout = 64: no 128-cout tile


def test_config_object_is_frozen_scoped_and_seeded_from_the_environment():
    """functional.Config: frozen and hashable (launch plans / hipGraph keys hash it), `using` blocks nest and restore, the module-level
    names are read-only views, environment variables only seed the default."""
    import dataclasses
    from stylegan_directions_face_reenactment_amd import functional as F_
    base = F_.config()
    assert base is F_.DEFAULT and hash(base) == hash(base.replace())
    with pytest.raises(dataclasses.FrozenInstanceError):
        base.precision = 'fp32'
    with pytest.raises(ValueError):
        base.replace(precision='fp8')
    with F_.using(base.replace(precision='fp32', use_wsplit=False)) as c1:
        assert F_.config() is c1 and F_.PRECISION == 'fp32' and F_.USE_WSPLIT is False
        with F_.precision('bf16x3'):
            assert F_.config().precision == 'bf16x3' and F_.config().use_wsplit is False
        assert F_.config() is c1
    assert F_.config() is base and F_.PRECISION == base.precision
    env = F_.Config.from_env({'SGDFR_PRECISION': 'bf16x3', 'SGDFR_RANGE_PLAN': 'exact', 'SGDFR_WSPLIT_F': '2', 'SGDFR_WSPLIT': '0'})
    assert (env.precision, env.range_plan, env.wsplit_f, env.use_wsplit) == ('bf16x3', 'exact', 2, False)
    # the old module-level switch names are read-only views: an assignment must not silently shadow them
    for name in ('PRECISION', 'USE_WSPLIT', 'WSPLIT_F', 'DEFAULT'):
        with pytest.raises(AttributeError):
            setattr(F_, name, None)
    assert F_.USE_WSPLIT is base.use_wsplit and F_.DEFAULT is base
    # launch timing is a per-thread context object, not a module global
    from stylegan_directions_face_reenactment_amd import timing
    assert timing.active() is None and not hasattr(F_, 'CONV_TIMING')
    with timing.collect() as t:
        assert timing.active() is t and t.conv == [] and t.hbm == []
        seen = []
        import threading
        th = threading.Thread(target=lambda: seen.append(timing.active()))
        th.start(); th.join()
        assert seen == [None]
    assert timing.active() is None
    from stylegan_directions_face_reenactment_amd.model import Generator
    G = Generator(32, 512, 8, channel_multiplier=1)
    assert G.config is None and G.range_mode() == base.precision
    G.config = base.replace(precision='fp32')
    assert G.range_mode() == 'fp32'
    import copy
    assert copy.deepcopy(G).config == G.config


def test_bench_compact_line_is_small_and_complete():
    """the line the driver parses is the compact record (<= bench.LINE_LIMIT bytes whatever the legs carry), with
    every contract key, the roofline fields and one scalar triple per secondary leg; built here from a committed full record."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r06_i_bench_detail.json')))
    args = bench.parse_args([])
    line = bench._sig(bench.compact_line(full, args))
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT and len(text) < 0.25 * len(json.dumps(full))
    assert set(bench.CONTRACT_KEYS) <= set(line)
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'dominant_kernel', 'end_to_end')) <= set(line['roofline'])
    assert line['roofline']['dominant_kernel']['name'] == 'split mode1/deep' and 0.2 < line['roofline']['dominant_kernel']['frac'] < 1
    assert set(('value', 'unit', 'cores', 'kind', 'sample', 'host_cores')) <= set(line['cpu_baseline'])
    for leg in ('single_stream', 'default_call', 'alt_arithmetic', 'synthesis_cm2', 'synthesis_fp8_cross_terms', 'inference', 'trainer', 'pti',
                'range_plan_stress'):
        assert 'value' in line['legs'][leg] or 'frames_per_s' in line['legs'][leg], leg
    # the fp8 leg is priced against its own peak (one fp16 + one fp8 MFMA per product), never against the three-product one
    assert line['legs']['synthesis_fp8_cross_terms']['peak'] > 1500 and line['legs']['synthesis_fp8_cross_terms']['frac'] < 0.5
    # nothing table-shaped survives in the compact line
    def depth_lists(o):
        return any(isinstance(v, list) and len(v) > 3 for v in (o.values() if isinstance(o, dict) else [])) or \
            any(depth_lists(v) for v in (o.values() if isinstance(o, dict) else []) if isinstance(v, dict))
    assert not depth_lists(line)


def test_bench_kernel_families_name_the_kernel_that_runs_a_row():
    """bench._family: conv rows group by kernel -- the deep-plan transposed conv apart from the K-sliced small layers, the F(4,3)
    rows by the tiling the library's own query picks (wswide_kernel vs the 64-tile wsplit_kernel), adjoint rows by their kind."""
    import bench
    assert bench._family('split mode1/deep 512->256 @32x32', 64) == 'split mode1/deep'
    assert bench._family('split mode1 512->512 @4x4 K/4', 64) == 'split mode1'
    assert bench._family('wsplit F(4,3) 512->512 @32x32', 64) == 'wsplit F(4,3)/wide'
    assert bench._family('wsplit F(4,3) 512->512 @16x16', 64) == 'wsplit F(4,3)'          # a quarter of a wide tile per image: 64-tile kernel
    assert bench._family('wsplit F(4,3) 512->512 @32x32 [f8 cross]', 64) == 'wsplit F(4,3)/wide [f8 cross]'
    assert bench._family('bwd split down3 256->512 @32x32', 16) == 'bwd split down3'
    assert bench._family('bwd split3 256->256 @64x64', 16) == 'bwd split3'
    assert bench.row_peak('wsplit F(4,3) 512->512 @32x32 [f8 cross]', bench.SPLIT_PEAK_TFLOPS) == bench.F8_CROSS_PEAK_TFLOPS
    assert abs(bench.F8_CROSS_PEAK_TFLOPS - 1666.7) < 0.1
