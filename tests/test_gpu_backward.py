"""GPU parity of the backward pass: HIP gradients vs torch autograd of the CPU oracle (fp64) and
vs the dL/dA golden captured from the real reference through generate_image (KAT-5)."""
import pytest
import torch

from util import O, S, SEED, golden, hip_generator, maxabs, synthetic_state, t

@pytest.fixture(autouse=True, params=['fp16x3', 'fp32'])
def _both_arithmetics(request):
    """Gradients are checked with the split-fp16 conv kernels (forward + plain-conv dL/dx) and with the fp32 MFMA kernels."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    with F_.precision(request.param):
        yield


pytestmark = pytest.mark.gpu


def _rel(a, b):
    return maxabs(a, b) / max(float(b.detach().abs().max()), 1e-12)


@pytest.mark.parametrize('cin,cout,h,up', [(16, 8, 5, False), (8, 16, 4, True), (64, 128, 8, False), (128, 64, 8, True),
                                            (256, 128, 16, True), (128, 128, 16, False), (6, 10, 7, False),
                                            (10, 6, 5, True), (32, 48, 19, True), (64, 64, 40, False)])
def test_styled_conv_gradients(cin, cout, h, up):
    _check_styled_conv_gradients(cin, cout, h, up, 3)


@pytest.mark.parametrize('cin,cout,h,up,B', [(64, 32, 16, True, 1), (32, 16, 8, True, 1), (48, 32, 16, True, 3)])
def test_styled_conv_gradients_with_a_partial_tail_wave(cin, cout, h, up, B):
    """blur_adjoint_kernel's grid-stride loop ends in a wave whose upper lanes have left the loop when
    planes * strips-per-plane is not a multiple of 64 (B=1, Cout=32, 16x16: 2720 strips, 32 lanes in the tail wave, all in
    one plane).  wave_sum (v_readlane ignores EXEC) then added stale registers into the demodulation gradient; the style
    gradient of exactly these shapes is checked here."""
    _check_styled_conv_gradients(cin, cout, h, up, B)


def _check_styled_conv_gradients(cin, cout, h, up, B):
    from stylegan_directions_face_reenactment_amd.model import StyledConv
    key = 'bw.%d.%d.%d.%d' % (cin, cout, h, up)
    m = StyledConv(cin, cout, 3, 64, upsample=up)
    sd = {k: S.counter_tensor(21, key + k, tuple(v.shape)) for k, v in m.state_dict().items() if 'kernel' not in k}
    sd['conv.modulation.bias'] = sd['conv.modulation.bias'] * 0.1 + 1.0
    sd['noise.weight'] = sd['noise.weight'] * 0.1 + 0.1
    if up:
        sd['conv.blur.kernel'] = m.conv.blur.kernel
    m.load_state_dict(sd)
    x = S.counter_tensor(21, key + 'x', (B, cin, h, h))
    st = S.counter_tensor(21, key + 's', (B, 64))
    r = 2 * h if up else h
    nz = S.counter_tensor(21, key + 'n', (1, 1, r, r))
    g = S.counter_tensor(21, key + 'g', (B, cout, r, r))
    # fp64 oracle + torch autograd
    P = {'L.' + k: v.double().requires_grad_(k in ('conv.modulation.weight', 'conv.modulation.bias', 'noise.weight',
                                                    'activate.bias')) for k, v in sd.items()}
    xr, sr = x.double().requires_grad_(True), st.double().requires_grad_(True)
    (O.styled_conv(P, 'L', xr, sr, nz.double(), upsample=up) * g.double()).sum().backward()
    # HIP
    m = m.cuda()
    xh, sh = x.cuda().requires_grad_(True), st.cuda().requires_grad_(True)
    out = m(xh, sh, noise=nz.cuda())
    (out * g.cuda()).sum().backward()
    assert _rel(xh.grad, xr.grad) <= 2e-5, 'dx'
    assert _rel(sh.grad, sr.grad) <= 5e-5, 'dstyle'
    assert _rel(m.conv.modulation.weight.grad, P['L.conv.modulation.weight'].grad) <= 5e-5
    assert _rel(m.conv.modulation.bias.grad, P['L.conv.modulation.bias'].grad) <= 5e-5
    assert _rel(m.noise.weight.grad, P['L.noise.weight'].grad) <= 5e-5
    assert _rel(m.activate.bias.grad, P['L.activate.bias'].grad) <= 5e-5


def test_torgb_gradients():
    from stylegan_directions_face_reenactment_amd.model import ToRGB
    for cin, h in ((64, 8), (512, 4), (20, 6), (128, 48)):
        key = 'bwrgb.%d.%d' % (cin, h)
        m = ToRGB(cin, 64, upsample=True)
        sd = {k: S.counter_tensor(22, key + k, tuple(v.shape)) for k, v in m.state_dict().items() if 'kernel' not in k}
        sd['upsample.kernel'] = m.upsample.kernel
        m.load_state_dict(sd)
        x = S.counter_tensor(22, key + 'x', (2, cin, h, h))
        st = S.counter_tensor(22, key + 's', (2, 64))
        skip = S.counter_tensor(22, key + 'k', (2, 3, h // 2, h // 2))
        g = S.counter_tensor(22, key + 'g', (2, 3, h, h))
        P = {'L.' + k: v.double().requires_grad_('kernel' not in k) for k, v in sd.items()}
        xr, sr, kr = (v.double().requires_grad_(True) for v in (x, st, skip))
        (O.to_rgb(P, 'L', xr, sr, kr) * g.double()).sum().backward()
        m = m.cuda()
        xh, sh, kh = (v.cuda().requires_grad_(True) for v in (x, st, skip))
        (m(xh, sh, kh) * g.cuda()).sum().backward()
        assert _rel(xh.grad, xr.grad) <= 2e-5
        assert _rel(sh.grad, sr.grad) <= 5e-5
        assert _rel(kh.grad, kr.grad) <= 2e-5
        assert _rel(m.conv.weight.grad, P['L.conv.weight'].grad) <= 5e-5
        assert _rel(m.bias.grad, P['L.bias'].grad) <= 5e-5
        assert _rel(m.conv.modulation.weight.grad, P['L.conv.modulation.weight'].grad) <= 5e-5


def test_mapping_network_gradients():
    G = hip_generator(32, 1)
    P = {k: v.double().requires_grad_(k.startswith('style.')) for k, v in synthetic_state(32, 1).items()}
    z = S.synthetic_z(23, 3, key='bw.z')
    g = S.counter_tensor(23, 'bw.mg', (3, 512))
    zr = z.double().requires_grad_(True)
    (O.mapping(P, zr) * g.double()).sum().backward()
    zh = z.cuda().requires_grad_(True)
    (G.get_latent(zh) * g.cuda()).sum().backward()
    assert _rel(zh.grad, zr.grad) <= 5e-5
    assert _rel(G.style[1].weight.grad, P['style.1.weight'].grad) <= 5e-5
    assert _rel(G.style[8].bias.grad, P['style.8.bias'].grad) <= 5e-5


def test_generator_gradient_to_latent_small():
    """dL/dW+ through the whole Generator(64) vs autograd of the fp64 oracle."""
    G = hip_generator(64, 1)
    P = {k: v.double() for k, v in synthetic_state(64, 1).items()}
    w = S.synthetic_latents(24, 2, n_latent=G.n_latent, key='bw.w')
    tr = S.counter_tensor(24, 'bw.t', (1, 512))
    wr = w.double().requires_grad_(True)
    img, _ = O.generator_forward(P, [wr], input_is_latent=True, truncation=0.7, truncation_latent=tr.double())
    (img ** 2).mean().backward()
    for p in G.parameters():
        p.requires_grad_(False)          # only the latent needs gradients here (what the trainer consumes)
    wh = w.cuda().requires_grad_(True)
    imgh, _ = G([wh], input_is_latent=True, truncation=0.7, truncation_latent=tr.cuda())
    assert maxabs(imgh, img) <= 2e-4
    (imgh ** 2).mean().backward()
    assert _rel(wh.grad, wr.grad) <= 2e-4


def test_fused_backward_matches_the_per_layer_functions():
    """autograd.SynthesisFn (one Function for the frozen synthesis network: functional.grad_join walks every saved
    activation once, functional.styles_batched_bwd does the 20 modulations' backward in two launches) against the per-layer
    Functions it replaces, on the same generator and latents: same image bits, dL/dW+ within the reduction-order noise of the
    atomics both paths use -- and against autograd of the fp64 oracle like the per-layer path."""
    for size, B in ((64, 3), (256, 2)):
        G = hip_generator(size, 1)
        for p in G.parameters():
            p.requires_grad_(False)
        w = S.synthetic_latents(31, B, n_latent=G.n_latent, key='fz.w')
        tr = S.counter_tensor(31, 'fz.t', (1, 512)).cuda()
        gimg = S.counter_tensor(31, 'fz.g', (B, 3, size, size)).cuda()
        grads, imgs = [], []
        for fused in (True, False):
            G.fused_backward = fused
            wh = w.cuda().requires_grad_(True)
            img, lat = G([wh], input_is_latent=True, truncation=0.7, truncation_latent=tr, return_latents=True)
            assert (type(img.grad_fn).__name__ == 'SynthesisFnBackward') == fused
            assert lat.shape == (B, G.n_latent, 512)
            (img * gimg).sum().backward()
            grads.append(wh.grad.clone())
            imgs.append(img.detach())
        # (same conv kernels; the fused path accumulates ToRGB in the conv epilogues as the no-grad path does: per-tile partial sums)
        assert maxabs(imgs[0], imgs[1]) <= 2e-6 * max(1.0, float(imgs[1].abs().max()))
        assert _rel(grads[0], grads[1]) <= 1e-5
        if size == 64:
            P = {k: v.double() for k, v in synthetic_state(64, 1).items()}
            wr = w.double().requires_grad_(True)
            ref, _ = O.generator_forward(P, [wr], input_is_latent=True, truncation=0.7, truncation_latent=tr.cpu().double())
            (ref * gimg.cpu().double()).sum().backward()
            # (a random upstream gradient cancels far more than the mean(img^2) of the tests above; under --precision fp32 the dL/dx
            #  convs of both paths run on bf16 hi+lo terms -- 16 operand bits -- and measure 1.6e-3 here, the fp16 terms 1e-4)
            from stylegan_directions_face_reenactment_amd import functional as F_
            assert _rel(grads[0], wr.grad) <= (2e-4 if F_.PRECISION == 'fp16x3' else 5e-3)
        assert G.saturated_pairs() == 0


def test_fused_backward_edge_cases():
    """The whole-synthesis Function at its edges: a hooked layer switches the forward back to the per-layer Functions; fresh per-sample
    noise (randomize_noise=True) is drawn once and reused by the backward; backward twice through a retained graph gives the same
    gradient; one image; a style-mixing call (two latents) differentiates to both."""
    G = hip_generator(32, 1)
    for p in G.parameters():
        p.requires_grad_(False)
    w = S.synthetic_latents(41, 3, n_latent=G.n_latent, key='fze.w').cuda()
    wh = w.clone().requires_grad_(True)
    img, _ = G([wh], input_is_latent=True)
    assert type(img.grad_fn).__name__ == 'SynthesisFnBackward'
    (img ** 2).mean().backward(retain_graph=True)
    g1 = wh.grad.clone()
    wh.grad = None
    (img ** 2).mean().backward()
    assert _rel(wh.grad, g1) <= 1e-6
    # a hook on one layer: per-layer Functions, same gradient to summation order
    seen = []
    h = G.convs[2].register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    w2 = w.clone().requires_grad_(True)
    img2, _ = G([w2], input_is_latent=True)
    h.remove()
    assert type(img2.grad_fn).__name__ != 'SynthesisFnBackward' and seen
    (img2 ** 2).mean().backward()
    assert _rel(w2.grad, g1) <= 1e-4
    # fresh noise: finite gradients, and the SAME noise in forward and backward (the gradient of sum(img) w.r.t. the noise strength
    # of the last layer is sum over pixels of act'(.) * noise: reproducible only if the backward sees the forward's draw)
    torch.manual_seed(5)
    w3 = w[:1].clone().requires_grad_(True)
    img3, _ = G([w3], input_is_latent=True, randomize_noise=True)
    img3.sum().backward()
    assert bool(torch.isfinite(w3.grad).all()) and float(w3.grad.abs().max()) > 0
    # style mixing: two latents, an inject index
    z = S.synthetic_z(41, 2, key='fze.z').cuda()
    wa, wb = G.get_latent(z).detach().requires_grad_(True), G.get_latent(z.flip(0)).detach().requires_grad_(True)
    img4, _ = G([wa, wb], input_is_latent=True, inject_index=3)
    (img4 ** 2).mean().backward()
    assert float(wa.grad.abs().max()) > 0 and float(wb.grad.abs().max()) > 0


def test_fused_backward_parameter_gradients_match_the_per_layer_functions():
    """The same Function with every generator parameter trainable (PTI, libs/optimization.py:47-68): dL/dW of all convs, the
    modulation weights / biases, noise strengths, activation biases, ToRGB weights / biases and the constant input -- against the
    per-layer Functions on the same generator, and (conv weights) against autograd of the fp64 oracle."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    size, B = 64, 2
    G = hip_generator(size, 1).train()
    w = S.synthetic_latents(37, B, n_latent=G.n_latent, key='fzp.w').cuda()
    tr = S.counter_tensor(37, 'fzp.t', (1, 512)).cuda()
    target = torch.tanh(S.counter_tensor(37, 'fzp.g', (B, 3, size, size))).cuda()
    got = []
    for fused in (True, False):
        G.fused_backward = fused
        G.zero_grad()
        img, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
        assert (type(img.grad_fn).__name__ == 'SynthesisFnBackward') == fused
        ((img - target) ** 2).mean().backward()
        got.append({k: p.grad.clone() for k, p in G.named_parameters() if p.grad is not None})
    fused_g, layer_g = got
    assert set(fused_g) == set(layer_g) and len(fused_g) >= 5 * 9 + 4 * 5 + 1
    # (a noise strength's gradient is ONE scalar, sum_{b,c,p} g_pre * noise: a sum of random signs accumulated by atomics in both
    #  paths -- its "relative" difference is the cancellation noise of the sum itself, 1e-3 here; it is checked against the oracle below)
    worst = max((_rel(fused_g[k], layer_g[k]), k) for k in fused_g if float(layer_g[k].abs().max()) > 0 and not k.endswith('noise.weight'))
    worst_nz = max((_rel(fused_g[k], layer_g[k]), k) for k in fused_g if k.endswith('noise.weight'))
    print('fused vs per-layer parameter gradients: worst rel %.2e (%s), noise strengths %.2e (%s), %d tensors'
          % (worst[0], worst[1], worst_nz[0], worst_nz[1], len(fused_g)))
    # (the two paths' forwards differ by 3e-6 in the image -- per-layer style kernels against the batched ones -- and a weight
    #  gradient cancels: measured 2.5e-4 between the paths where the fused one is 7.6e-5 and the per-layer one 2.5e-4 from the oracle)
    assert worst[0] <= 1e-3, worst
    assert worst_nz[0] <= 1e-2, worst_nz
    P = {k: v.double() for k, v in synthetic_state(size, 1).items()}
    for k in P:
        P[k].requires_grad_(k in fused_g)
    ref, _ = O.generator_forward(P, [w.cpu().double()], input_is_latent=True, truncation=0.7, truncation_latent=tr.cpu().double())
    ((ref - target.cpu().double()) ** 2).mean().backward()
    tol = 2e-4 if F_.PRECISION == 'fp16x3' else 5e-3
    for k in ('convs.5.conv.weight', 'convs.4.conv.weight', 'conv1.conv.weight', 'to_rgbs.2.conv.weight', 'convs.3.activate.bias',
              'convs.2.conv.modulation.weight', 'to_rgbs.1.conv.modulation.bias', 'input.input'):
        assert _rel(fused_g[k], P[k].grad) <= tol, (k, _rel(fused_g[k], P[k].grad))
    nz_scale = max(float(P[k].grad.abs().max()) for k in P if k.endswith('noise.weight'))          # (scalars: against the largest of them)
    for k in fused_g:
        if k.endswith('noise.weight'):
            assert abs(float(fused_g[k]) - float(P[k].grad)) <= 10 * tol * nz_scale, k


def test_grad_join_equals_the_three_passes_it_replaces():
    """functional.grad_join on one activation: g = gu*s_next + ToRGB term, then act_grad_reduce -- against scale_reduce + torgb_bwd +
    a tensor add + act_grad_reduce (g_pre bit-identical: same expressions, a two-term sum; reductions within summation order)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    for B, C, H in ((3, 16, 4), (2, 64, 32), (2, 8, 64), (1, 24, 6), (2, 8, 5), (3, 4, 47)):      # (H = 5, 47: the scalar path, ragged chunks)
        mk = lambda key, shape, scale=1.0: S.counter_tensor(33, key, shape, 0.0, scale).cuda()
        out, gu, g_rgb = mk('o', (B, C, H, H)), mk('gu', (B, C, H, H)), mk('gr', (B, 3, H, H))
        s_next, s_rgb, w_rgb = mk('sn', (B, C)), mk('sr', (B, C)), mk('wr', (3, C))
        noise, nw, bias = mk('nz', (1, 1, H, H)), mk('nw', (1,), 0.3), mk('b', (C,), 0.5)
        for use_gu, use_rgb in ((True, True), (True, False), (False, True)):
            dx = dx2 = r = r2 = None
            if use_gu:
                dx, r = F_.scale_reduce(gu.clone(), out, s_next)
            if use_rgb:
                dx2, r2 = F_.torgb_bwd(out, g_rgb, w_rgb, s_rgb)
            g = dx + dx2 if (use_gu and use_rgb) else (dx if use_gu else dx2)
            g_pre, sums, gmax = F_.act_grad_reduce(g, out, noise, nw, bias, want_y=True, want_absmax=True)
            j_pre, j_sums, j_gmax, j_r, j_r2 = F_.grad_join(out, gu=gu if use_gu else None, s_next=s_next if use_gu else None,
                                                            g_rgb=g_rgb if use_rgb else None, w_rgb=w_rgb if use_rgb else None,
                                                            s_rgb=s_rgb if use_rgb else None, noise=noise, noise_weight=nw, bias=bias,
                                                            want_y=True)
            assert torch.equal(j_pre, g_pre) and torch.equal(j_gmax, gmax)
            assert _rel(j_sums, sums) <= 1e-5
            if use_gu:
                assert _rel(j_r, r) <= 1e-5
            else:
                assert j_r is None
            if use_rgb:
                assert _rel(j_r2, r2) <= 1e-5


@pytest.mark.parametrize('B,C,H', [(3, 16, 4), (2, 24, 8), (2, 8, 16), (1, 16, 64), (5, 8, 33)])
def test_blur_leaves_the_per_plane_maximum_of_what_it_writes(B, C, H):
    """sgdfr_blur_bias_act_f32(y_absmax=): the words are the fp32 bit patterns of max |y| per (image, channel) plane -- what a
    separate sgdfr_absmax_f32 pass would measure -- for planes of every size (one atomic per wave / per strip), and the exact range
    plan built from them equals the one built from absmax(y)."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    planes = S.counter_tensor(43, 'bm.p', (B, C, 4, H + 1, H + 1)).cuda()
    fir = torch.tensor(O.make_fir([1, 3, 3, 1], gain=4.0).numpy()).cuda()
    nz = S.counter_tensor(43, 'bm.n', (1, 1, 2 * H, 2 * H)).cuda()
    nw, bias = torch.full((1,), 0.2).cuda(), S.counter_tensor(43, 'bm.b', (C,), 0.0, 0.3).cuda()
    words = torch.zeros(B, C, dtype=torch.int32, device='cuda')
    y = F_.blur_bias_act(planes, fir, H, H, nz, nw, bias, True, absmax_out=words)
    y0 = F_.blur_bias_act(planes, fir, H, H, nz, nw, bias, True)
    assert torch.equal(y, y0)
    want = y.abs().amax((2, 3)).view(torch.int32)
    assert torch.equal(words, want)
    s_, d_ = S.counter_tensor(43, 'bm.s', (B, C), 1.0, 0.3).cuda(), S.counter_tensor(43, 'bm.d', (B, 8), 1.0, 0.2).cuda()
    a = F_.split_range(s_, d_, words)
    b = F_.split_range(s_, d_, F_.absmax(y, per_image=True))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_fused_adam_equals_torch_adam():
    """finetune.FusedAdam (sgdfr_adam_f32: every parameter tensor of the step in one launch, step count on the device) against
    torch.optim.Adam with the reference's settings (libs/optimization.py:41: default betas / eps, no weight decay) over several steps;
    the parameters' version counters move (the weight packs of ModulatedConv2d are keyed on them)."""
    from stylegan_directions_face_reenactment_amd.finetune import FusedAdam
    torch.manual_seed(3)
    shapes = [(1, 64, 32, 3, 3), (512,), (1,), (130, 512), (3, 7)]
    a = [torch.randn(*sh, device='cuda').requires_grad_(True) for sh in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa, ob = FusedAdam(a, lr=3e-3), torch.optim.Adam(b, lr=3e-3)
    for step in range(6):
        grads = [torch.randn_like(t) * (10.0 ** (step - 3)) for t in a]
        for t, u, g in zip(a, b, grads):
            t.grad, u.grad = g.clone(), g.clone()
        v0 = [t._version for t in a]
        oa.step()
        ob.step()
        assert all(t._version > v for t, v in zip(a, v0))
        for t, u in zip(a, b):
            assert torch.allclose(t, u, rtol=2e-6, atol=1e-7), (step, tuple(t.shape), maxabs(t, u))
    assert float(oa.step_count) == 6.0
    oa.zero_grad()
    assert all(t.grad is None for t in a)


def test_batched_style_backward_and_small_parameter_gradients_match_torch():
    """functional.styles_batched_bwd (ds of demodulated / plain / ToRGB layers, latent gradient summed per latent row, modulation weight
    and bias gradients), functional.demod_dq and functional.param_grads against the tensor expressions of the per-layer Functions
    (autograd.StyleFn / StyledConvFn / ToRGBFn backward) in fp64."""
    from stylegan_directions_face_reenactment_amd import functional as F_, _native as N
    torch.manual_seed(0)
    B, L, D = 5, 4, 512
    dev = 'cuda'
    rnd = lambda *shape: torch.randn(*shape, device=dev)
    latent = rnd(B, L, D)
    layers = []          # (kind, latent row, cin, cout)
    for kind, li, cin, cout in (('demod', 0, 64, 48), ('rgb', 1, 48, 3), ('demod', 1, 48, 130), ('plain', 2, 130, 16), ('rgb', 3, 16, 3), ('demod', 3, 512, 512)):
        e = {'latent_index': li, 'mod_w': rnd(cin, D), 'want_w': True, 'want_b': True}
        if kind == 'rgb':
            e['rgb_r'], e['rgb_w'] = rnd(B, 3, cin), rnd(3, cin)
        else:
            e['gs'] = rnd(B, cin)
            if kind == 'demod':
                sums = rnd(B, cout, 3)
                e['a'], e['d'], e['s'], e['qt'] = sums[:, :, 2], rnd(B, cout).abs() + 0.5, rnd(B, cin), rnd(cin, cout).abs()
        layers.append((kind, e))
    glat = F_.styles_batched_bwd([e for _, e in layers], B, L, D, latent=latent)
    want = torch.zeros(B, L, D, dtype=torch.float64, device=dev)
    for kind, e in layers:
        cin = e['mod_w'].shape[0]
        if kind == 'rgb':
            ds = (e['rgb_r'].double() * e['rgb_w'].double().unsqueeze(0)).sum(1) / cin ** 0.5
        elif kind == 'plain':
            ds = e['gs'].double()
        else:
            a, d = e['a'].double(), e['d'].double()
            ds = e['gs'].double() + e['s'].double() * ((-(a / d) * d ** 3) @ e['qt'].double().t())
        want[:, e['latent_index']] += ds @ e['mod_w'].double() / D ** 0.5
        assert _rel(e['gmod_w'], ds.t() @ latent[:, e['latent_index']].double() / D ** 0.5) <= 2e-5
        assert _rel(e['gmod_b'], ds.sum(0)) <= 2e-5
    assert _rel(glat, want) <= 2e-5
    # dL/dQ of the demodulation
    cout, cin = 48, 64
    sums, d, s_ = rnd(B, cout, 3), rnd(B, cout).abs() + 0.5, rnd(B, cin)
    a = sums[:, :, 2]
    dq = F_.demod_dq(a, d, s_)
    coeff = (a.double() / d.double()) * d.double() ** 3 * -0.5
    assert _rel(dq, coeff.t() @ (s_.double() ** 2)) <= 2e-5
    # bias / noise strength / ToRGB weight / ToRGB bias gradients in one launch
    C, HW = 40, 33 * 33
    sums2, r_rgb, s_rgb, g_rgb = rnd(B, C, 3), rnd(B, 3, C), rnd(B, C), rnd(B, 3, 33, 33)
    gb, gn, gw, gbr = F_.param_grads([(N.PGRAD_BIAS, sums2, None, C, 0), (N.PGRAD_NOISE, sums2, None, C, 0),
                                      (N.PGRAD_RGB_W, r_rgb, s_rgb, C, 0), (N.PGRAD_RGB_B, g_rgb, None, 3, HW)], B)
    assert _rel(gb, sums2[:, :, 0].double().sum(0)) <= 1e-5 and _rel(gn, sums2[:, :, 1].double().sum().view(1)) <= 1e-5
    assert _rel(gw, (r_rgb.double() * s_rgb.double().unsqueeze(1)).sum(0) / C ** 0.5) <= 1e-5
    assert _rel(gbr, g_rgb.double().sum((0, 2, 3))) <= 1e-5


def test_direction_matrix_gradient_golden():
    """KAT-5: dL/dA for L = mean(img^2) through generate_image, z path and W+ path, vs the REAL reference's autograd."""
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
    import warnings
    for path, code, is_lat in (('z', z, False), ('w', w, True)):
        A.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')       # G's own parameters keep requires_grad=True like in libs/trainer.py
            img, lat = generate_image(G, code, 0.7, trunc, shift_code=A(sv), input_is_latent=is_lat, return_latents=True)
            (img ** 2).mean().backward()
        ref = t(g['%s.gA' % path])
        print('KAT-5 %s path: rel err dL/dA %.2e, dL/db %.2e' % (path, _rel(A.linear.weight.grad, ref), _rel(A.linear.bias.grad, t(g['%s.gAb' % path]))))
        assert _rel(A.linear.weight.grad, ref) <= 1e-4, path      # (measured 2.5e-5 .. 3.8e-5 in every arithmetic: the fp32 reference's own rounding)
        assert _rel(A.linear.bias.grad, t(g['%s.gAb' % path])) <= 1e-4, path


_C5_ORACLE = {}


def _config5_oracle(B, lr, wd):
    """fp64 oracle of one direction-learning step at config 5's per-rank shape, by torch autograd, in chunks of 2 images
    (bounded memory); cached: both arithmetics compare against the same numbers."""
    if B in _C5_ORACLE:
        return _C5_ORACLE[B]
    from oracle import shift_oracle as SO
    g8 = golden('kat8_shift.npz')
    cfg = SO.initialize_directions('voxceleb', 15, 6.0, g8['ranges_voxceleb'])
    ang_s, par_s = S.synthetic_shape_params(SEED, 'c5.src', B)
    ang_t, par_t = S.synthetic_shape_params(SEED, 'c5.tgt', B)
    which = (torch.arange(B // 2) * 7 + 3) % 15                              # injected draws (np.random.choice / torch.rand in
    u = S.counter_tensor(SEED, 'c5.u', (B // 2,), 0.5, 0.25).clamp_(0.0, 1.0)   # utils_train.py:184-190)
    sv = torch.as_tensor(SO.make_shift_vector_50(cfg, par_s, par_t, ang_s, ang_t, which.numpy(), u))
    P = O.cast_state(synthetic_state(256, 1), torch.float64)
    A0 = S.synthetic_direction_state(SEED)
    A = {k: v.double().requires_grad_(True) for k, v in A0.items()}
    z = S.synthetic_z(SEED, B, key='c5.z')
    trunc = O.mapping(P, S.synthetic_z(SEED, 64, key='c5.tz').double()).mean(0, keepdim=True)      # mean latent of a fixed z batch
    loss = 0.0
    for lo in range(0, B, 2):
        img = O.generate_image(P, z[lo:lo + 2].double(), 0.7, trunc, shift_code=O.direction_matrix(A, sv[lo:lo + 2].double()),
                               input_is_latent=False)
        part = (img ** 2).sum() / (B * img[0].numel())                      # L = mean(img^2) over the whole batch
        part.backward()
        loss += float(part)
    grads = {k: v.grad.clone() for k, v in A.items()}
    opt = torch.optim.Adam(list(A.values()), lr=lr, weight_decay=wd)        # trainer.py:145
    opt.step()
    _C5_ORACLE[B] = dict(sv=sv, which=which, u=u, z=z, trunc=trunc.float(), loss=loss, grads=grads,
                         stepped={k: v.detach().clone() for k, v in A.items()}, A0=A0,
                         shape=((ang_s, par_s), (ang_t, par_t)))
    return _C5_ORACLE[B]


@pytest.mark.timeout(1500)
def test_config5_direction_step_at_per_rank_size():
    """BASELINE configs[4] at its REAL per-rank shape (B=16, 256x256, cm=1): the step of libs/trainer.py:155-189
    with G frozen -- make_shift_vector_50 (injected draws, utils_train.py:177-288) -> A -> shift -> generate_image(z, psi=0.7)
    -> L = mean(img^2) -> backward -> Adam(lr 1e-4, weight decay 5e-4) -- against torch autograd of the fp64 oracle: shift
    vectors bit-exact, loss, dL/dA, dL/db within 1e-4 relative, and the updated A."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    from stylegan_directions_face_reenactment_amd.shift import ShiftVectors
    B, lr, wd = 16, 1e-4, 5e-4
    ref = _config5_oracle(B, lr, wd)
    G = hip_generator(256, 1)
    for p in G.parameters():
        p.requires_grad_(False)                                             # only A is optimised (trainer.py:144)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(ref['A0'])
    A = A.cuda()
    opt = torch.optim.Adam(A.parameters(), lr=lr, weight_decay=wd)
    (ang_s, par_s), (ang_t, par_t) = ref['shape']
    cu = lambda d: {k: v.cuda() for k, v in d.items()}
    shifts = ShiftVectors('voxceleb', 15, 6.0, ranges=golden('kat8_shift.npz')['ranges_voxceleb'])
    sv, which = shifts.make_shift_vector_50(cu(par_s), cu(par_t), ang_s.cuda(), ang_t.cuda(), target_indices=ref['which'].numpy(),
                                            u=ref['u'].cuda())
    assert torch.equal(sv.cpu(), ref['sv'].float()) and torch.equal(which.cpu().long(), ref['which'])
    img = generate_image(G, ref['z'].cuda(), 0.7, ref['trunc'].cuda(), shift_code=A(sv), input_is_latent=False)
    assert img.shape == (B, 3, 256, 256)
    loss = (img ** 2).mean()
    A.zero_grad()
    loss.backward()
    gw, gb = A.linear.weight.grad.clone(), A.linear.bias.grad.clone()
    opt.step()
    ew, eb = _rel(gw, ref['grads']['linear.weight']), _rel(gb, ref['grads']['linear.bias'])
    el = abs(float(loss) - ref['loss']) / abs(ref['loss'])
    # Adam's first step is lr * g / (|g| + 1e-8) ~ lr * sign(g): where the gradient is not vanishing against the largest one
    # (elements at <= 1e-3 of it carry the 1e-4 relative error as a 10 % or larger error of their own) the updates must agree
    sig_w = ref['grads']['linear.weight'].abs() >= 1e-3 * ref['grads']['linear.weight'].abs().max()
    sig_b = ref['grads']['linear.bias'].abs() >= 1e-3 * ref['grads']['linear.bias'].abs().max()
    dw = (A.linear.weight.detach().cpu().double() - ref['stepped']['linear.weight']).abs()
    db = (A.linear.bias.detach().cpu().double() - ref['stepped']['linear.bias']).abs()
    step_sig = max(float(dw[sig_w].max()), float(db[sig_b].max()))
    step_all = max(float(dw.max()), float(db.max()))
    print('config 5 @B=16 256^2 [%s fwd / %s bwd]: loss rel %.1e, dL/dA rel %.2e, dL/db rel %.2e; A after Adam (lr %.0e): max abs diff '
          '%.2e on the %.0f %% significant elements, %.2e overall'
          % (F_.PRECISION, F_.BACKWARD_ARITH if F_.PRECISION != 'fp32' else 'fp32', el, ew, eb, lr, step_sig,
             100.0 * float(sig_w.float().mean()), step_all))
    assert el <= 1e-5 and ew <= 1e-4 and eb <= 1e-4
    assert step_sig <= 0.02 * lr and step_all <= 2.0 * lr + 1e-9
    assert G.saturated_pairs() == 0                  # nothing, forward or backward, left the fp16 range plans


@pytest.mark.parametrize('cin,cout,h,up,B', [(16, 8, 8, False, 3), (8, 16, 4, True, 3), (64, 128, 16, False, 2),
                                              (128, 64, 16, True, 2), (64, 64, 64, False, 1), (32, 64, 64, True, 1),
                                              (6, 10, 7, False, 2), (10, 6, 5, True, 2), (64, 64, 128, False, 1)])
def test_conv_weight_gradient(cin, cout, h, up, B):
    """dL/dW of the modulated conv (what PTI optimises, libs/optimization.py:32-35), incl. the demodulation path."""
    from stylegan_directions_face_reenactment_amd.model import StyledConv
    key = 'wg.%d.%d.%d.%d' % (cin, cout, h, up)
    m = StyledConv(cin, cout, 3, 64, upsample=up)
    sd = {k: S.counter_tensor(25, key + k, tuple(v.shape)) for k, v in m.state_dict().items() if 'kernel' not in k}
    sd['conv.modulation.bias'] = sd['conv.modulation.bias'] * 0.1 + 1.0
    if up:
        sd['conv.blur.kernel'] = m.conv.blur.kernel
    m.load_state_dict(sd)
    x = S.counter_tensor(25, key + 'x', (B, cin, h, h))
    st = S.counter_tensor(25, key + 's', (B, 64))
    r = 2 * h if up else h
    nz = S.counter_tensor(25, key + 'n', (1, 1, r, r))
    g = S.counter_tensor(25, key + 'g', (B, cout, r, r))
    P = {'L.' + k: v.double().requires_grad_(k == 'conv.weight') for k, v in sd.items()}
    yo = O.styled_conv(P, 'L', x.double(), st.double(), nz.double(), upsample=up)
    # an output within rounding of 0 takes either slope of the leaky ReLU depending on the last bit of any fp32 kernel (one
    # such element among 262144 moves dL/dW by 1.7e-3): those elements get no upstream gradient on either side
    g = g * (yo.detach().abs() > 1e-5).float()
    (yo * g.double()).sum().backward()
    m = m.cuda()
    (m(x.cuda(), st.cuda(), noise=nz.cuda()) * g.cuda()).sum().backward()
    assert m.conv.weight.grad is not None and m.conv.weight.grad.shape == (1, cout, cin, 3, 3)
    assert _rel(m.conv.weight.grad, P['L.conv.weight'].grad) <= 1e-4


def test_pti_style_step_updates_generator():
    """One optimiser step on convs[4..11] parameters as in libs/optimization.py:32-68 (B=1, MSE to a target)."""
    import copy
    G = hip_generator(64, 1)
    P = {k: v.double() for k, v in synthetic_state(64, 1).items()}
    names = [n for n, _ in G.named_parameters() if n.startswith('convs.4.') or n.startswith('convs.7.')]
    for k in P:
        P[k].requires_grad_(k in names)
    w = S.synthetic_latents(26, 1, n_latent=G.n_latent, key='pti.w')
    target = S.counter_tensor(26, 'pti.t', (1, 3, 64, 64))
    img, _ = O.generator_forward(P, [w.double()], input_is_latent=True)
    ((img - target.double()) ** 2).mean().backward()
    G2 = copy.deepcopy(G).train()
    params = [p for n, p in G2.named_parameters() if n in names]
    opt = torch.optim.Adam(params, lr=3e-4)
    imgh, _ = G2([w.cuda()], input_is_latent=True)
    loss = ((imgh - target.cuda()) ** 2).mean()
    loss.backward()
    for n, p in G2.named_parameters():
        if n in names:
            assert p.grad is not None, n
            assert _rel(p.grad, P[n].grad) <= 5e-4, n
    before = G2.convs[4].conv.weight.detach().clone()
    opt.step()
    assert float((G2.convs[4].conv.weight - before).abs().max()) > 0
    with torch.no_grad():
        img2, _ = G2([w.cuda()], input_is_latent=True)      # repacked weights are picked up
    assert float(((img2 - target.cuda()) ** 2).mean()) < float(loss)


def test_pti_driver_graph_replay_matches_eager_steps():
    """finetune.optimize_g (counterpart of libs/optimization.py:25-72): 12 steps replayed as one captured hipGraph end at the
    same weights as 12 eager steps, the loss goes down, and only convs[4..11] move."""
    import copy
    from stylegan_directions_face_reenactment_amd import finetune
    G0 = hip_generator(256, 1)
    w = S.synthetic_latents(SEED, 1, n_latent=G0.n_latent, key='pti.w').cuda()
    trunc = S.counter_tensor(SEED, 'pti.t', (1, 512)).cuda()
    with torch.no_grad():
        base, _ = G0([w], input_is_latent=True, truncation=0.7, truncation_latent=trunc)
    target = (base + 0.3 * S.counter_tensor(SEED, 'pti.d', tuple(base.shape)).cuda()).clamp(-1, 1)
    runs = {}
    for graph in (False, True):
        G = copy.deepcopy(G0)
        first = finetune.l2_loss_fn(G([w], input_is_latent=True, truncation=0.7, truncation_latent=trunc)[0], target, 100).item()
        G, loss = finetune.optimize_g(G, w, target, trunc, opt_steps=12, lr=1e-3, graph=graph)
        assert loss.item() < first
        runs[graph] = (G, loss.item())
    Ge, Gg = runs[False][0], runs[True][0]
    assert abs(runs[False][1] - runs[True][1]) <= 2e-3 * abs(runs[False][1])
    for (k, a), (_, b), (_, c) in zip(Ge.state_dict().items(), Gg.state_dict().items(), G0.state_dict().items()):
        moved = k.startswith('convs.') and 4 <= int(k.split('.')[1]) <= 11 and 'kernel' not in k
        if moved:
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-4), k
        else:
            assert torch.equal(a, c) and torch.equal(b, c), k


@pytest.mark.parametrize('cin,cout,h,up', [(64, 64, 40, False), (256, 128, 16, True), (32, 48, 19, True), (128, 128, 16, False)])
def test_backward_arithmetics_of_the_split_dx_convs(cin, cout, h, up, capsys):
    """dL/dx of a StyledConv on the split kernels in both backward arithmetics (functional.BACKWARD_ARITH): bf16 terms (16 operand
    bits, fp32 range) and range-planned fp16 terms (22 bits, the forward's arithmetic; e from the true max |g| of each image),
    also with gradients scaled far away from 1 -- the plan makes the fp16 form scale-free."""
    from stylegan_directions_face_reenactment_amd import functional as F_
    from stylegan_directions_face_reenactment_amd.model import StyledConv
    if F_.PRECISION == 'fp32':
        pytest.skip('the fp32 kernels have one backward arithmetic')
    B = 2
    key = 'bwa.%d.%d.%d.%d' % (cin, cout, h, up)
    m = StyledConv(cin, cout, 3, 64, upsample=up)
    sd = {k: S.counter_tensor(23, key + k, tuple(v.shape)) for k, v in m.state_dict().items() if 'kernel' not in k}
    sd['conv.modulation.bias'] = sd['conv.modulation.bias'] * 0.1 + 1.0
    if up:
        sd['conv.blur.kernel'] = m.conv.blur.kernel
    m.load_state_dict(sd)
    x = S.counter_tensor(23, key + 'x', (B, cin, h, h))
    st = S.counter_tensor(23, key + 's', (B, 64))
    r = 2 * h if up else h
    nz = S.counter_tensor(23, key + 'n', (1, 1, r, r))
    g = S.counter_tensor(23, key + 'g', (B, cout, r, r))
    P = {'L.' + k: v.double() for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    (O.styled_conv(P, 'L', xr, st.double(), nz.double(), upsample=up) * g.double()).sum().backward()
    m = m.cuda()
    for p_ in m.parameters():
        p_.requires_grad_(False)
    errs = {}
    for arith in ('bf16x3', 'fp16x3'):
        for scale in (1.0, 2.0 ** -30, 2.0 ** 20):
            xh = x.cuda().requires_grad_(True)
            with F_.using(F_.config().replace(backward_arith=arith)):      # the Function remembers the forward's config for its backward
                out = m(xh, st.cuda(), noise=nz.cuda())
            (out * (g.cuda() * scale)).sum().backward()
            errs[(arith, scale)] = _rel(xh.grad / scale, xr.grad)
    with capsys.disabled():
        print('\n  dx rel err %s: ' % key + ', '.join('%s@2^%d %.1e' % (a, round(__import__('math').log2(sc)), e) for (a, sc), e in errs.items()))
    assert F_.split_saturation_count(reset=True) == 0
    assert all(e <= 2e-5 for (a, sc), e in errs.items() if a == 'bf16x3')
    assert all(e <= 4e-6 for (a, sc), e in errs.items() if a == 'fp16x3')
