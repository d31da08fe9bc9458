"""Where the fused up-conv + blur launch (csrc/upfir.hip) spends its time: the layer alone at B=64 with the probe build's ablation
switches (SGDFR_SPLIT_DBG: 2 = no epilogue, 4 = no K loop; wrong results), next to the two-pass form.
    python scripts/upfir_probe.py build      (in the container: csrc/libsgdfr_hip_upfir_probe.so, travels with the snapshot)
    python scripts/upfir_probe.py [cin cout H]     (on the GPU box; re-executes itself per switch)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'stylegan_directions_face_reenactment_amd', 'csrc', 'libsgdfr_hip_upfir_probe.so')


def build():
    from stylegan_directions_face_reenactment_amd import build_native as b
    b.build()
    obj = os.path.join(b.CSRC, 'upfir_probe.o')
    subprocess.run([b._hipcc()] + b.FLAGS + ['-DSGDFR_SPLIT_PROBE', '-c', os.path.join(b.CSRC, 'upfir.hip'), '-o', obj], check=True)
    objs = [s[:-4] + '.o' for s in b.sources() if not s.endswith('upfir.hip')] + [obj]
    subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', LIB] + objs, check=True)
    print(LIB)


def leg(cin, cout, H):
    import torch
    from stylegan_directions_face_reenactment_amd import functional as F_, synthetic as S
    B = 64
    x = S.counter_tensor(3, 'p.x', (B, cin, H, H)).cuda()
    w = S.counter_tensor(3, 'p.w', (1, cout, cin, 3, 3)).cuda()
    s = S.counter_tensor(3, 'p.s', (B, cin), 1.0, 0.3).cuda()
    d = torch.ones(B, cout).cuda()
    sn = torch.ones(B, cout).cuda()
    nz = S.counter_tensor(3, 'p.n', (1, 1, 2 * H, 2 * H)).cuda()
    nw, bias = torch.full((1,), 0.3).cuda(), torch.zeros(cout).cuda()
    k = torch.tensor([1., 3., 3., 1.])
    fir = (torch.outer(k, k) / 16).cuda()
    wsp, xs = F_.prepack_split(w, 'fp16x3'), F_.to_split(x, s, 'fp16x3')
    ps = ((H + 1) * (H + 1) + 31) // 32 * 32

    def fused():
        return F_.modconv_upfir_split(xs, (B, cin, H, H), wsp, d, cout, fir, sn, nz, nw, bias, True, arith='fp16x3')

    def two_pass():
        planes = F_.modconv_split(xs, wsp, None, d, cout, mode=F_.N.MODE_UP3, x_split=(B, cin, H, H), plane_stride=ps, arith='fp16x3')
        return F_.blur_bias_act_split(planes, fir, H, H, sn, nz, nw, bias, True, plane_stride=ps, arith='fp16x3')

    out = {}
    for name, fn in (('fused', fused), ('two_pass', two_pass)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 20 * 1e3
    print('dbg=%s  %d->%d @%dx%d  fused %.1f us   two-pass (conv + blur) %.1f us' % (os.environ.get('SGDFR_SPLIT_DBG', '0'), cin, cout, H, H,
                                                                                   out['fused'], out['two_pass']), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    elif os.environ.get('UPFIR_PROBE_LEG'):
        leg(*[int(a) for a in sys.argv[1:4]])
    else:
        shape = sys.argv[1:4] if len(sys.argv) >= 4 else ['128', '64', '128']
        for dbg in ('0', '2', '4'):
            env = dict(os.environ, SGDFR_LIB=LIB, SGDFR_ALLOW_LIB_OVERRIDE='1', SGDFR_SPLIT_DBG=dbg, UPFIR_PROBE_LEG='1')
            subprocess.run([sys.executable, os.path.abspath(__file__)] + shape, env=env)
