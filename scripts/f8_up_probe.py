"""fp16 + fp8 cross terms (SGDFR_SPLIT_FP16F8) on the transposed conv's deep plan against fp16x3, layer by layer (GPU box): time per
launch in one process and the error of the parity planes against an fp64 evaluation.   B=64 python scripts/f8_up_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N

B = int(os.environ.get('B', 64))
EB = 2
torch.manual_seed(0)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, h in [(512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda')
    x = torch.randn(B, cin, h, h, device='cuda') * 2.0 ** (12 - 4 * torch.rand(B, 1, 1, 1, device='cuda'))
    x = torch.where(x > 0, x, 0.2 * x)
    s = torch.randn(B, cin, device='cuda') * 0.3 + 1.0
    d = torch.rand(B, cout, device='cuda') + 0.5
    assert N.load().sgdfr_modconv2d_split_f8_ok(B, cin, cout, h, h, N.MODE_UP3)
    # fp64 transposed conv of the first EB images: planes T[.., py, px, a, b] = conv_transpose2d(x*s, w, stride 2)[2a+py, 2b+px]
    xs64 = (x[:EB].double() * s[:EB].double()[:, :, None, None])
    t = torch.nn.functional.conv_transpose2d(xs64, (w[0].double() / (cin * 9) ** 0.5).transpose(0, 1), stride=2) * d[:EB].double()[:, :, None, None]
    t = torch.nn.functional.pad(t, (0, 1, 0, 1))                      # (2h+1)^2 -> (2h+2)^2
    ref = t.view(EB, cout, h + 1, 2, h + 1, 2).permute(0, 1, 3, 5, 2, 4).reshape(EB, cout, 4, h + 1, h + 1)
    line = 'up %d->%d@%d B=%d:' % (cin, cout, h, B)
    ps = ((h + 1) * (h + 1) + 31) // 32 * 32
    for arith in ('fp16x3', 'fp16f8'):
        xs = F_.to_split(x, s, arith)
        wsp = F_.prepack_split(w, arith)
        planes = F_.modconv_split(xs, wsp, None, d, cout, arith=arith, mode=N.MODE_UP3, x_split=(B, cin, h, h), batch=B)
        err = (planes[:EB].double() - ref).abs().max().item() / ref.abs().max().item()
        buf = torch.empty(B, cout, 4, ps, device='cuda')
        tm = bench(lambda: F_.modconv_split(xs, wsp, None, d, cout, arith=arith, mode=N.MODE_UP3, x_split=(B, cin, h, h), batch=B, out=buf, plane_stride=ps))
        line += '  %s %.0f us, err %.2e of max|T|' % (arith, tm, err)
        del xs, wsp, planes, buf
    print(line, flush=True)

# the direct plain layer of the bench generator (4-wave plan, fused ToRGB, nothing else stored)
for cin, cout, h in [(64, 64, 256)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda')
    x = torch.randn(B, cin, h, h, device='cuda') * 2.0 ** (12 - 4 * torch.rand(B, 1, 1, 1, device='cuda'))
    x = torch.where(x > 0, x, 0.2 * x)
    s = torch.randn(B, cin, device='cuda') * 0.3 + 1.0
    d = torch.rand(B, cout, device='cuda') + 0.5
    bias = torch.zeros(cout, device='cuda')
    rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda') * 0.3 + 1.0)
    assert N.load().sgdfr_modconv2d_split_f8_ok(B, cin, cout, h, h, N.MODE_PLAIN3)
    ref = torch.nn.functional.conv2d(x[:EB].double() * s[:EB].double()[:, :, None, None], w[0].double() / (cin * 9) ** 0.5, padding=1) * d[:EB].double()[:, :, None, None]
    line = 'plain %d->%d@%d B=%d:' % (cin, cout, h, B)
    for arith in ('fp16x3', 'fp16f8'):
        xs = F_.to_split(x, s, arith)
        wsp = F_.prepack_split(w, arith)
        y = F_.modconv_split(xs, wsp, None, d, cout, None, None, bias, False, arith=arith, x_split=(B, cin, h, h), batch=B)
        err = (y[:EB].double() - ref).abs().max().item() / ref.abs().max().item()
        del y
        tm = bench(lambda: F_.modconv_split(xs, wsp, None, d, cout, None, None, bias, True, arith=arith, x_split=(B, cin, h, h), batch=B, rgb=rgb, want_y=False))
        line += '  %s %.0f us, err %.2e of max|y|' % (arith, tm, err)
        del xs, wsp
    print(line, flush=True)
