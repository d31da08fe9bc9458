"""fp16 + fp8 cross terms (SGDFR_SPLIT_FP16F8) on the wide F(4,3) kernel against fp16x3, layer by layer (GPU box): time per launch in
one process and the error of both against an fp64 evaluation of the same modulated conv.   B=64 python scripts/f8_layer_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_

B = int(os.environ.get('B', 64))
EB = int(os.environ.get('EB', 2))          # images of the fp64 check
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


for cin, cout, h in [(512, 512, 32), (256, 256, 64), (128, 128, 128)]:
    w = torch.randn(1, cout, cin, 3, 3, device='cuda')
    # loudness as the generator's range plan leaves it: the layer's calibrated max |x * s| * 2^-4 near 2^10, single images up to 2^4 quieter
    x = torch.randn(B, cin, h, h, device='cuda') * 2.0 ** (12 - 4 * torch.rand(B, 1, 1, 1, device='cuda'))
    x = torch.where(x > 0, x, 0.2 * x)
    s = torch.randn(B, cin, device='cuda') * 0.3 + 1.0
    wd = (w[0].double()[None] * s.double()[:EB, None, :, None, None]) / (cin * 9) ** 0.5
    d = torch.rsqrt(wd.pow(2).sum(dim=(2, 3, 4)) + 1e-8).float()
    d = torch.cat([d, torch.rand(B - EB, cout, device='cuda') + 0.5]) if B > EB else d
    bias = torch.randn(cout, device='cuda') * 0.1
    nw = torch.full((1,), 0.1, device='cuda')
    noise = torch.randn(1, 1, h, h, device='cuda')
    sn = torch.randn(B, cout, device='cuda') * 0.3 + 1.0
    rgb = (torch.randn(3, cout, device='cuda'), torch.randn(B, cout, device='cuda') * 0.3 + 1.0)
    # fp64 reference of the first EB images (no activation: the raw conv is what the arithmetic changes)
    ref = torch.stack([torch.nn.functional.conv2d(x[i:i + 1].double(), wd[i], padding=1)[0] * d[i].double()[:, None, None] for i in range(EB)])
    line = '%d->%d@%d B=%d:' % (cin, cout, h, B)
    for arith in ('fp16x3', 'fp16f8'):
        vs = F_.to_wsplit(x, s, arith, f=4)
        wsp = F_.prepack_wsplit(w, arith, f=4)
        os.environ['SGDFR_WSPLIT_WIDE_NOW'] = '2'
        y = F_.modconv_wsplit(vs, (B, cin, h, h), wsp, d, cout, None, None, torch.zeros_like(bias), False, arith=arith, f=4)
        err = (y[:EB].double() - ref).abs().max().item() / ref.abs().max().item()
        t = bench(lambda: F_.modconv_wsplit(vs, (B, cin, h, h), wsp, d, cout, noise, nw, bias, True, arith=arith, f=4, rgb=rgb, s_next=sn,
                                            want_y=False))
        line += '  %s %.0f us, err %.2e of max|y|' % (arith, t, err)
        del vs, wsp, y
    print(line, flush=True)
