"""DOWN3 split conv per layer (B=64 shapes of the 256x256 generator), optional SGDFR_SPLIT_DBG probe bits (8: no x DMA, 16: no w DMA)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_, _native as N
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 64
out = []
for C, cin, h in [(64, 128, 128), (128, 256, 64), (256, 512, 32), (512, 512, 16)]:
    w = torch.randn(1, C, cin, 3, 3, device='cuda'); gT = torch.randn(B, C, 4, h + 1, h + 1, device='cuda')
    d = torch.rand(B, C, device='cuda') + 0.5
    xs = F_.planes_to_split(gT, d, 'bf16x3'); del gT
    wsp = F_.prepack_split(w, 'bf16x3', adjoint='down')
    t = bench(lambda: F_.modconv_split(xs, wsp, None, None, cin, mode=N.MODE_DOWN3, arith='bf16x3', x_split=(B, C, h, h), batch=B))
    fl = B * F_.conv_flops(cin, C, h, h, upsample=True)
    out.append('%d<-%d@%d %.0f us %.0f TF' % (cin, C, h, t, fl / t / 1e6))
print('dbg=%s: ' % os.environ.get('SGDFR_SPLIT_DBG', '0') + ' | '.join(out), flush=True)
