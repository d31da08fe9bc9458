"""Stand-alone timing of the blur's two hand-over forms on the generator's levels (B=64): the next conv's split input (4 bytes per
element) against its Winograd input form (8 bytes per element):  python scripts/blur_wino_time.py [--batch 64]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import functional as F_      # noqa: E402


def timed(fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
B = args.batch
fir = torch.tensor([[1., 3., 3., 1.]]).cuda()
fir = fir.t() @ fir
fir = fir / fir.sum() * 4
for C, H in ((512, 8), (512, 16), (256, 32), (128, 64)):
    ps = ((H + 1) * (H + 1) + 31) // 32 * 32
    planes = torch.randn(B, C, 4, ps, device='cuda')
    nz = torch.randn(1, 1, 2 * H, 2 * H, device='cuda')
    nw = torch.full((1,), 0.1, device='cuda')
    bias = torch.randn(C, device='cuda')
    sn = torch.randn(B, C, device='cuda')
    best = [1e9, 1e9, 1e9]
    for _ in range(3):
        for k, wino in enumerate((0, 2, 4)):
            best[k] = min(best[k], timed(lambda: F_.blur_bias_act_split(planes, fir, H, H, sn, nz, nw, bias, True, plane_stride=ps, wino=wino)))
    mb = B * C * 4 * H * H * 4 / 1e6
    print('blur %3d ch %3d -> %3d | split form %.0f us (%.2f TB/s) | F(2,3) form %.0f us (%.2f TB/s) +%.0f us | F(4,3) form %.0f us (%.2f TB/s) +%.0f us'
          % (C, H, 2 * H, best[0], 2 * mb / best[0], best[1], 3 * mb / best[1], best[1] - best[0], best[2], 2.5 * mb / best[2],
             best[2] - best[0]), flush=True)
