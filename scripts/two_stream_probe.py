"""Do consecutive independent forwards overlap usefully when they alternate between two HIP streams (the head of a forward --
4x4 ... 16x16 layers, K-sliced, under-filling the chip, and ~200 us of launch gaps -- against the big layers of the previous
one)?  python scripts/two_stream_probe.py [--batch 64] [--streams 2]"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                  # noqa: E402
from stylegan_directions_face_reenactment_amd import synthetic as S          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--streams', type=int, default=2)
a = ap.parse_args()
args = bench.parse_args(['--batch', str(a.batch)])
dev = torch.device('cuda:0')
G = bench.build_generator(args, 0, dev)
G.use_graphs = False
ws = [S.synthetic_latents(bench.SEED, a.batch, n_latent=G.n_latent, key='two.w%d' % i).to(dev) for i in range(a.streams)]
streams = [torch.cuda.Stream() for _ in range(a.streams)]


def run(nstreams, steps):
    outs = [None] * nstreams
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            outs[k] = G([ws[k]], input_is_latent=True)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, outs


with torch.no_grad():
    for n in (1, a.streams):
        run(n, 10)
    ref = run(1, 2)[1][0].clone()
    for rep in range(3):
        for n in (1, a.streams):
            ms, outs = run(n, a.steps)
            print('%d stream(s): %.3f ms per forward = %.0f frames/s' % (n, ms, a.batch / ms * 1e3), flush=True)
    print('same images:', bool(torch.equal(run(a.streams, a.streams)[1][0], ref)))
