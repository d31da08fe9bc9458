"""Per-forward time of the no-grad generator at small batches (launch-bound regime), eager launches vs the default hipGraph
replay of Generator.forward, raw and through generate_image (verified forwards): python scripts/small_batch_time.py"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import synthetic as S, functional as F_
from stylegan_directions_face_reenactment_amd.model import Generator
G = Generator(256, 512, 8, channel_multiplier=1)
G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7))
G = G.eval().cuda()
from stylegan_directions_face_reenactment_amd.generic import generate_image
for B, graphs in [(b, g) for b in (1, 2, 4, 8, 32) for g in (False, True)]:
    G.use_graphs = graphs
    w = S.synthetic_latents(7, B, n_latent=14).cuda()
    with torch.no_grad():
        for _ in range(5): G([w], input_is_latent=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): G([w], input_is_latent=True)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
        # host time alone: how long does Python need to ENQUEUE one forward?
        t1 = time.perf_counter()
        for _ in range(50): G([w], input_is_latent=True)
        th = (time.perf_counter() - t1) / 50
        torch.cuda.synchronize()
        for _ in range(5): generate_image(G, w, 1.0, None, input_is_latent=True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(50): generate_image(G, w, 1.0, None, input_is_latent=True)
        torch.cuda.synchronize(); tg = (time.perf_counter() - t2) / 50
    print('B=%d %s: %.3f ms per forward (%.0f frames/s), host enqueue %.3f ms; generate_image (verified) %.3f ms (%.0f frames/s)'
          % (B, 'graph replay' if graphs else 'eager      ', t * 1e3, B / t, th * 1e3, tg * 1e3, B / tg), flush=True)
