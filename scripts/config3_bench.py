"""BASELINE.json configs[2]: the run_inference.py path on one MI355X -- e4e W+ of the source (once), then per batch of
32 target frames DirectionMatrix shift + truncation (psi=0.7) + HIP generator + uint8 frames.  Also times the e4e
encoder alone at B=32 (the invert_images.py loop).  Synthetic weights / inputs.   python scripts/config3_bench.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession, grid_frames_uint8


def timed(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    B, steps, warmup = int(os.environ.get('B', 32)), int(os.environ.get('STEPS', 20)), 5
    dev = torch.device('cuda:0')
    G = Generator(256, 512, 8, channel_multiplier=1)
    G.load_state_dict(S.synthetic_state_dict({k: v for k, v in G.state_dict().items()}, seed=1))
    G = G.eval().to(dev)
    enc = Encoder4Editing(50, 'ir_se', 256).eval()
    enc.load_state_dict(S.synthetic_encoder_state(enc.state_dict(), seed=2))
    enc = enc.to(dev)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    A.load_state_dict(S.synthetic_direction_state(3)); A = A.to(dev)
    with torch.no_grad():
        trunc = G.mean_latent(4096)
    src_img = S.counter_tensor(4, 'c3.src', (1, 3, 256, 256), 0.0, 0.5).clamp_(-1, 1).to(dev)
    tgt = S.counter_tensor(4, 'c3.tgt', (B, 3, 256, 256), 0.0, 0.5).clamp_(-1, 1).to(dev)
    sv = S.counter_tensor(4, 'c3.sv', (B, 15), 0.0, 2.0).to(dev)
    with torch.no_grad():
        t_src = timed(lambda: enc(src_img), 5, 3)
        t_enc = timed(lambda: enc(tgt), max(3, steps // 4), 2)
        sess = ReenactmentSession(G, A, enc(src_img), 0.7, trunc, batch=B)
        t_gen = timed(lambda: sess.render(sv), steps, warmup)
        t_all = timed(lambda: sess.video_frames(src_img, tgt, sv), steps, warmup)
    print(json.dumps({'config': 'run_inference path, B=%d, 256x256, cm=1, psi=0.7' % B,
                      'e4e_source_ms': round(t_src * 1e3, 2), 'e4e_batch_images_per_s': round(B / t_enc, 1),
                      'reenact_frames_per_s': round(B / t_gen, 1),
                      'reenact_plus_video_grid_frames_per_s': round(B / t_all, 1)}))


main()
