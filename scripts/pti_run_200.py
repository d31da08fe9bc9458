"""finetune.optimize_g as run_inference.py drives libs/optimization.py (200 steps, one source, lr 3e-3) on synthetic weights: losses of
the graph-replayed run with the one-launch Adam against the eager run with torch.optim.Adam, wall time, clamped fp16 pairs.
python scripts/pti_run_200.py"""
import copy, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import finetune, synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
import warnings; warnings.simplefilter('ignore')
G0 = Generator(256, 512, 8, channel_multiplier=1)
G0.load_state_dict(S.synthetic_state_dict(G0.state_dict(), seed=7))
G0 = G0.cuda()
w = S.synthetic_latents(7, 1, n_latent=G0.n_latent, key='pti.w').cuda()
trunc = S.counter_tensor(7, 'pti.t', (1, 512)).cuda()
with torch.no_grad():
    base, _ = G0([w], input_is_latent=True, truncation=0.7, truncation_latent=trunc)
target = (base + 0.3 * S.counter_tensor(7, 'pti.d', tuple(base.shape)).cuda()).clamp(-1, 1)
first = finetune.l2_loss_fn(base, target, 100).item()
for name, kw in (('hipGraph replay + FusedAdam (default)', {}), ('eager + torch.optim.Adam', {'graph': False, 'fused_adam': False})):
    G = copy.deepcopy(G0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G, loss = finetune.optimize_g(G, w, target, trunc, opt_steps=200, lr=3e-3, **kw)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('%-40s 200 steps in %.2f s (%.2f ms/step incl. warm-up and capture): loss %.4f -> %.4f, fp16 pairs clamped %d'
          % (name, el, el / 200 * 1e3, first, loss.item(), G.saturated_pairs()))
