"""Where does the host time of one eager no-grad forward go (B=1: the forward is host-bound)?  python scripts/host_profile.py"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
G = Generator(256, 512, 8, channel_multiplier=1)
G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7))
G = G.eval().cuda()
w = S.synthetic_latents(7, 1, n_latent=14).cuda()
with torch.no_grad():
    for _ in range(10): G([w], input_is_latent=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): G([w], input_is_latent=True)
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
