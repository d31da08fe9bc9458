"""Which host-side op issues device-to-device copies inside a no-grad forward?  (4 __amd_rocclr_copyBuffer per
forward.)  python scripts/find_copies.py"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
G = Generator(256, 512, 8, channel_multiplier=1)
G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7))
G = G.eval().cuda()
w = S.synthetic_latents(7, 64, n_latent=14).cuda()
with torch.no_grad():
    for _ in range(3):
        G([w], input_is_latent=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        G([w], input_is_latent=True)
        torch.cuda.synchronize()
for e in prof.events():
    if e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::view', 'aten::slice', 'aten::select', 'aten::as_strided', 'aten::empty_strided', 'aten::_unsafe_view', 'aten::reshape'):
        st = [f for f in (e.stack or []) if 'stylegan_directions' in f][:2]
        print(e.name, [tuple(s) for s in (e.input_shapes or [])][:2] if hasattr(e, 'input_shapes') else '', st)
