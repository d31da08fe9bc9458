import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = Generator(256, 512, 8, channel_multiplier=1)
G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7)); G = G.eval().cuda()
w = S.synthetic_latents(7, B).cuda()
with torch.no_grad():
    for _ in range(3): ref, _ = G([w], input_is_latent=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): G([w], input_is_latent=True)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    static_w = w.clone()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): G([static_w], input_is_latent=True)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out, _ = G([static_w], input_is_latent=True)
    g.replay(); torch.cuda.synchronize()
    print('max diff vs eager', float((out - ref).abs().max()))
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 50
print('B=%d eager %.3f ms  graph %.3f ms' % (B, eager * 1e3, graphed * 1e3))
