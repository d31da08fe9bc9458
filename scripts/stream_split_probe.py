"""Does splitting one batch over several HIP streams fill the wave-quantisation tails?  python scripts/stream_split_probe.py"""
import os
os.environ.setdefault('SGDFR_VERIFY_RANGE', '0')      # timing script: raw forwards return at once (the product default verifies)
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd.model import Generator
from stylegan_directions_face_reenactment_amd import synthetic

def main():
    B = int(os.environ.get('B', 64))
    G = Generator(256, 512, 8, channel_multiplier=1).cuda().eval()
    G.load_state_dict(synthetic.synthetic_state_dict({k: v.cpu() for k, v in G.state_dict().items()}, seed=1), strict=False)
    w = synthetic.synthetic_latents(1, B, G.n_latent, 512).cuda()
    for ns in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        chunks = w.chunk(ns)
        def step():
            cur = torch.cuda.current_stream()
            outs = []
            for s, c in zip(streams, chunks):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(G([c], input_is_latent=True, randomize_noise=False)[0])
            for s in streams: cur.wait_stream(s)
            return outs
        with torch.no_grad():
            for _ in range(3): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        print('streams %d: %.2f ms/step %.0f frames/s' % (ns, t * 1e3, B / t), flush=True)
main()
