"""Times the generator fine-tuning step shape of libs/optimization.py:47-68 (PTI: one source image, Adam over convs[4..11],
every generator parameter left with requires_grad=True as the reference does) on one GPU with synthetic weights; the loss
is an L2 stand-in for LPIPS + L2 (out-of-scope neighbours).  Numbers go to DESIGN.md."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    only_needed = 'needed' in sys.argv[2:]       # freeze what the optimizer does not touch
    G = Generator(256, 512, 8, channel_multiplier=1)
    G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7))
    G = G.train().cuda()
    params = [p for i in range(4, 12) for p in G.convs[i].parameters()]
    if only_needed:
        ids = {id(p) for p in params}
        for p in G.parameters():
            p.requires_grad_(id(p) in ids)
    opt = torch.optim.Adam(params, lr=3e-3)
    trunc = S.counter_tensor(7, 'trunc', (1, 512)).cuda()
    latent = S.synthetic_latents(7, B, n_latent=G.n_latent, key='pti.w').cuda()
    target = torch.tanh(S.counter_tensor(7, 'pti.t', (B, 3, 256, 256))).cuda()
    import warnings; warnings.simplefilter('ignore')
    def step():
        img, _ = G([latent], input_is_latent=True, return_latents=False, truncation=0.7, truncation_latent=trunc)
        loss = ((img - target) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    graph = 'graph' in sys.argv[2:]
    if graph:        # the whole step (forward, backward, Adam) as one hipGraph replay: nothing left of the ~560 host launches
        from stylegan_directions_face_reenactment_amd.finetune import FusedAdam
        opt = FusedAdam(params, lr=3e-3) if 'fused' in sys.argv[2:] else torch.optim.Adam(params, lr=3e-3, capturable=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        for p in G.parameters():
            p.grad = None
        with torch.cuda.graph(cg):
            static_loss = step()
        eager_step = step
        def step():
            cg.replay()
            return static_loss
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        l = step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
    print('PTI-shaped step B=%d (%s): %.2f ms/step, 200 steps = %.2f s; loss %.4f' %
          (B, 'grads of the optimised layers only' if only_needed else 'all generator gradients', el * 1e3, el * 200, float(l)))

if __name__ == '__main__':
    main()
