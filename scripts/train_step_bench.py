"""Times the direction-learning step shape of libs/trainer.py:155-189 on one GPU (synthetic weights):
two no-grad forwards + one grad forward + backward to A, loss = mean(img^2) stand-in (the real losses are
out-of-scope neighbours).  Not the headline bench; numbers go to DESIGN.md."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
from stylegan_directions_face_reenactment_amd.generic import generate_image

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    freeze = 'full' not in sys.argv[2:]
    G = Generator(256, 512, 8, channel_multiplier=1)
    G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=7))
    G = G.eval().cuda()
    if freeze:
        for p in G.parameters():
            p.requires_grad_(False)
    A = DirectionMatrix(512, 15, 512, w_plus=True, num_layers=8, verbose=False).cuda()
    opt = torch.optim.Adam(A.parameters(), lr=1e-4)
    trunc = S.counter_tensor(7, 'trunc', (1, 512)).cuda()
    z_s, z_t = S.synthetic_z(7, B, key='zs').cuda(), S.synthetic_z(7, B, key='zt').cuda()
    sv = S.counter_tensor(7, 'sv', (B, 15), 0.0, 3.0).cuda()
    z_st = torch.cat([z_s, z_t])
    two_calls = 'two_calls' in sys.argv[2:]
    G.fused_backward = 'per_layer' not in sys.argv[2:]        # (the whole-synthesis Function is the default)
    import warnings; warnings.simplefilter('ignore')
    def step():
        with torch.no_grad():
            if two_calls:
                generate_image(G, z_s, 0.7, trunc, input_is_latent=False, return_latents=True)
                generate_image(G, z_t, 0.7, trunc, input_is_latent=False, return_latents=True)
            else:       # source and target rows in one forward (what bench.py --config trainer does)
                generate_image(G, z_st, 0.7, trunc, input_is_latent=False, return_latents=True)
        opt.zero_grad()
        img, lat = generate_image(G, z_s, 0.7, trunc, shift_code=A(sv), input_is_latent=False, return_latents=True)
        (img ** 2).mean().backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
    print('train-shaped step B=%d freeze_G=%s: %.2f ms/step -> %.1f samples/s' % (B, freeze, el * 1e3, B / el))

if __name__ == '__main__':
    main()
