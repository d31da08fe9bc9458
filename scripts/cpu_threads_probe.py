import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from oracle import sg2_oracle as O
from stylegan_directions_face_reenactment_amd import synthetic as S
P = S.synthetic_state_dict(O.template_state(256, 512, 8, 1), seed=7)
w = S.synthetic_latents(7, 2, key='cpu.w')
print('cpu_count', os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    with torch.no_grad():
        O.generator_forward(P, [w], input_is_latent=True)
        t0 = time.perf_counter(); O.generator_forward(P, [w], input_is_latent=True); el = time.perf_counter() - t0
    print('threads', th, 'B=2 forward %.3f s -> %.2f img/s' % (el, 2 / el), flush=True)
