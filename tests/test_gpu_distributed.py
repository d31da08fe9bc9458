"""GPU: the multi-rank flow of bench.py on real devices.
  * RCCL itself (backend "nccl") is exercised on the one GPU of the test box with a one-rank group (communicator init, the
    flat state broadcast, all-reduce, barrier), and with 2 ranks when 2 devices are visible;
  * the 2-rank bench flow (launcher -> process group -> broadcast -> shards -> timed region -> one JSON line with n_gpus 2)
    runs on one GPU with the ranks sharing the device over gloo (test-only switch), and over RCCL when 2 devices exist;
  * the inference / trainer configs of bench.py produce their lines (small batches)."""
import os
import subprocess
import sys

import pytest
import torch

from test_distributed import _free_port, _run_bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RCCL_ONE_RANK = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from stylegan_directions_face_reenactment_amd import distributed as D, synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
G = Generator(64, 512, 8, channel_multiplier=1)
ref = S.synthetic_state_dict(G.state_dict(), seed=5)
G.load_state_dict(ref)
G = G.cuda()
n = D.broadcast_state(G, src=0, force=True)                 # ncclBroadcast of the flat 95 MB state on this GPU
assert n == 4 * sum(v.numel() for v in G.state_dict().values()), n
assert all(torch.equal(G.state_dict()[k].cpu(), ref[k]) for k in ref)
t = torch.arange(1024, device='cuda', dtype=torch.float32)
dist.all_reduce(t)
assert torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))
dist.barrier()
torch.cuda.synchronize()
print('rccl-ok backend=%%s bytes=%%d' %% (dist.get_backend(), n))
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_rccl_one_rank_group_on_this_gpu():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_ONE_RANK % ROOT], capture_output=True, text=True, env=env, timeout=560)
    assert r.returncode == 0 and 'rccl-ok backend=nccl' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_this_gpu_over_gloo():
    r, line = _run_bench('--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--size', '64', '--no-alt',
                         '--no-cpu-baseline', env={'SGDFR_ALLOW_GPU_SHARING': '1', 'SGDFR_DIST_BACKEND': 'gloo'}, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 8 and line['value'] > 0
    assert line['config']['weight_broadcast_bytes'] > 4 * 20e6
    lo, hi = line['config']['per_rank_frames_per_s_min_max']
    assert 0 < lo <= hi


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 visible GPUs (the 1-GPU test box covers RCCL with a one-rank group and the 2-rank flow over gloo)')
    r, line = _run_bench('--gpus', '2', '--steps', '3', '--warmup', '2', '--no-alt', '--no-cpu-baseline', timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 128
    assert abs(line['config']['weight_broadcast_bytes'] - 99.07e6) < 0.5e6


@pytest.mark.timeout(900)
def test_bench_refuses_two_ranks_on_one_gpu():
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a 1-GPU box')
    r, line = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '1', timeout=300)
    assert r.returncode != 0 and line is None and 'GPU(s) visible' in (r.stderr + r.stdout)


@pytest.mark.timeout(1500)
def test_bench_inference_and_trainer_configs_small():
    r, line = _run_bench('--config', 'inference', '--steps', '2', '--warmup', '1', '--batch', '4', timeout=700)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['n_gpus'] == 1 and line['unit'] == 'frames/s' and line['value'] > 0
    assert line['roofline']['per_layer'] and line['config']['e4e_source_ms'] > 0
    r, line = _run_bench('--config', 'trainer', '--steps', '2', '--warmup', '1', '--batch', '4', timeout=700)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['unit'] == 'samples/s' and line['value'] > 0 and line['config']['losses_finite'] is True
    # the arithmetic that actually ran, not a substring of the prose: forward = the bench default, backward = functional.BACKWARD_ARITH
    from stylegan_directions_face_reenactment_amd import functional as F_
    assert line['config']['forward_arithmetic'] == 'fp16x3' and line['config']['backward_arithmetic'] == F_.BACKWARD_ARITH
    assert line['roofline']['frac'] > 0 and line['config']['generator_only_ms_per_step'] > 0
    first, last = line['config']['loss_first_last']
    assert first > 0 and last > 0
