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
# the other collectives of bench.py's N > 1 flow, on RCCL: MAX all-reduce of a float64 (max_over_ranks), all_gather of float64
# tensors (rank_spread), all_gather_object (gather_objects: rank affinity, the last rank's oracle check)
t64 = torch.tensor([3.25], dtype=torch.float64, device='cuda')
dist.all_reduce(t64, op=dist.ReduceOp.MAX)
assert float(t64) == 3.25
got = [torch.zeros_like(t64)]
dist.all_gather(got, t64)
assert float(got[0]) == 3.25
objs = [None]
dist.all_gather_object(objs, {'rank': 0, 'within_bar': True, 'fp16x3': 1.5e-5})
assert objs[0]['within_bar'] is True and objs[0]['fp16x3'] == 1.5e-5
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
    # the oracle check also ran on a shard that is not rank 0's (the last rank's first rows, its own range plan)
    # (the compact line carries both figures as scalars, the BENCH_DETAIL record the whole check)
    assert line['max_abs_vs_oracle'] <= 1e-3 and line['max_abs_vs_oracle_last_rank'] <= 1e-3
    full = line['_detail']
    far = full['max_abs_vs_oracle']['last_rank_shard']
    assert far['within_bar'] and far['fp16x3'] <= 1e-3 and full['max_abs_vs_oracle']['within_bar']
    assert abs(far['fp16x3'] - line['max_abs_vs_oracle_last_rank']) <= 1e-4 * far['fp16x3']     # (the compact line keeps 5 significant digits)
    assert line['config']['weight_broadcast_bytes'] > 4 * 20e6
    lo, hi = line['config']['per_rank_frames_per_s_min_max']
    assert 0 < lo <= hi


_SHARD_RENDER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from stylegan_directions_face_reenactment_amd import distributed as D, synthetic as S
from stylegan_directions_face_reenactment_amd.model import Generator
rank, local_rank, world = D.init_from_env()
dev = torch.device('cuda', torch.cuda.current_device())
G = Generator(64, 512, 8, channel_multiplier=1)
if rank == 0:
    G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=11))
G = G.eval().to(dev)
D.broadcast_state(G, src=0)
n = int(sys.argv[1])
lo, hi = D.shard_range(n, rank, world)
w = S.synthetic_latents(11, n, n_latent=G.n_latent, key='shard.w')[lo:hi].contiguous().to(dev)
with torch.no_grad():
    img, _ = G([w], input_is_latent=True)
torch.save({'lo': lo, 'hi': hi, 'img': img.cpu(), 'mode': G.range_mode(), 'sat': G.saturated_pairs()}, sys.argv[2] + '.%%d' %% rank)
D.barrier()
D.shutdown()
'''


@pytest.mark.timeout(900)
def test_sharded_rows_equal_the_single_process_rendering(tmp_path):
    """SURVEY 8e / VERDICT r3 #6: global row i rendered by rank r of a 2-rank run (contiguous shards, weights from rank 0's flat
    broadcast, every rank calibrating its OWN fp16 range plan from its own shard) equals the single-process rendering of the same
    latent within 1e-5, and both are within the bar of the oracle.  Ranks share this box's GPU over gloo; over RCCL when 2 devices
    are visible."""
    from util import O, maxabs
    from stylegan_directions_face_reenactment_amd import synthetic as S
    from stylegan_directions_face_reenactment_amd.model import Generator
    two = torch.cuda.device_count() >= 2
    n = 13                                           # uneven shards: 7 + 6
    out = str(tmp_path / 'shard')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if not two:
        env.update(SGDFR_ALLOW_GPU_SHARING='1', SGDFR_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    script = tmp_path / 'shard_render.py'
    script.write_text(_SHARD_RENDER % ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script), str(n), out]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=840)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    G = Generator(64, 512, 8, channel_multiplier=1)
    state = S.synthetic_state_dict(G.state_dict(), seed=11)
    G.load_state_dict(state)
    G = G.eval().cuda()
    w = S.synthetic_latents(11, n, n_latent=G.n_latent, key='shard.w')
    with torch.no_grad():
        whole, _ = G([w.cuda()], input_is_latent=True)
        ref, _ = O.generator_forward(state, [w], input_is_latent=True)
    seen = 0
    for rank in range(2):
        part = torch.load(out + '.%d' % rank)
        lo, hi = part['lo'], part['hi']
        assert part['mode'] == G.range_mode() and part['sat'] == 0
        d_whole, d_ref = maxabs(part['img'], whole[lo:hi]), maxabs(part['img'], ref[lo:hi])
        print('rank %d rows %d..%d: vs single process %.2e, vs oracle %.2e' % (rank, lo, hi, d_whole, d_ref))
        assert d_whole <= 1e-5 and d_ref <= 2e-4
        seen += hi - lo
    assert seen == n


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
    assert line['_detail']['roofline']['per_layer'] and line['config']['e4e_source_ms'] > 0
    r, line = _run_bench('--config', 'trainer', '--steps', '2', '--warmup', '1', '--batch', '4', timeout=700)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['unit'] == 'samples/s' and line['value'] > 0
    compact, line = line, line['_detail']          # (the long-form record; the compact line keeps value / roofline / the leg scalars)
    assert compact['roofline']['frac'] > 0 and compact['config']['generator_only_ms_per_step'] > 0
    assert line['config']['losses_finite'] is True
    # the arithmetic that actually ran, not a substring of the prose: forward = the bench default, backward = functional.BACKWARD_ARITH
    from stylegan_directions_face_reenactment_amd import functional as F_
    assert line['config']['forward_arithmetic'] == 'fp16x3' and line['config']['backward_arithmetic'] == F_.BACKWARD_ARITH
    assert line['roofline']['frac'] > 0 and line['config']['generator_only_ms_per_step'] > 0
    first, last = line['config']['loss_first_last']
    assert first > 0 and last > 0
