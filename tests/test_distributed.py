"""CPU, world_size 2 over gloo: the multi-process path of bench.py (flat one-shot state broadcast from rank 0,
contiguous latent sharding, max-over-ranks timing, gradient averaging for A)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stylegan_directions_face_reenactment_amd import distributed as D, synthetic as S
    from stylegan_directions_face_reenactment_amd.model import Generator
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    r, lr, w = D.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different random init per rank
    G = Generator(32, 512, 8, channel_multiplier=1)
    A = DirectionMatrix(512, 15, 512, w_plus=True, num_layers=8, verbose=False)
    trunc = torch.full((1, 512), float(rank))
    if rank == 0:
        G.load_state_dict(S.synthetic_state_dict(G.state_dict(), seed=5))
    nbytes = D.broadcast_state(G, A, [trunc], src=0)
    ref = S.synthetic_state_dict(G.state_dict(), seed=5)
    ok = all(torch.equal(G.state_dict()[k], ref[k]) for k in ref if not k.endswith('.kernel'))
    ok = ok and float(trunc.sum()) == 0.0
    # A must now be rank 0's
    gathered = [torch.zeros_like(A.linear.weight) for _ in range(world)]
    dist.all_gather(gathered, A.linear.weight.detach())
    ok = ok and torch.equal(gathered[0], gathered[1])
    # sharding of a global latent batch
    lo, hi = D.shard_range(10, rank, world)
    total = torch.tensor([hi - lo])
    dist.all_reduce(total)
    ok = ok and int(total) == 10
    # gradient averaging
    A.linear.weight.grad = torch.full_like(A.linear.weight, float(rank + 1))
    A.linear.bias.grad = torch.full_like(A.linear.bias, float(10 * (rank + 1)))
    D.allreduce_grads(A)
    ok = ok and float(A.linear.weight.grad[0, 0]) == 1.5 and float(A.linear.bias.grad[0]) == 15.0
    m = D.max_over_ranks(float(rank), torch.device('cpu'))
    ok = ok and m == float(world - 1)
    D.barrier()
    q.put((rank, bool(ok), nbytes))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_broadcast_and_shard_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][2] > 4 * 20e6     # ~23.6 M floats of Generator(32) state in one flat buffer


def _run_bench(*argv, env=None, timeout=280, retry_abort=True):
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    cmd = [sys.executable, os.path.join(root, 'bench.py')] + list(argv)
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=timeout)
    if r.returncode != 0 and 'Signal 6' in r.stderr and retry_abort:
        # One rank of an 8-process gloo launch on this 8-core container has been seen to die with SIGABRT about once in five
        # suite runs (never reproduced by the same command outside pytest; bench.py now leaves through distributed.shutdown()):
        # keep the evidence, REPORT the retry in the test summary (a warning), try once more, fail if it repeats.
        import warnings
        log = os.path.join(tempfile.gettempdir(), 'sgdfr_bench_abort_%d.log' % os.getpid())
        with open(log, 'a') as f:
            f.write(' '.join(cmd) + '\n' + r.stderr + '\n')
        warnings.warn('bench.py %s died with SIGABRT once and was retried (log: %s)' % (' '.join(argv), log), RuntimeWarning)
        r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=timeout)
    out = r.stdout.splitlines()
    if not out or not out[-1].startswith('{'):
        return r, None
    # the driver's view: the LAST stdout line, alone, compact (bench.LINE_LIMIT); the tables travel on the BENCH_DETAIL stderr line
    assert sum(l.startswith('{') for l in out) == 1, 'bench.py must print exactly one JSON line on stdout'
    assert len(out[-1]) <= 4096, 'the bench line is %d bytes: the driver stopped parsing it at ~21 KB in round 5' % len(out[-1])
    line = json.loads(out[-1])
    detail = [l for l in r.stderr.splitlines() if l.startswith('BENCH_DETAIL ')]
    line['_detail'] = json.loads(detail[-1][len('BENCH_DETAIL '):]) if detail else None
    return r, line


@pytest.mark.timeout(300)
def test_bench_gpus2_spawns_two_ranks_host_check():
    """`python bench.py --gpus 2` (no torchrun environment) starts 2 ranks itself; the host flow -- process group, one flat
    weight broadcast, contiguous shards -- is checked on CPU tensors over gloo."""
    r, line = _run_bench('--gpus', '2', '--host-check', '--size', '32', '--batch', '64')
    assert r.returncode == 0, r.stderr[-2000:]
    full = line['_detail']
    assert line['n_gpus'] == 2 and full['backend'] == 'gloo'
    assert full['shards'] == [[0, 64], [64, 128]]
    assert full['weights_identical_on_all_ranks'] is True
    assert line['config']['weight_broadcast_bytes'] > 4 * 20e6


@pytest.mark.timeout(600)
def test_bench_gpus8_host_check_with_rank_affinity():
    """The 8-rank launch the driver's SCALE step uses, on CPU over gloo: 8 shards of 64, identical weights everywhere, and
    every rank pinned to its own non-empty CPU slice (distributed.bind_rank)."""
    r, line = _run_bench('--gpus', '8', '--host-check', '--size', '32', '--batch', '64', timeout=560)
    assert r.returncode == 0, r.stderr[-2000:]
    full = line['_detail']
    assert line['n_gpus'] == 8 and full['backend'] == 'gloo'
    assert full['shards'] == [[64 * i, 64 * (i + 1)] for i in range(8)]
    assert full['weights_identical_on_all_ranks'] is True
    aff = full['rank_affinity']
    assert len(aff) == 8 and all(a['bound'] and a['n_cpus'] >= 1 for a in aff)
    host = len(os.sched_getaffinity(0))
    # the JSON contract of an N > 1 line: bench.finalize_line has checked this line's keys; here the test pins
    # what the 8-rank line carries -- a roofline object, cpu_baseline null WITH a reason, the last rank's oracle field
    assert line['scaling'] == 'weak' and line['config']['global_batch'] == 8 * 64 and line['vs_baseline'] is None
    # (the compact line the driver parses carries them all; the detail object keeps the long form)
    from bench import CONTRACT_KEYS
    assert set(CONTRACT_KEYS) <= set(line)
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(line['roofline'])
    assert line['cpu_baseline']['value'] is None and 'N=1' in line['cpu_baseline']['reason']
    assert 'last_rank_shard' in full['max_abs_vs_oracle']
    if host >= 8:       # disjoint slices that cover the host's allowed CPUs
        from stylegan_directions_face_reenactment_amd.distributed import parse_cpulist
        sets = [set(parse_cpulist(a['cpus'])) for a in aff]
        assert sum(len(s) for s in sets) == len(set().union(*sets)) == host


def test_affinity_plan_two_sockets():
    from stylegan_directions_face_reenactment_amd.distributed import plan_affinity, parse_cpulist, format_cpulist
    node_cpus = {0: parse_cpulist('0-63,128-191'), 1: parse_cpulist('64-127,192-255')}
    allowed = list(range(256))
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    seen = set()
    for r in range(8):
        node, cpus = plan_affinity(r, 8, numa, node_cpus, allowed)
        assert node == numa[r] and len(cpus) == 32 and set(cpus) <= set(node_cpus[node]) and not (seen & set(cpus))
        seen |= set(cpus)
    assert len(seen) == 256
    # unknown topology: even split of the allowed set; a restricted cpuset is respected
    node, cpus = plan_affinity(1, 2, [None, None], {}, [3, 4, 5, 6])
    assert node is None and cpus == [5, 6]
    node, cpus = plan_affinity(0, 2, [0, 1], node_cpus, list(range(0, 8)))      # rank 1's node has no allowed CPU -> falls back
    assert cpus == list(range(0, 8))
    node, cpus = plan_affinity(1, 2, [0, 1], node_cpus, list(range(0, 8)))
    assert cpus == list(range(0, 8))
    assert format_cpulist(parse_cpulist('0-3,8,10-11')) == '0-3,8,10-11'


@pytest.mark.timeout(120)
def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N on a host with fewer than N devices must fail loudly, never measure fewer GPUs and report N (or 1)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('needs a host with fewer than 2 GPUs')
    r, line = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '1')
    assert r.returncode != 0 and line is None
    assert 'GPU(s) visible' in (r.stderr + r.stdout)
    # and a torchrun-style environment that disagrees with --gpus is refused by the worker itself
    r, line = _run_bench('--gpus', '1', '--host-check', env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0',
                                                             'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port())},
                         timeout=100)
    assert r.returncode != 0 and line is None
