"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The path shards trivially: every image depends only on its own latent row and on read-only
generator state, so the batch is split contiguously across ranks and steady-state inference has NO
collective.  The only traffic is one start-up broadcast of the generator state (99 MB at cm=1), `A`
(262 KB) and the truncation latent from rank 0, packed into a single flat buffer so it is one ring
broadcast rather than 135 small ones (xGMI is point-to-point: few large messages beat many small ones).
For direction learning, `allreduce_grads` averages the gradients of `A` (65,536 floats, latency-bound).
Device-agnostic: the same code runs under gloo on CPU tensors (tests/test_distributed.py).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_from_env(backend=None, use_gpu=True):
    """Initialise the default process group from torchrun's env (no-op for world size 1).

    One process per GPU: rank `local_rank` takes device `local_rank` and the call FAILS when that device does not exist --
    ranks are never stacked on one GPU silently (RCCL would refuse duplicate devices anyway).  Only the flow tests may share
    a device: SGDFR_ALLOW_GPU_SHARING=1 together with SGDFR_DIST_BACKEND=gloo."""
    rank, local_rank, world = env_world()
    gpu = use_gpu and torch.cuda.is_available()
    if gpu:
        n_dev = torch.cuda.device_count()
        if local_rank >= n_dev:
            if os.environ.get('SGDFR_ALLOW_GPU_SHARING') != '1':
                raise RuntimeError('rank with LOCAL_RANK=%d but only %d GPU(s) visible: one process per GPU '
                                   '(set SGDFR_ALLOW_GPU_SHARING=1 and SGDFR_DIST_BACKEND=gloo for flow tests)' % (local_rank, n_dev))
            torch.cuda.set_device(local_rank % n_dev)
        else:
            torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('SGDFR_DIST_BACKEND') or ('nccl' if gpu else 'gloo')
        if backend == 'nccl' and gpu:
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


# ---- CPU placement of the ranks.  Every rank issues ~65 launches per forward from Python; on a 2-socket host the ranks must not
# migrate across sockets or pile onto the same cores.  (The reference is single-process: run_inference.py:31, trainer.py:25.)

def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append('%d-%d' % (cpus[i], cpus[j]) if j > i else '%d' % cpus[i])
        i = j + 1
    return ','.join(runs)


def plan_affinity(local_rank, local_world, gpu_numa, node_cpus, allowed):
    """CPUs for `local_rank` (pure function; see bind_rank).  gpu_numa[r]: NUMA node of rank r's GPU or None/-1 when unknown;
    node_cpus: {node: [cpu, ...]}; allowed: the CPUs this process may use.  Ranks whose GPUs hang off the same node share that
    node's allowed CPUs in equal contiguous slices; with unknown topology the allowed CPUs are sliced evenly over all ranks.
    Never returns an empty set (falls back to `allowed`)."""
    allowed = sorted(set(allowed))
    known = all(n is not None and n >= 0 and node_cpus.get(n) for n in gpu_numa) and len(gpu_numa) == local_world
    if known:
        node = gpu_numa[local_rank]
        peers = [r for r in range(local_world) if gpu_numa[r] == node]
        pool = [c for c in sorted(node_cpus[node]) if c in set(allowed)]
        j, k = peers.index(local_rank), len(peers)
    else:
        node, pool, j, k = None, allowed, local_rank, local_world
    lo, hi = shard_range(len(pool), j, k)
    mine = pool[lo:hi]
    if not mine:
        mine = pool or allowed
    return node, mine


def gpu_numa_node(index):
    """NUMA node of HIP device `index` from sysfs (PCI address of the device), or None."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open('/sys/bus/pci/devices/%s/numa_node' % bdf) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def host_numa_cpus():
    """{node: [cpus]} from /sys/devices/system/node (empty when the host does not expose it)."""
    out, base = {}, '/sys/devices/system/node'
    try:
        for name in os.listdir(base):
            if name.startswith('node') and name[4:].isdigit():
                with open(os.path.join(base, name, 'cpulist')) as f:
                    out[int(name[4:])] = parse_cpulist(f.read())
    except OSError:
        pass
    return out


def bind_rank(local_rank, local_world, max_threads=16, use_gpu=True):
    """Pin this process to its slice of the CPUs of its GPU's NUMA node and cap torch's intra-op threads; returns what was done
    (echoed into bench.py's line).  SGDFR_NO_AFFINITY=1 leaves the process alone."""
    if os.environ.get('SGDFR_NO_AFFINITY') == '1' or not hasattr(os, 'sched_setaffinity'):
        return {'bound': False}
    allowed = sorted(os.sched_getaffinity(0))
    gpu = use_gpu and torch.cuda.is_available()
    numa = [gpu_numa_node(r) if gpu and r < torch.cuda.device_count() else None for r in range(local_world)]
    node, cpus = plan_affinity(local_rank, local_world, numa, host_numa_cpus(), allowed)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return {'bound': False}
    threads = max(1, min(len(cpus), max_threads))
    torch.set_num_threads(threads)
    return {'bound': True, 'numa_node': node, 'cpus': format_cpulist(cpus), 'n_cpus': len(cpus), 'torch_threads': threads}


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (small python objects)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def shard_range(total, rank, world):
    """Contiguous [start, stop) of `total` items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _tensors_of(obj):
    """Ordered (name, tensor) list of a module's state / a dict / a list of tensors."""
    if isinstance(obj, torch.nn.Module):
        return [(k, v) for k, v in obj.state_dict(keep_vars=True).items()]
    if isinstance(obj, dict):
        return list(obj.items())
    return [(str(i), v) for i, v in enumerate(obj)]


def broadcast_state(*objs, src=0, group=None, force=False):
    """Broadcast every tensor of the given modules / dicts / tensor lists from `src` in ONE collective.
    All ranks must pass identically-shaped objects (ranks != src may hold uninitialised values).
    force=True issues the collective even in a one-rank group (exercises the RCCL path on a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return 0
    items = [t for o in objs for _, t in _tensors_of(o)]
    if not items:
        return 0
    device = items[0].device
    total = sum(t.numel() for t in items)
    if dist.get_rank(group) == src:      # one gather launch (torch.cat) instead of one copy per tensor
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in items])
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        with torch.no_grad():
            views = [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in items]), items)]
            same = [(t, v) for t, v in zip(items, views) if t.dtype == torch.float32]
            if same:                     # one multi-tensor scatter for the fp32 state, per-tensor only for odd dtypes
                torch._foreach_copy_([t.detach() for t, _ in same], [v for _, v in same])
            for t, v in zip(items, views):
                if t.dtype != torch.float32:
                    t.copy_(v)
    return total * 4


def allreduce_grads(module, group=None):
    """Average the gradients of `module` over ranks with one flat all-reduce (training of `A`)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
