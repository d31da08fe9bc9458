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


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:   # SGDFR_DIST_BACKEND=gloo lets several ranks share one GPU (smoke tests of the N>1 flow)
            backend = os.environ.get('SGDFR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


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


def broadcast_state(*objs, src=0, group=None):
    """Broadcast every tensor of the given modules / dicts / tensor lists from `src` in ONE collective.
    All ranks must pass identically-shaped objects (ranks != src may hold uninitialised values)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    items = [t for o in objs for _, t in _tensors_of(o)]
    if not items:
        return 0
    device = items[0].device
    total = sum(t.numel() for t in items)
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank(group) == src:
        off = 0
        for t in items:
            flat[off:off + t.numel()].copy_(t.detach().reshape(-1))
            off += t.numel()
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        off = 0
        with torch.no_grad():
            for t in items:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
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
