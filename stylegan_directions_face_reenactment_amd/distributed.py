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
