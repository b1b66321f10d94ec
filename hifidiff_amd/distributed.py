"""Multi-GPU plumbing of the sampling path: faces are independent, so ranks take contiguous slices of the
batch (what `accelerator.prepare(dataloader)` does in the reference, test_refiner.py:174) and the only
collective is the final gather of the latents (4 KB per face).  `torch.distributed` with backend "nccl"
is RCCL over xGMI on ROCm; the same code runs on CPU tensors with "gloo" (tests, world_size 2)."""
import os
import sys

import torch
import torch.distributed as dist

_traced = False


def shard_range(n_faces, rank, world):
    """Contiguous slice [lo, hi) of the global batch owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_faces, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t, rank, world):
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def gather_faces(local, n_faces=None):
    """All ranks receive the latents of the whole batch, in global face order.  Without an initialised process group
    (the plain single-process run) this is the identity; with one -- a world of ONE rank included, which is how the
    multi-rank path is rehearsed on a one-GPU box -- the latents go through the backend's all_gather."""
    global _traced
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    if os.environ.get("HD_TRACE_GATHER") and not _traced:
        _traced = True
        sys.stderr.write("gather_faces: all_gather over %s, world %d, %d faces local\n" % (dist.get_backend(), world, local.shape[0]))
    n_faces = n_faces if n_faces is not None else local.shape[0] * world
    sizes = [shard_range(n_faces, r, world) for r in range(world)]
    if len({hi - lo for lo, hi in sizes}) == 1:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(out, local.contiguous())
        return torch.cat(out, dim=0)
    width = max(hi - lo for lo, hi in sizes)                       # ragged: pad to the widest shard
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)
