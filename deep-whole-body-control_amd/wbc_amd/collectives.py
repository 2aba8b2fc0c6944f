"""The learner's three collectives behind one seam: sum all-reduce, max all-reduce, broadcast.

Production: the process group is RCCL (`backend="nccl"` on ROCm), device tensors go straight to `torch.distributed` and
travel over xGMI. Test / single-device rigs: a `gloo` group whose ranks may share ONE device (two processes on cuda:0); a
device tensor is staged through host memory around the gloo call, so the very same learner code (fused minibatch kernels ->
flat gradient bucket -> all-reduce -> fused clip + Adam) runs with world_size > 1 on a box with a single GPU. The reference
has no counterpart (its only multi-GPU artefact is the dead `--horovod` flag, legged_gym/utils/helpers.py:164)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _needs_staging(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t: torch.Tensor, group, op=None) -> None:
    """In place; SUM unless `op` says otherwise. Ordered on torch's current stream for device tensors."""
    op = dist.ReduceOp.SUM if op is None else op
    if _needs_staging(t, group):
        host = t.detach().cpu()               # synchronises with the stream that produced t
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


def broadcast(t: torch.Tensor, src_group_rank: int, group) -> None:
    src = dist.get_global_rank(group, src_group_rank)
    if _needs_staging(t, group):
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)
