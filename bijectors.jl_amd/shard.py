"""Batch sharding across GPUs: one process per GPU, every rank owns a contiguous
block of columns, parameters are replicated, and the ONLY collective on the path is one sum
all-reduce of the float64 partial Σ logabsdetjac (8 bytes, latency-bound) over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_columns(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous column block [lo, hi) of `rank`: GPU g gets columns [g*N/G, (g+1)*N/G)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} for world size {world_size}")
    lo = (batch * rank) // world_size
    hi = (batch * (rank + 1)) // world_size
    return lo, hi


_LIBRARY_COLLECTIVE = [False]


def use_library_collective(on: bool = True) -> None:
    """Route the Σ logabsdetjac all-reduce through the library's own RCCL communicator (bjx_allreduce_sum_f64 on the
    context's stream, after `init_comm`) instead of torch.distributed — the path a Julia host takes."""
    _LIBRARY_COLLECTIVE[0] = bool(on)


def allreduce_logabsdetjac(partial: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum all-reduce of the per-rank float64 partial log-det sum."""
    import torch.distributed as dist

    if partial.dtype != torch.float64:
        raise TypeError("the partial log-det sum is reduced in float64 so the result does not depend on the shard count")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _LIBRARY_COLLECTIVE[0] and partial.is_cuda:
            from . import _lib as L
            from . import interface as I

            ctx = I.context(partial.device)
            L.check(ctx.h, L.load().bjx_allreduce_sum_f64(ctx.h, partial.data_ptr(), partial.numel()), "bjx_allreduce_sum_f64")
        else:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def with_logabsdet_jacobian_sharded(b, x_shard: torch.Tensor, group=None, out: Optional[torch.Tensor] = None,
                                    per_sample: bool = True):
    """Hot path on this rank's column block + the single collective.

    Returns (y_shard, ladj_per_sample_shard or None, ladj_sum_global[float64, 1 element]); outputs
    stay sharded, only the scalar is global.  per_sample=False computes just the scalar (the shape
    the reference returns for elementwise chains)."""
    from . import interface as I

    mode = "both" if per_sample else "sum64"
    ops = I._fused_ops(b)
    if ops is not None:
        y, l = I._run_chain(ops, x_shard, mode, True, out_y=out)
    else:
        with I._into(out):
            y, l = b._wlj(x_shard, per_sample=mode)
        if out is not None and y.data_ptr() != out.data_ptr():
            out.copy_(y)
            y = out
    lps, lsum = l if per_sample else (None, l)
    allreduce_logabsdetjac(lsum, group)
    return y, lps, lsum


def init_comm(device: Optional[torch.device] = None, group=None) -> None:
    """Give this rank's context an RCCL communicator (bjx_comm_init) so that entry points with an
    in-library collective — InvertibleBatchNorm in training mode: one all-reduce of the 2·dim+1 Float64
    batch sums, SURVEY.md §8(e) — see the GLOBAL batch.  The 128-byte ncclUniqueId is made on rank 0 and
    broadcast through torch.distributed (any backend).  No-op for a single process."""
    import ctypes as C

    import torch.distributed as dist

    from . import _lib as L
    from . import interface as I

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    ctx = I.context(device)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        L.check(ctx.h, L.load().bjx_comm_unique_id(buf), "bjx_comm_unique_id")
    box = [bytes(buf)]
    dist.broadcast_object_list(box, src=0, group=group)
    ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
    L.check(ctx.h, L.load().bjx_comm_init(ctx.h, world, rank, ident), "bjx_comm_init")
