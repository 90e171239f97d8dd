"""Batch sharding across GPUs: one process per GPU, every rank owns a contiguous
block of columns, parameters are replicated, and the ONLY collective on the forward path is one sum
all-reduce of the float64 partial Σ logabsdetjac (8 bytes, latency-bound) over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm; "gloo" in the CPU tests).  Training adds the sum of the
parameter cotangents over the ranks as ONE float64 bucket (`allreduce_param_cotangents`)."""
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


def _is_world(group) -> bool:
    """True when `group` is the default (world) group — the set of ranks the library's own communicator (init_comm) spans."""
    import torch.distributed as dist

    return group is None or group is dist.group.WORLD


def allreduce_logabsdetjac(partial: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum all-reduce of the per-rank float64 partial log-det sum."""
    import torch.distributed as dist

    if partial.dtype != torch.float64:
        raise TypeError("the partial log-det sum is reduced in float64 so the result does not depend on the shard count")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _LIBRARY_COLLECTIVE[0] and partial.is_cuda and _is_world(group):      # the library communicator spans the world:
            from . import _lib as L                                              # a sub-group goes through torch.distributed
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


def _leaves(tree, out):
    if isinstance(tree, torch.Tensor):
        out.append(tree)
    elif isinstance(tree, dict):
        for k in sorted(tree):
            _leaves(tree[k], out)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            _leaves(v, out)
    elif tree is not None:
        raise TypeError(f"parameter cotangents must be tensors, dicts, lists or None (got {type(tree).__name__})")
    return out


def allreduce_param_cotangents(grads, group=None):
    """Data-parallel training: every rank holds the parameter cotangents of ITS column block (`vjp_params` sums over the local
    batch); the parameters are replicated, so the cotangents are summed over the ranks — the one exchange step of a training
    step.  All leaves of `grads` (the dictionaries of `vjp_params`, nested `{"stages": [...]}` included) travel as ONE float64
    bucket (a flow's parameters are a few KiB: one latency-bound collective instead of one per tensor; float64 so the sum does
    not depend on the shard count beyond the ranks' own rounding) and are written back in place.  Returns `grads`."""
    import torch.distributed as dist

    leaves = _leaves(grads, [])
    if not leaves or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    bucket = torch.cat([t.detach().reshape(-1).to(torch.float64) for t in leaves])
    if _LIBRARY_COLLECTIVE[0] and bucket.is_cuda and _is_world(group):
        from . import _lib as L
        from . import interface as I

        ctx = I.context(bucket.device)
        L.check(ctx.h, L.load().bjx_allreduce_sum_f64(ctx.h, bucket.data_ptr(), bucket.numel()), "bjx_allreduce_sum_f64")
    else:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in leaves:
        n = t.numel()
        t.copy_(bucket[off:off + n].reshape(t.shape).to(t.dtype))
        off += n
    return grads


def vjp_params_sharded(b, x_shard: torch.Tensor, out_bar_shard: torch.Tensor, ladj_bar_shard=None, group=None):
    """`vjp_params` on this rank's column block + the all-reduce of the parameter cotangents: (x_bar_shard, global cotangents).
    The input cotangent stays sharded like the data."""
    from . import interface as I

    # a sharded training step must not normalise with per-rank batch statistics by accident: every InvertibleBatchNorm of the chain
    # has to say how its batch is sharded (sync=True / a group) or that per-rank statistics are intended (sync=False)
    if I.istraining():
        for st in (b._stages() if isinstance(b, I.ComposedFunction) else [b]):
            base = st.orig if isinstance(st, I.Inverse) else st
            if isinstance(base, I.InvertibleBatchNorm) and base.sync is None:
                raise ValueError("vjp_params_sharded: InvertibleBatchNorm(sync=None) in training mode — pass sync=True (statistics of the "
                                 "whole sharded batch) or sync=False (per-rank statistics on purpose)")
    x_bar, grads = I.vjp_params(b, x_shard, out_bar_shard, ladj_bar_shard)
    return x_bar, allreduce_param_cotangents(grads, group)


def init_comm(device: Optional[torch.device] = None, group=None, timeout_ms: int = 0) -> None:
    """Give this rank's context an RCCL communicator (bjx_comm_init): `use_library_collective()` then routes the Σ logabsdetjac
    all-reduce (and the parameter-cotangent bucket) through bjx_allreduce_sum_f64 on the context's stream, and the C entry
    bjx_batchnorm_train — what a Julia host calls — sees the GLOBAL batch (one all-reduce of the 2·dim+1 Float64 sums,
    SURVEY.md §8e).  The Python `InvertibleBatchNorm` does NOT go through that entry: it all-reduces its statistics itself and
    only when constructed with `sync=True` / a group (sync=None warns once in a multi-process job, `vjp_params_sharded` refuses
    it).  The 128-byte ncclUniqueId is made on rank 0 and broadcast through torch.distributed (any backend).  No-op for a single process."""
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
    if timeout_ms > 0:        # watchdog of bjx_synchronize: a collective some rank never joins becomes an error, not a hang
        L.check(ctx.h, L.load().bjx_set_option(ctx.h, L.BJX_OPT_COLLECTIVE_TIMEOUT_MS, int(timeout_ms)), "bjx_set_option")
