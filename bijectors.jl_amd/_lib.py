"""ctypes binding of libbjx_hip.so (include/bjx.h).  No torch types cross this boundary.

The library is the product: if it is missing or fails to load this module raises — there is
no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BJX_LIB_PATH") or os.path.join(_HERE, "libbjx_hip.so")   # override: A/B of two builds in one GPU call

BJX_F32, BJX_F64 = 0, 1
BJX_ACCUMULATE = 1 << 0
BJX_REF_VECTOR_SCALE_LADJ = 1 << 1
BJX_BASE_STDNORMAL = 1 << 2
BJX_INPUT_STDNORMAL = 1 << 3
BJX_COUPLING_SCALE_BCAST = 1 << 4
BJX_COUPLING_SHIFT_BCAST = 1 << 5
BJX_MAX_OPS = 8

(OP_EXP, OP_LOG, OP_SHIFT, OP_SCALE, OP_SCALE_INV, OP_LOGIT, OP_LOGIT_INV, OP_LEAKY_RELU,
 OP_TRUNCATED, OP_TRUNCATED_INV, OP_SIGNFLIP, OP_IDENTITY, OP_STDNORMAL_LOGPDF) = range(1, 14)

ERR_ARG, ERR_SHAPE, ERR_UNSUPPORTED, ERR_NOCOMM, ERR_FINALIZE = -1, -2, -3, -4, -5


class BjxOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("param_len", C.c_int32),
        ("p0", C.c_double),
        ("p1", C.c_double),
        ("v0", C.c_void_p),
        ("v1", C.c_void_p),
    ]


BJX_MAX_SEG_OPS = 4


class BjxSegment(C.Structure):
    _fields_ = [("in_lo", C.c_int64), ("out_lo", C.c_int64), ("len", C.c_int64), ("n_ops", C.c_int32), ("reserved", C.c_int32),
                ("ops", BjxOp * BJX_MAX_SEG_OPS)]


class BjxBlock(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("in_lo", C.c_int64), ("out_lo", C.c_int64), ("len_in", C.c_int64), ("len_out", C.c_int64)]


BLOCK_SIMPLEX, BLOCK_SIMPLEX_INV, BLOCK_ORDERED, BLOCK_ORDERED_INV = 1, 2, 3, 4

_vp, _i, _i64, _u32, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
_tail = [_vp, _vp, _i64, _i64, _u32]  # ladj_ps, ladj_sum, dim/K, batch, flags

BJX_OPT_INKERNEL_FINALIZE = 1
BJX_OPT_COLLECTIVE_TIMEOUT_MS = 2
BJX_OPT_PARAM_EPOCH = 3
BJX_OPT_DEBUG_FIN_DROP_BLOCK = 4      # fault injection, tests only
BJX_OPT_DEBUG_FIN_POISON_SLOT = 5
BJX_PLAN_CHAIN, BJX_PLAN_SIMPLEX, BJX_PLAN_ORDERED = 1, 2, 3

# name -> (restype, argtypes); mirrors include/bjx.h line by line
SIGNATURES = {
    "bjx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "bjx_destroy": (_i, [_vp]),
    "bjx_set_stream": (_i, [_vp, _vp]),
    "bjx_last_error": (C.c_char_p, [_vp]),
    "bjx_version": (_i, []),
    "bjx_workspace_bytes": (C.c_size_t, [_vp]),
    "bjx_synchronize": (_i, [_vp]),
    "bjx_check_state": (_i, [_vp]),
    "bjx_ordered_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_simplex_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_vec_cholesky_inv_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_vec_cholesky_fwd_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i64, _i64]),
    "bjx_stacked": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _vp, _vp, _vp, _vp, _i64, _i64, _u32]),
    "bjx_stacked_ld": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _u32]),
    "bjx_stacked_mixed": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _u32]),
    "bjx_ordered_ld": (_i, [_vp, _i, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _u32]),
    "bjx_simplex_ld": (_i, [_vp, _i, _i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _u32]),
    "bjx_stacked_vjp": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_stacked_vjp_moments": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_set_option": (_i, [_vp, _i, _i]),
    "bjx_set_rng": (_i, [_vp, _u64, _i64]),
    "bjx_chain": (_i, [_vp, _i, C.POINTER(BjxOp), _i, _vp, _vp] + _tail),
    "bjx_ordered": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_simplex": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_vec_cholesky": (_i, [_vp, _i, _i, _i, _vp, _vp] + _tail),
    "bjx_vec_corr": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_corr": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_pd": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_pd_vec": (_i, [_vp, _i, _i, _vp, _vp] + _tail),
    "bjx_vec_corr_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_corr_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_pd_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_pd_vec_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_scale_matrix": (_i, [_vp, _i, _i, _vp, _vp, _vp] + _tail),
    "bjx_scale_matrix_chain": (_i, [_vp, _i, _i, _vp, C.POINTER(BjxOp), _i, _vp, _vp, _vp, _i64, _i64, _u32]),
    "bjx_scale_matrix_vjp_params": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _d, _vp, _i64, _i64]),
    "bjx_planar": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp] + _tail),
    "bjx_pack_vectors": (_i, [_vp, _i, _i, C.POINTER(_vp), _i64, _vp]),
    "bjx_planar_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_radial_vjp_params": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_planar_vjp_params": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_radial": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp] + _tail),
    "bjx_radial_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_batchnorm": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _d, _vp, _vp] + _tail),
    "bjx_row_moments": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i64]),
    "bjx_batchnorm_train": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp] + _tail),
    "bjx_batchnorm_stats": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i64]),
    "bjx_batchnorm_train_vjp": (_i, [_vp, _i, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_batchnorm_train_apply": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp, _vp] + _tail),
    "bjx_rqs": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp] + _tail),
    "bjx_rqs_vjp": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_rqs_params": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _d, _vp, _vp, _vp]),
    "bjx_rqs_vjp_knots": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_rqs_params_vjp": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bjx_permute": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i64]),
    "bjx_coupling_affine": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp] + _tail),
    "bjx_coupling_affine_vjp": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "bjx_coupling_rqs": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _vp, _vp] + _tail),
    "bjx_comm_unique_id": (_i, [_vp]),
    "bjx_comm_init": (_i, [_vp, _i, _i, _vp]),
    "bjx_comm_destroy": (_i, [_vp]),
    "bjx_allreduce_sum_f64": (_i, [_vp, _vp, _i64]),
    "bjx_fill_normal": (_i, [_vp, _i, _vp, _i64, _i64, _i64, _u64, _d, _d]),
    "bjx_time_begin": (_i, [_vp]),
    "bjx_time_end": (_i, [_vp, C.POINTER(C.c_float)]),
    "bjx_kernel_time_begin": (_i, [_vp]),
    "bjx_kernel_time_end": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "bjx_launch_count": (_u64, []),
    "bjx_plan_chain": (_i, [_vp, _i, C.POINTER(BjxOp), _i, _i64, _u32, C.POINTER(_vp)]),
    "bjx_plan_structured": (_i, [_vp, _i, _i, _i, _i64, _u32, C.POINTER(_vp)]),
    "bjx_plan_run": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "bjx_plan_stacked": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _i64, _u32, C.POINTER(_vp)]),
    "bjx_plan_stacked_vjp": (_i, [_vp, _i, C.POINTER(BjxSegment), _i, _i64, C.POINTER(_vp)]),
    "bjx_plan_run_vjp": (_i, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "bjx_plan_destroy": (_i, [_vp]),
    "bjx_graph_begin": (_i, [_vp]),
    "bjx_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "bjx_graph_launch": (_i, [_vp, _vp]),
    "bjx_graph_destroy": (_i, [_vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load libbjx_hip.so; raises (never falls back) when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [n for n in SIGNATURES if not hasattr(lib, n)]
    if missing:  # an incomplete ABI is a build error, not something to paper over
        raise ImportError(f"{LIB_PATH} does not export {missing}; rebuild it")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class BjxError(RuntimeError):
    """hipError_t / ncclResult_t surfaced by the library (Julia: ErrorException)."""


class BjxFinalizeError(BjxError):
    """BJX_ERR_FINALIZE: the in-kernel finalize of an EARLIER launch timed out (its Σ logabsdetjac is NaN); the context has re-armed
    and switched to the two-pass finalize — repeat the call."""


def check(ctx, code: int, what: str):
    """Map ABI status codes onto the exception classes the reference raises."""
    if code == 0:
        return
    msg = load().bjx_last_error(ctx)
    msg = msg.decode() if msg else ""
    if code == ERR_ARG:
        raise ValueError(f"{what}: {msg}")  # Julia ArgumentError
    if code == ERR_SHAPE:
        raise ValueError(f"{what}: DimensionMismatch: {msg}")
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if code == ERR_FINALIZE:
        raise BjxFinalizeError(f"{what}: {msg}")
    raise BjxError(f"{what}: status {code}: {msg}")
