"""ctypes front-end of the CPU oracle (oracle/bjx_oracle.cpp).

TEST INFRASTRUCTURE ONLY — see the header of bjx_oracle.cpp.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

Arrays use the reference's orientation: numpy arrays of shape (dim, batch) in Fortran
(column-major) order, i.e. exactly a Julia Matrix; 1-D arrays are Julia Vectors (one sample).
Every function returns what the corresponding reference method returns (SURVEY.md §8a').
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbjx_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bjx_oracle.cpp")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Op(C.Structure):
    """Mirror of `bjx_op` (include/bjx.h); v0/v1 are HOST pointers for the oracle."""

    _fields_ = [
        ("kind", C.c_int32),
        ("param_len", C.c_int32),
        ("p0", C.c_double),
        ("p1", C.c_double),
        ("v0", C.c_void_p),
        ("v1", C.c_void_p),
    ]


OP_EXP, OP_LOG, OP_SHIFT, OP_SCALE, OP_SCALE_INV, OP_LOGIT, OP_LOGIT_INV = 1, 2, 3, 4, 5, 6, 7
OP_LEAKY_RELU, OP_TRUNCATED, OP_TRUNCATED_INV, OP_SIGNFLIP, OP_IDENTITY = 8, 9, 10, 11, 12

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        for suf in ("f32", "f64"):
            for name in ("chain", "chain_fused", "find_alpha", "logistic", "log1pexp", "logcosh", "logabsdetjac_inv_corr"):
                getattr(_lib, f"bjo_{name}_{suf}").restype = C.c_double
        _lib.bjo_triu1_dim_from_length.restype = C.c_int64
        _lib.bjo_triu1_dim_from_length.argtypes = [C.c_int64]
    return _lib


def _suf(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return "f32", C.c_float
    if dt == np.float64:
        return "f64", C.c_double
    raise TypeError(f"oracle supports float32/float64, got {dt}")


def _f(x, dtype=None):
    a = np.asfortranarray(np.asarray(x, dtype=dtype))
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _dims(x):
    if x.ndim == 1:
        return x.shape[0], 1
    if x.ndim == 2:
        return x.shape[0], x.shape[1]
    raise ValueError("expected a vector or a (dim, batch) matrix")


def make_ops(ops, dtype, keep):
    """ops: list of (kind, p0, p1) where p0/p1 are floats or 1-D arrays (per-row)."""
    arr = (Op * len(ops))()
    for i, (kind, p0, p1) in enumerate(ops):
        o = arr[i]
        o.kind = kind
        o.param_len = 0
        o.p0 = 0.0
        o.p1 = 0.0
        o.v0 = None
        o.v1 = None
        for j, p in enumerate((p0, p1)):
            if p is None:
                continue
            if np.ndim(p) == 0:
                o.param_len = max(o.param_len, 1)
                setattr(o, f"p{j}", float(p))
            else:
                v = np.ascontiguousarray(np.asarray(p, dtype=dtype))
                keep.append(v)
                o.param_len = v.shape[0]
                setattr(o, f"v{j}", v.ctypes.data)
        # mixed scalar/vector bounds: broadcast the scalar side
        if o.param_len > 1:
            for j, p in enumerate((p0, p1)):
                if p is not None and np.ndim(p) == 0:
                    v = np.full(o.param_len, float(p), dtype=dtype)
                    keep.append(v)
                    setattr(o, f"v{j}", v.ctypes.data)
    return arr


def chain(ops, x, fused=False):
    """with_logabsdet_jacobian of a ComposedFunction chain (application order).  -> (y, ladj scalar)."""
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    keep = []
    arr = make_ops(ops, x.dtype, keep)
    y = np.empty_like(x, order="F")
    fn = getattr(lib(), f"bjo_chain_fused_{suf}" if fused else f"bjo_chain_{suf}")
    l = fn(arr, C.c_int(len(ops)), _p(x), _p(y), C.c_int64(dim), C.c_int64(batch))
    return y, x.dtype.type(l) if not fused else l


def ordered(x, inverse=False):
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_ordered_{suf}")(C.c_int(int(inverse)), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, (ladj if x.ndim == 2 else ladj[0])


def simplex(x, inverse=False, transform=True):
    """-> (out, per-column ladj).  The reference returns the SUM over columns (simplex.jl:141-143)."""
    x = _f(x)
    suf, _ = _suf(x.dtype)
    rows, batch = _dims(x)
    K = rows + 1 if inverse else rows
    orow = K if inverse else K - 1
    out = np.empty((orow, batch) if x.ndim == 2 else (orow,), dtype=x.dtype, order="F") if transform else None
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_simplex_{suf}")(C.c_int(int(inverse)), _p(x), _p(out), C.c_int64(K), C.c_int64(batch), _p(ladj))
    return out, ladj


def vec_cholesky(x, inverse=False, uplo="U", transform=True):
    """inverse: y[n(,N)] -> W[K,K(,N)], logJ;  forward: W -> y, ladj.  Per-sample ladj vector."""
    x = _f(x)
    suf, _ = _suf(x.dtype)
    if inverse:
        nv = x.shape[0]
        batch = 1 if x.ndim == 1 else x.shape[1]
        K = int(lib().bjo_triu1_dim_from_length(nv))
        out = np.empty((K, K, batch) if x.ndim == 2 else (K, K), dtype=x.dtype, order="F") if transform else None
    else:
        K = x.shape[0]
        batch = 1 if x.ndim == 2 else x.shape[2]
        nv = K * (K - 1) // 2
        out = np.empty((nv, batch) if x.ndim == 3 else (nv,), dtype=x.dtype, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_vec_cholesky_{suf}")(C.c_int(int(inverse)), C.c_int(ord(uplo)), _p(x), _p(out), C.c_int64(K), C.c_int64(batch), _p(ladj))
    return out, ladj


MATRIX_KINDS = {"vec_corr": 0, "corr": 1, "pd": 2, "pd_vec": 3}


def matrix_bijector(kind, x, inverse=False):
    """SURVEY.md §8(f) f-4 — VecCorrBijector / CorrBijector (corr.jl:64-162) and PDBijector / PDVecBijector (pd.jl:1-60),
    one sample (x.ndim = 1 or 2) or a batch along the LAST axis.  forward: X[K,K(,N)] -> y[n(,N)] or Y[K,K(,N)];
    inverse: the other way.  Returns (out, per-sample ladj) with ladj what with_logabsdet_jacobian returns."""
    k = MATRIX_KINDS[kind]
    x = _f(x)
    suf, _ = _suf(x.dtype)
    vec_side = k in (0, 3)
    if not inverse:
        K = x.shape[0]
        assert x.shape[1] == K
        batch = 1 if x.ndim == 2 else x.shape[2]
        single = x.ndim == 2
        nv = K * (K - 1) // 2 if k == 0 else (K * (K + 1) // 2 if k == 3 else None)
        out = np.empty(((nv,) if vec_side else (K, K)) + (() if single else (batch,)), dtype=x.dtype, order="F")
    else:
        if vec_side:
            nv = x.shape[0]
            K = int(lib().bjo_triu1_dim_from_length(nv)) if k == 0 else (int(lib().bjo_triu1_dim_from_length(nv)) - 1)
            if k == 3:
                K = (int(round(np.sqrt(1 + 8 * nv))) - 1) // 2          # src/utils.jl:135 _triu_dim_from_length
            batch = 1 if x.ndim == 1 else x.shape[1]
            single = x.ndim == 1
        else:
            K = x.shape[0]
            batch = 1 if x.ndim == 2 else x.shape[2]
            single = x.ndim == 2
        out = np.empty((K, K) + (() if single else (batch,)), dtype=x.dtype, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_matrix_bijector_{suf}")(C.c_int(k), C.c_int(int(inverse)), _p(x), _p(out), C.c_int64(K), C.c_int64(batch), _p(ladj))
    return out, ladj


def vec_corr(x, inverse=False):
    return matrix_bijector("vec_corr", x, inverse)


def corr(x, inverse=False):
    return matrix_bijector("corr", x, inverse)


def pd(x, inverse=False):
    return matrix_bijector("pd", x, inverse)


def pd_vec(x, inverse=False):
    return matrix_bijector("pd_vec", x, inverse)


def logabsdetjac_inv_corr(y):
    """logabsdetjac(inverse(CorrBijector()), Y) (matrix, corr.jl:453-461) / (inverse(VecCorrBijector()), y) (vector, :463-472)."""
    y = _f(y)
    suf, _ = _suf(y.dtype)
    if y.ndim == 1:
        K = int(lib().bjo_triu1_dim_from_length(y.shape[0]))
        return getattr(lib(), f"bjo_logabsdetjac_inv_corr_{suf}")(C.c_int(1), _p(y), C.c_int64(K))
    return getattr(lib(), f"bjo_logabsdetjac_inv_corr_{suf}")(C.c_int(0), _p(y), C.c_int64(y.shape[0]))


def planar(w, u, b, x, inverse=False):
    """w,u: (dim, n_layers) or (dim,), b: (n_layers,) or scalar.  -> (out, per-column ladj)."""
    x = _f(x)
    suf, ct = _suf(x.dtype)
    dim, batch = _dims(x)
    w = _f(np.asarray(w, dtype=x.dtype).reshape(dim, -1))
    u = _f(np.asarray(u, dtype=x.dtype).reshape(dim, -1))
    b = np.ascontiguousarray(np.asarray(b, dtype=x.dtype).reshape(-1))
    nl = w.shape[1]
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_planar_{suf}")(C.c_int(int(inverse)), _p(w), _p(u), _p(b), C.c_int(nl), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def find_alpha(wt_y, wt_u_hat, b, dtype=np.float64):
    suf, ct = _suf(dtype)
    return getattr(lib(), f"bjo_find_alpha_{suf}")(ct(wt_y), ct(wt_u_hat), ct(b))


def radial(alpha_, beta, z0, x, inverse=False):
    x = _f(x)
    suf, ct = _suf(x.dtype)
    dim, batch = _dims(x)
    z0 = np.ascontiguousarray(np.asarray(z0, dtype=x.dtype))
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_radial_{suf}")(C.c_int(int(inverse)), ct(float(np.asarray(alpha_).reshape(-1)[0])), ct(float(np.asarray(beta).reshape(-1)[0])), _p(z0), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def batchnorm(b, logs, m, v, eps, x, inverse=False):
    x = _f(x)
    suf, ct = _suf(x.dtype)
    dim, batch = _dims(x)
    ps = [np.ascontiguousarray(np.asarray(p, dtype=x.dtype)) for p in (b, logs, m, v)]
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_batchnorm_{suf}")(C.c_int(int(inverse)), *[_p(p) for p in ps], ct(float(eps)), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def rqs(widths, heights, derivs, x, inverse=False):
    """knot matrices (dim, K+1) Fortran order; x (dim, batch) or (dim,).  -> (out, per-column ladj)."""
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    w, h, d = (_f(np.asarray(a, dtype=x.dtype).reshape(dim, -1)) for a in (widths, heights, derivs))
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_rqs_{suf}")(C.c_int(int(inverse)), _p(w), _p(h), _p(d), C.c_int64(w.shape[1]), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def rqs_params(raw_w, raw_h, raw_d, B):
    raw_w = _f(raw_w)
    suf, ct = _suf(raw_w.dtype)
    dim, K = raw_w.shape
    raw_h = _f(np.asarray(raw_h, dtype=raw_w.dtype))
    raw_d = _f(np.asarray(raw_d, dtype=raw_w.dtype))
    w, h, d = (np.empty((dim, K + 1), dtype=raw_w.dtype, order="F") for _ in range(3))
    getattr(lib(), f"bjo_rqs_params_{suf}")(_p(raw_w), _p(raw_h), _p(raw_d), C.c_int64(K), C.c_int64(dim), ct(float(B)), _p(w), _p(h), _p(d))
    return w, h, d


def permute(src, x):
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    src = np.ascontiguousarray(np.asarray(src, dtype=np.int32))
    out = np.empty_like(x, order="F")
    getattr(lib(), f"bjo_permute_{suf}")(_p(src), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch))
    return out


def coupling_affine(idx1, scale, shift, x, inverse=False):
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    idx1 = np.ascontiguousarray(np.asarray(idx1, dtype=np.int32))
    s = _f(np.asarray(scale, dtype=x.dtype)) if scale is not None else None
    t = _f(np.asarray(shift, dtype=x.dtype)) if shift is not None else None
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_coupling_affine_{suf}")(C.c_int(int(inverse)), _p(idx1), C.c_int64(len(idx1)), _p(s), _p(t), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def coupling_rqs(idx1, widths, heights, derivs, x, inverse=False):
    x = _f(x)
    suf, _ = _suf(x.dtype)
    dim, batch = _dims(x)
    idx1 = np.ascontiguousarray(np.asarray(idx1, dtype=np.int32))
    n1 = len(idx1)
    w, h, d = (_f(np.asarray(a, dtype=x.dtype).reshape(n1, -1)) for a in (widths, heights, derivs))
    out = np.empty_like(x, order="F")
    ladj = np.empty(batch, dtype=x.dtype)
    getattr(lib(), f"bjo_coupling_rqs_{suf}")(C.c_int(int(inverse)), _p(idx1), C.c_int64(n1), _p(w), _p(h), _p(d), C.c_int64(w.shape[1]), _p(x), _p(out), C.c_int64(dim), C.c_int64(batch), _p(ladj))
    return out, ladj


def logistic(x, dtype=np.float64):
    suf, ct = _suf(dtype)
    return getattr(lib(), f"bjo_logistic_{suf}")(ct(x))


def log1pexp(x, dtype=np.float64):
    suf, ct = _suf(dtype)
    return getattr(lib(), f"bjo_log1pexp_{suf}")(ct(x))


def logcosh(x, dtype=np.float64):
    suf, ct = _suf(dtype)
    return getattr(lib(), f"bjo_logcosh_{suf}")(ct(x))


# ------------------------------------------------------------------ reverse-mode pullbacks
def ordered_vjp(inp, out_bar, ladj_bar=None, inverse=False):
    """Pullback of with_logabsdet_jacobian(OrderedBijector() or its inverse, inp) for a (dim, batch) matrix:
    ext/BijectorsChainRulesCoreExt.jl:90-112 (_transform_ordered, matrix) and :149-197
    (_transform_inverse_ordered, matrix), with the log-det cotangent added (ordered.jl:79-80:
    logabsdetjac = sum(y[2:end, :]; dims=1), and its negative for the inverse, interface.jl:276-281).
    numpy restatement, row loop in the reference's order, vectorised over the batch."""
    a = np.asarray(inp)
    g = np.asarray(out_bar, dtype=a.dtype)
    n, N = a.shape
    lb = np.zeros(N, dtype=a.dtype) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=a.dtype), (N,))
    res = np.empty_like(a)
    if not inverse:
        s = g.sum(axis=0)                      # :99
        res[0] = s                             # :100
        for i in range(1, n):                  # :101-108
            s = s - g[i - 1]
            res[i] = s * np.exp(a[i]) + lb
        return res
    r = np.ones_like(a)                       # :153-160
    r[1:] = a[1:] - a[:-1]
    d = g.copy()
    d[1:] -= lb                                # cotangent of -sum(log r_i, i >= 2)
    for i in range(n - 1):                     # :168-170
        res[i] = d[i] / r[i] - d[i + 1] / r[i + 1]
    res[n - 1] = d[n - 1] / r[n - 1]           # :172-174
    return res


def vec_cholesky_inv_vjp(y, W_bar, logJ_bar=None, uplo="U"):
    """Pullback of _inv_link_chol_lkj for y (n, batch), W_bar (K, K, batch), logJ_bar (batch,):
    src/bijectors/corr.jl:402-451 (_inv_link_chol_lkj_rrule), loops in the reference's order,
    vectorised over the batch.  uplo="L": W is the transposed (lower) factor, so ΔW is read transposed.
    At z == 0 exactly the reference's (inv(z) - z) * W[i,j] * ΔW[i,j] is Inf * 0; its limit
    (1 - z²) exp(log_remainder) ΔW[i,j] is used there."""
    y = np.asarray(y)
    n, N = y.shape
    K = (1 + int(round(np.sqrt(1 + 8 * n)))) // 2
    dW = np.asarray(W_bar, dtype=y.dtype)
    if uplo == "L":
        dW = np.transpose(dW, (1, 0, 2))
    dl = np.zeros(N, dtype=y.dtype) if logJ_bar is None else np.broadcast_to(np.asarray(logJ_bar, dtype=y.dtype), (N,))
    z_vec = np.tanh(y)                                                        # :411
    lc_vec = np.abs(y) + np.log1p(np.exp(-2 * np.abs(y))) - np.log(2.0)       # :412 LogExpFunctions.logcosh
    W = np.zeros((K, K, N), dtype=y.dtype)
    E = np.empty_like(y)                                                      # exp(log_remainder) before each entry
    W[0, 0] = 1
    idx = 0
    for j in range(1, K):                                                     # :416-431 (primal)
        lr = np.zeros(N, dtype=y.dtype)
        for i in range(j):
            E[idx] = np.exp(lr)
            W[i, j] = z_vec[idx] * E[idx]
            lr = lr - lc_vec[idx]
            idx += 1
        W[j, j] = np.exp(lr)
    dy = np.empty_like(y)
    idx = n - 1
    for j in range(K - 1, 0, -1):                                             # :437-447
        dlr = W[j, j] * dW[j, j] + 2 * dl
        for i in range(j - 1, -1, -1):
            W_dW = W[i, j] * dW[i, j]
            z = z_vec[idx]
            with np.errstate(divide="ignore", invalid="ignore"):
                first = np.where(z == 0, E[idx] * dW[i, j], (1 / z - z) * W_dW)
            dy[idx] = first - z * dlr
            idx -= 1
            dlr = dlr + dl + W_dW
    return dy


def chain_vjp(ops, x, y_bar, ladj_bar=None):
    """Input pullback of with_logabsdet_jacobian for a chain of elementwise bijectors, (dim, batch) input:
    x_bar = (dy/dx) y_bar + ladj_bar (d ladj/dx) with ladj the PER-COLUMN log-det.  Closed-form derivatives of
    the reference's scalar maps (exp_log.jl:5-9, shift.jl:14, scale.jl:13, logit.jl:15-30,
    leaky_relu.jl:25-29, truncated.jl:15-91), applied in reverse over the stages; numpy, float64."""
    x = np.asarray(x, dtype=np.float64)
    dim, N = x.shape
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    col = lambda p: np.broadcast_to(np.asarray(p, dtype=np.float64).reshape(-1, 1), (dim, 1)) if np.ndim(p) else float(p)
    stages = []            # (dy/dx, dl/dx) of every stage at its own input
    v = x
    for kind, p0, p1 in ops:
        a = None if p0 is None else col(p0)
        b = None if p1 is None else col(p1)
        one, zero = np.ones_like(v), np.zeros_like(v)
        if kind == OP_EXP:
            out, dy, dl = np.exp(v), np.exp(v), one
        elif kind == OP_LOG:
            out, dy, dl = np.log(v), 1 / v, -1 / v
        elif kind == OP_SHIFT:
            out, dy, dl = v + a, one, zero
        elif kind == OP_SCALE:
            out, dy, dl = v * a, one * a, zero
        elif kind == OP_SCALE_INV:
            out, dy, dl = v / a, one / a, zero
        elif kind == OP_SIGNFLIP:
            out, dy, dl = -v, -one, zero
        elif kind == OP_LOGIT:
            w = b - a
            z = (v - a) / w
            out, dy = np.log(z / (1 - z)), 1 / (w * z * (1 - z))
            dl = -(1 - 2 * z) / (w * z * (1 - z))                 # d/dx of -log((x-a)(b-x)/(b-a))
        elif kind == OP_LOGIT_INV:
            w = b - a
            sg = 1 / (1 + np.exp(-v))
            out, dy, dl = w * sg + a, w * sg * (1 - sg), 1 - 2 * sg   # ladj = log(s(1-s)) + log(b-a)
        elif kind == OP_LEAKY_RELU:
            J = np.where(v < 0, a, 1.0)
            out, dy, dl = J * v, J * one, zero
        elif kind in (OP_TRUNCATED, OP_TRUNCATED_INV):
            lo, hi = a * one, b * one
            lbm, ubm = np.isfinite(lo), np.isfinite(hi)
            if kind == OP_TRUNCATED:
                inside = (v >= lo) & (v <= hi)
                xc = np.clip(v, lo, hi)
                with np.errstate(all="ignore"):
                    w = hi - lo
                    z = (xc - lo) / w
                    out = np.where(lbm & ubm, np.log(z / (1 - z)), np.where(lbm, np.log(xc - lo), np.where(ubm, np.log(hi - xc), xc)))
                    dy = np.where(lbm & ubm, 1 / (w * z * (1 - z)), np.where(lbm, 1 / (xc - lo), np.where(ubm, -1 / (hi - xc), 1.0)))
                    dl = np.where(lbm & ubm, -(1 - 2 * z) / (w * z * (1 - z)), np.where(lbm, -1 / (xc - lo), np.where(ubm, 1 / (hi - xc), 0.0)))
                dy, dl = np.where(inside, dy, 0.0), np.where(inside, dl, 0.0)
            else:
                with np.errstate(all="ignore"):
                    w = hi - lo
                    sg = 1 / (1 + np.exp(-v))
                    ev = np.exp(v)
                    raw = np.where(lbm & ubm, w * sg + lo, np.where(lbm, ev + lo, np.where(ubm, hi - ev, v)))
                    dy = np.where(lbm & ubm, w * sg * (1 - sg), np.where(lbm, ev, np.where(ubm, -ev, 1.0)))
                    dl = np.where(lbm & ubm, 1 - 2 * sg, np.where(lbm | ubm, 1.0, 0.0))
                out = np.clip(raw, lo, hi)
                dy = np.where((raw >= lo) & (raw <= hi), dy, 0.0)
        else:
            raise ValueError(kind)
        stages.append((dy, dl))
        v = out
    g = np.asarray(y_bar, dtype=np.float64)
    for dy, dl in reversed(stages):
        g = g * dy + lb * dl
    return g


def _simplex_terms(x_k, s_k, eps, first):
    """t_k of logabsdetjac (simplex.jl:122-138) and its partials w.r.t. x_k and s_k (s_k = sum of the earlier rows)."""
    if first:
        m1, m2 = np.maximum(x_k, eps), np.maximum(1 - x_k, eps)
        t = np.log(m1) + np.log(m2)
        dtdx = np.where(x_k > eps, 1 / m1, 0.0) - np.where(1 - x_k > eps, 1 / m2, 0.0)
        return t, dtdx, np.zeros_like(x_k)
    M = np.maximum(1 - s_k, eps)
    zl = x_k / M
    m1, m2 = np.maximum(zl, eps), np.maximum(1 - zl, eps)
    t = np.log(m1) + np.log(m2) + np.log(M)
    dtdzl = np.where(zl > eps, 1 / m1, 0.0) - np.where(1 - zl > eps, 1 / m2, 0.0)
    dtdx = dtdzl / M
    dtdM = dtdzl * (-x_k / M**2) + 1 / M
    dtds = -np.where(1 - s_k > eps, dtdM, 0.0)
    return t, dtdx, dtds


def simplex_vjp(inp, out_bar, ladj_bar=None, inverse=False, eps=None):
    """Pullback of with_logabsdet_jacobian(SimplexBijector() or its inverse, inp), (rows, batch) input, per-column
    log-det cotangent.  Reverse sweep of the stick-breaking recurrences simplex.jl:47-64 / :102-120 and of the
    log-det terms :122-138 (the reference's own adjoints: simplex.jl:145-215 logabsdetjac gradient, :248-308
    link, :358-470 invlink — O(K²) loops there, O(K) here; same derivative conventions: a clamped value has
    zero derivative, max(v, ε) has derivative 1 only where v > ε).  numpy float64, loops over rows only.

    `eps`: the reference's ε is `eps(T)` of the ELEMENT TYPE it is called with (src/Bijectors.jl:91-93, simplex.jl:32,88,126) — part of the
    function, not of the arithmetic: max(z, ε) switches its derivative off below ε, and the ε-shifted sticks differ by ε/remainder.
    Default: ε of `inp`'s dtype (Float32 input -> Float32 ε, evaluated here in Float64 arithmetic); round 5 used the Float64 ε for
    Float32 inputs too, which is a different function wherever a stick is within ~1e-7 of a kink (round 6: the flat-tolerance pullback
    tests found it — up to 2.5 % of a column's cotangent scale at K = 64)."""
    if eps is None:
        in_dt = np.asarray(inp).dtype
        eps = float(np.finfo(in_dt if in_dt in (np.float32, np.float64) else np.float64).eps)
    a = np.asarray(inp, dtype=np.float64)
    g = np.asarray(out_bar, dtype=np.float64)
    c, E = 1 / (1 - 2 * eps), 1 + eps
    N = a.shape[1]
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    if inverse:
        y = a
        K = y.shape[0] + 1
        lk = np.log(np.arange(K - 1, 0, -1, dtype=np.float64))          # log(K-k), k = 1..K-1
        z = 1 / (1 + np.exp(-(y - lk[:, None])))
        x = np.zeros((K, N))
        u = np.zeros((K - 1, N))
        s = np.zeros((K + 1, N))                                        # s[k] = sum of x[:k]
        for k in range(K - 1):
            r = E - s[k]
            u[k] = (z[k] - eps) * c if k == 0 else r * c * z[k] - eps
            x[k] = np.clip(u[k], 0, 1)
            s[k + 1] = s[k] + x[k]
        last = 1 - s[K - 1]
        x[K - 1] = np.clip(last, 0, 1)
        yb = np.zeros_like(y)
        sb = -np.where((last > 0) & (last < 1), g[K - 1], 0.0)          # adjoint of s[K-1]
        for k in range(K - 2, -1, -1):
            xb = g[k] + sb                                              # s[k+1] = s[k] + x[k]
            _, dtdx, dtds = _simplex_terms(x[k], s[k], eps, k == 0)
            xb = xb + lb * dtdx                                         # ladj(inverse) = + sum_k t_k
            sb = sb + lb * dtds
            ub = np.where((u[k] > 0) & (u[k] < 1), xb, 0.0)
            if k == 0:
                zb = ub * c
            else:
                r = E - s[k]
                zb = ub * r * c
                sb = sb - ub * c * z[k]
            yb[k] = zb * z[k] * (1 - z[k])
        return yb
    x = a
    K = x.shape[0]
    s = np.zeros((K + 1, N))
    for k in range(K):
        s[k + 1] = s[k] + x[k]
    xb = np.zeros_like(x)
    sb_next = np.zeros(N)                                               # adjoint of s[k+1], carried down
    for k in range(K - 2, -1, -1):
        # s[k+1] = s[k] + x[k]: x[k] and s[k] both receive the adjoint of s[k+1]
        sb = sb_next.copy()
        xbk = sb_next.copy()
        if k == 0:
            zf = x[0] * (1 - 2 * eps) + eps
            zfb = g[0] / (zf * (1 - zf))
            xbk = xbk + zfb * (1 - 2 * eps)
        else:
            d = E - s[k]
            an = (x[k] + eps) * (1 - 2 * eps)
            zf = an / d
            zfb = g[k] / (zf * (1 - zf))
            xbk = xbk + zfb * (1 - 2 * eps) / d
            sb = sb + zfb * an / d**2                                   # d = E - s[k]
        _, dtdx, dtds = _simplex_terms(x[k], s[k], eps, k == 0)
        xbk = xbk - lb * dtdx                                           # ladj(forward) = - sum_k t_k
        sb = sb - lb * dtds
        xb[k] = xbk
        sb_next = sb
    xb[K - 1] = 0.0                                                     # row K enters neither y nor the log-det
    return xb


def batchnorm_train(b, logs, m, v, eps, mtm, x):
    """InvertibleBatchNorm with istraining() == true on a (channels, batch) matrix — normalise.jl:41-68, numpy:
    -> (result, per-column logabsdetjac, updated moving mean, updated moving variance)."""
    x = np.asarray(x)
    T = x.dtype
    b, logs, m, v = (np.asarray(p, dtype=T) for p in (b, logs, m, v))
    n = x.shape[1]
    mb = x.mean(axis=1, dtype=np.float64)                                   # :53
    vb = ((x.astype(np.float64) - mb[:, None]) ** 2).sum(axis=1) / n        # :54
    mbT, vbT = mb.astype(T), vb.astype(T)
    m_new = (1 - T.type(mtm)) * m + T.type(mtm) * mbT                       # :58
    v_new = (1 - T.type(mtm)) * v + T.type(mtm * n / (n - 1)) * vbT         # :59
    s = np.exp(logs)
    result = s[:, None] * (x - mbT[:, None]) / np.sqrt(vbT + T.type(eps))[:, None] + b[:, None]   # :66
    ladj = np.full(n, (logs - np.log(vbT + T.type(eps)) / 2).sum(), dtype=T)                     # :67
    return result, ladj, m_new, v_new


def mvnormal_full_logpdf(x, mu, cov):
    """Per-column log-density of MvNormal(mu, cov) with a FULL covariance — the `logpdf(td.dist, x)` term of
    src/transformed_distribution.jl:165-169 (the density itself lives in Distributions.jl / PDMats.jl, un-vendored: its
    textbook formula through the Cholesky factor).  numpy, Float64."""
    x = np.asarray(x, dtype=np.float64)
    mu, cov = np.asarray(mu, dtype=np.float64).reshape(-1), np.asarray(cov, dtype=np.float64)
    Lc = np.linalg.cholesky(cov)
    z = np.linalg.solve(Lc, x - mu[:, None])
    return -0.5 * np.sum(z * z, axis=0) - np.sum(np.log(np.diag(Lc))) - 0.5 * x.shape[0] * np.log(2.0 * np.pi)


def batchnorm_train_vjp(logs, eps, x, out_bar, ladj_bar):
    """Pullback of `batchnorm_train` (normalise.jl:51-60: batch mean / biased batch variance are functions of x; the reference
    leaves this adjoint to the AD package) — numpy Float64 restatement of the closed form, pinned by central differences of
    `batchnorm_train` in tests/test_oracle_golden.py.  -> (x_bar, b_bar, logs_bar)."""
    x, g = np.asarray(x, dtype=np.float64), np.asarray(out_bar, dtype=np.float64)
    logs = np.asarray(logs, dtype=np.float64)
    lb = np.zeros(x.shape[1]) if ladj_bar is None else np.asarray(ladj_bar, dtype=np.float64)
    n = x.shape[1]
    m = x.mean(axis=1)
    v = ((x - m[:, None]) ** 2).sum(axis=1) / n
    sig = np.sqrt(v + eps)
    xh = (x - m[:, None]) / sig[:, None]
    gam = np.exp(logs)
    L = lb.sum()
    x_bar = (gam / sig)[:, None] * (g - g.mean(axis=1)[:, None] - xh * (g * xh).mean(axis=1)[:, None]) - (L / n) * xh / sig[:, None]
    return x_bar, g.sum(axis=1), gam * (g * xh).sum(axis=1) + L


# ------------------------------------------------------------------ SURVEY.md §8(f) f-3
def mvnormal_diag_logpdf(x, mu=None, sigma=None):
    """Per-column log-density of MvNormal(mu, Diagonal(sigma.^2)) — the `logpdf(td.dist, x)` term of
    src/transformed_distribution.jl:165-169 for the diagonal-normal bases the flows use (the density itself
    lives in Distributions.jl, un-vendored; this is its textbook formula).  numpy, float64 accumulation."""
    x = np.asarray(x, dtype=np.float64)
    d = x.shape[0]
    m = np.zeros(d) if mu is None else np.asarray(mu, dtype=np.float64).reshape(-1)
    s = np.ones(d) if sigma is None else np.asarray(sigma, dtype=np.float64).reshape(-1)
    z = (x - m[:, None]) / s[:, None]
    return -0.5 * np.sum(z * z, axis=0) - np.sum(np.log(s)) - 0.5 * d * np.log(2.0 * np.pi)


def planar_inv_vjp(w, u, b, y, x_bar, ladj_bar=None):
    """Input pullback of with_logabsdet_jacobian(inverse(flow), y) for a PlanarLayer stack (planar_layer.jl:112-127 with
    the implicit-function rule of find_alpha, ext/BijectorsChainRulesCoreExt.jl:42-46: dα/d(wᵀy) = 1/(1 + c sech²(α+b))).
    The inverse undoes the LAST layer first; per layer, with t = tanh(α + b), q = 1 - t²:
        z = y - û t,  logabsdetjac = -log1p(c q)
        s̄ = q/(1 + c q) · (-ûᵀz̄ + ℓ̄ · 2 c t/(1 + c q)),   ȳ = z̄ + w s̄.
    The t of every layer are read off a forward pass over the pre-image (α_k = w_kᵀz_{k-1}).  numpy, float64."""
    y = np.asarray(y, dtype=np.float64)
    dim, N = y.shape
    w = np.asarray(w, dtype=np.float64).reshape(dim, -1)
    u = np.asarray(u, dtype=np.float64).reshape(dim, -1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    nl = w.shape[1]
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    x, _ = planar(w, u, b, np.asfortranarray(y), inverse=True)
    u_hat, c = np.empty_like(u), np.empty(nl)
    for k in range(nl):
        wtu = float(w[:, k] @ u[:, k])
        u_hat[:, k] = u[:, k] + (log1pexp(-wtu) - 1.0) / float(w[:, k] @ w[:, k]) * w[:, k]
        c[k] = log1pexp(wtu) - 1.0
    ts, cur = [], np.asarray(x, dtype=np.float64)
    for k in range(nl):
        t = np.tanh(w[:, k] @ cur + b[k])
        ts.append(t)
        cur = cur + np.outer(u_hat[:, k], t)
    g = np.asarray(x_bar, dtype=np.float64).copy()          # cotangent of the pre-image = output of the inverse of layer 0
    for k in range(nl):                                       # the inverse applied layer nl-1 first: pull back in the opposite order
        t = ts[k]
        q = 1.0 - t * t
        den = 1.0 + c[k] * q
        sbar = q / den * (-(u_hat[:, k] @ g) + lb * 2.0 * c[k] * t / den)
        g = g + np.outer(w[:, k], sbar)
    return g


def planar_vjp(w, u, b, z, y_bar, ladj_bar=None):
    """Input pullback of with_logabsdet_jacobian for a stack of PlanarLayers (planar_layer.jl:65-110; the reference
    leaves it to the AD package — these are the closed-form derivatives of its expressions):
        s_k = w_kᵀz_{k-1} + b_k,  t_k = tanh s_k,  z_k = z_{k-1} + û_k t_k,  ℓ_k = log1p(c_k (1 - t_k²)),  c_k = w_kᵀû_k
        s̄_k = (û_kᵀ z̄_k)(1 - t_k²) + ℓ̄ · c_k (-2 t_k)(1 - t_k²) / (1 + c_k (1 - t_k²)),   z̄_{k-1} = z̄_k + w_k s̄_k.
    w, u: (dim, n_layers) or (dim,); z, y_bar: (dim, N); ladj_bar: (N,) or None.  numpy, float64."""
    z = np.asarray(z, dtype=np.float64)
    dim, N = z.shape
    w = np.asarray(w, dtype=np.float64).reshape(dim, -1)
    u = np.asarray(u, dtype=np.float64).reshape(dim, -1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    nl = w.shape[1]
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    u_hat, c = np.empty_like(u), np.empty(nl)
    for k in range(nl):
        wtu = float(w[:, k] @ u[:, k])
        u_hat[:, k] = u[:, k] + (log1pexp(-wtu) - 1.0) / float(w[:, k] @ w[:, k]) * w[:, k]   # planar_layer.jl:65-70
        c[k] = log1pexp(wtu) - 1.0
    ts, cur = [], z
    for k in range(nl):
        t = np.tanh(w[:, k] @ cur + b[k])
        ts.append(t)
        cur = cur + np.outer(u_hat[:, k], t)
    g = np.asarray(y_bar, dtype=np.float64).copy()
    for k in range(nl - 1, -1, -1):
        t = ts[k]
        q = 1.0 - t * t
        sbar = (u_hat[:, k] @ g) * q + lb * c[k] * (-2.0 * t) * q / (1.0 + c[k] * q)
        g = g + np.outer(w[:, k], sbar)
    return g


def vec_cholesky_fwd_vjp(W, y_bar, uplo="U"):
    """Pullback of `_link_chol_lkj_from_upper` / `_from_lower` as the reference ships it
    (ext/BijectorsChainRulesCoreExt.jl:199-254 / :256-311): the rule lives on the constraint manifold of Cholesky
    factors of correlation matrices (unit-norm columns; the strict triangle is the free parameter, the diagonal is a
    function of it), so ΔW[j,j] = 0 and the entries outside the strict triangle are left undefined there (zeros here).
    W: (K, K) or (K, K, N) column-major per sample; y_bar: (n,) or (n, N).  numpy, float64, the reference's loops."""
    W = np.asarray(W, dtype=np.float64)
    single = W.ndim == 2
    Wb = W[:, :, None] if single else W
    K, _, N = Wb.shape
    n = K * (K - 1) // 2
    yb = np.asarray(y_bar, dtype=np.float64).reshape(n, -1)
    out = np.zeros_like(Wb)
    for s in range(N):
        A = Wb[:, :, s] if uplo == "U" else Wb[:, :, s].T      # work on the upper factor (the :L rule is its transpose)
        dA = np.zeros((K, K))
        for j in range(1, K):                                     # 0-based column j has rows 0..j-1 above the diagonal
            base = j * (j - 1) // 2
            rs = A[j, j] ** 2
            dtmp = 0.0
            for i in range(j - 1, 0, -1):
                w = A[i, j]
                rs += w * w
                tmp = np.sqrt(rs)                                 # remainders[...]: sqrt(W[j,j]² + Σ_{i'>=i} W[i',j]²)
                p = w / tmp
                ftmp = np.sqrt(1.0 - p * p)
                d_ftmp_p = -p / ftmp
                d_p_tmp = -w / (tmp * tmp)
                dp = yb[base + i, s] / (1.0 - p * p) + dtmp * tmp * d_ftmp_p
                dA[i, j] = dp / tmp
                dtmp = dp * d_p_tmp + dtmp * ftmp
            w0 = A[0, j]
            dA[0, j] = yb[base, s] / (1.0 - w0 * w0) - dtmp / np.sqrt(1.0 - w0 * w0) * w0
        out[:, :, s] = dA if uplo == "U" else dA.T
    return out[:, :, 0] if single else out


def radial_vjp(alpha_, beta, z0, x, out_bar, ladj_bar=None, inverse=False):
    """Input pullback of with_logabsdet_jacobian for a RadialLayer and its inverse (radial_layer.jl:43-129; closed-form
    derivatives — the reference leaves them to the AD package).  With δ = z - z₀, r = ‖δ‖, h = 1/(α + r):
        J = ∂f/∂z = (1 + β̂h) I - (β̂h²/r) δδᵀ   (symmetric),   ℓ = (d-1) log(1 + β̂h) + log(1 + β̂h - β̂h²r)
        forward:  z̄ = J ȳ + ℓ̄ ℓ'(r) δ/r;       inverse (y ↦ z, log-det -ℓ(z)):  ȳ = J⁻¹ (z̄ - ℓ̄ ℓ'(r) δ/r)
    numpy, float64; x, out_bar: (dim, N)."""
    x = np.asarray(x, dtype=np.float64)
    d, N = x.shape
    z0 = np.asarray(z0, dtype=np.float64).reshape(-1, 1)
    al = float(log1pexp(float(np.asarray(alpha_).reshape(-1)[0])))
    bh = -al + float(log1pexp(float(np.asarray(beta).reshape(-1)[0])))
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    g = np.asarray(out_bar, dtype=np.float64)
    if inverse:
        z, _ = radial(np.asarray(alpha_, dtype=np.float64), np.asarray(beta, dtype=np.float64), z0.reshape(-1), np.asfortranarray(x), inverse=True)
    else:
        z = x
    dl = z - z0
    r = np.sqrt((dl * dl).sum(axis=0))
    h = 1.0 / (al + r)
    a = 1.0 + bh * h
    with np.errstate(divide="ignore", invalid="ignore"):
        rinv = np.where(r > 0, 1.0 / r, 0.0)
    c = -bh * h * h * rinv
    lr = (d - 1) * (-bh * h * h) / a + (-2.0 * bh * h * h + 2.0 * bh * h ** 3 * r) / (1.0 + bh * h - bh * h * h * r)
    if not inverse:
        return a * g + c * (dl * g).sum(axis=0) * dl + lb * lr * rinv * dl
    v = g - lb * lr * rinv * dl
    return (v - c * (dl * v).sum(axis=0) * dl / (a + c * r * r)) / a


def radial_param_vjp(alpha_, beta, z0, z, y_bar, ladj_bar=None):
    """Parameter pullback of with_logabsdet_jacobian for a RadialLayer: (ᾱ_, β̄, z̄₀) summed over the batch, for the RAW
    parameters behind softplus (radial_layer.jl:43-60; the reference leaves it to the AD package).  With α̂ = softplus(α_),
    β̂ = -α̂ + softplus(β), h = 1/(α̂ + r), a = 1 + β̂h, D = 1 + β̂h - β̂h²r:
        g_β̂ = h δᵀȳ + ℓ̄[(d-1)h/a + (h - h²r)/D],  g_α̂ = -h²[β̂ δᵀȳ + ℓ̄((d-1)β̂/a + (β̂ - 2β̂hr)/D)],
        ᾱ_ = σ(α_)(Σ g_α̂ - Σ g_β̂),  β̄ = σ(β) Σ g_β̂,  z̄₀ = Σ_n (ȳ_n - z̄_n).   numpy, float64."""
    z = np.asarray(z, dtype=np.float64)
    d, N = z.shape
    a_raw = float(np.asarray(alpha_).reshape(-1)[0])
    b_raw = float(np.asarray(beta).reshape(-1)[0])
    al = float(log1pexp(a_raw))
    bh = -al + float(log1pexp(b_raw))
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    g = np.asarray(y_bar, dtype=np.float64)
    dl = z - np.asarray(z0, dtype=np.float64).reshape(-1, 1)
    r = np.sqrt((dl * dl).sum(axis=0))
    dg = (dl * g).sum(axis=0)
    h = 1.0 / (al + r)
    a = 1.0 + bh * h
    D = 1.0 + bh * h - bh * h * h * r
    gb = h * dg + lb * ((d - 1) * h / a + (h - h * h * r) / D)
    ga = -h * h * (bh * dg + lb * ((d - 1) * bh / a + (bh - 2.0 * bh * h * r) / D))
    sa, sb = 1.0 / (1.0 + np.exp(-a_raw)), 1.0 / (1.0 + np.exp(-b_raw))
    zb = radial_vjp(alpha_, beta, z0, z, g, lb)
    return sa * (ga.sum() - gb.sum()), sb * gb.sum(), (g - zb).sum(axis=1)


def planar_param_vjp(w, u, b, z, y_bar, ladj_bar=None):
    """Parameter pullback of with_logabsdet_jacobian for a PlanarLayer stack: (w̄, ū, b̄) summed over the batch
    (planar_layer.jl:65-110 incl. get_u_hat :65-70; the reference leaves it to the AD package).  Per layer, with
    s̄_k, t_k, q_k = 1 - t_k² of `planar_vjp`:
        b̄_k = Σ_n s̄_kn,   w̄_k(direct) = Σ_n z_{k-1,n} s̄_kn,   û̄_k = Σ_n z̄_{k,n} t_kn,   c̄_k = Σ_n ℓ̄_n q_kn/(1 + c_k q_kn)
    and through û = u + κ w, κ = (log1pexp(-a) - 1)/‖w‖², c = log1pexp(a) - 1, a = wᵀu:
        ū = û̄ + w [κ_a (wᵀû̄) + c̄ σ(a)],   w̄ = w̄(direct) + κ û̄ + (wᵀû̄)(κ_a u - 2κ/‖w‖² w) + c̄ σ(a) u,   κ_a = -σ(-a)/‖w‖².
    numpy, float64.  Returns (w_bar, u_bar, b_bar) with the shapes of (w, u, b)."""
    z = np.asarray(z, dtype=np.float64)
    dim, N = z.shape
    w = np.asarray(w, dtype=np.float64).reshape(dim, -1)
    u = np.asarray(u, dtype=np.float64).reshape(dim, -1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    nl = w.shape[1]
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    u_hat, c, a, n2, kap = np.empty_like(u), np.empty(nl), np.empty(nl), np.empty(nl), np.empty(nl)
    for k in range(nl):
        a[k] = float(w[:, k] @ u[:, k])
        n2[k] = float(w[:, k] @ w[:, k])
        kap[k] = (log1pexp(-a[k]) - 1.0) / n2[k]
        u_hat[:, k] = u[:, k] + kap[k] * w[:, k]
        c[k] = log1pexp(a[k]) - 1.0
    zs, ts, cur = [], [], z
    for k in range(nl):
        zs.append(cur)
        t = np.tanh(w[:, k] @ cur + b[k])
        ts.append(t)
        cur = cur + np.outer(u_hat[:, k], t)
    g = np.asarray(y_bar, dtype=np.float64).copy()
    wb, ub, bb = np.zeros_like(w), np.zeros_like(u), np.zeros(nl)
    for k in range(nl - 1, -1, -1):
        t = ts[k]
        q = 1.0 - t * t
        den = 1.0 + c[k] * q
        uhb = g @ t                                                    # û̄_k = Σ_n z̄_{k,n} t_kn
        cb = float(np.sum(lb * q / den))
        sbar = (u_hat[:, k] @ g) * q + lb * c[k] * (-2.0 * t) * q / den
        bb[k] = sbar.sum()
        wdir = zs[k] @ sbar
        g = g + np.outer(w[:, k], sbar)
        ka = -sig(-a[k]) / n2[k]
        wtuhb = float(w[:, k] @ uhb)
        ub[:, k] = uhb + w[:, k] * (ka * wtuhb + cb * sig(a[k]))
        wb[:, k] = wdir + kap[k] * uhb + wtuhb * (ka * u[:, k] - 2.0 * kap[k] / n2[k] * w[:, k]) + cb * sig(a[k]) * u[:, k]
    return wb, ub, bb


def rqs_vjp(widths, heights, derivs, x, out_bar, ladj_bar=None, inverse=False):
    """Input pullback of with_logabsdet_jacobian for the elementwise RationalQuadraticSpline and its inverse
    (rational_quadratic_spline.jl:128-357; closed-form derivatives — the reference leaves them to the AD package):
        f'(x) = s²(d_{k+1}ξ² + 2sξ(1-ξ) + d_k(1-ξ)²)/den²,  den = s + (d_{k+1} + d_k - 2s)ξ(1-ξ),
        x̄ = ȳ f' + ℓ̄ (log f')'          (forward);      ȳ = (x̄ - ℓ̄ (log f')'(x))/f'(x)   (inverse, x = f⁻¹(y)).
    widths/heights/derivs: (dim, K) knot arrays as for `rqs`; x, out_bar: (dim, N).  numpy, float64."""
    W = np.asarray(widths, dtype=np.float64)
    H = np.asarray(heights, dtype=np.float64)
    D = np.asarray(derivs, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    dim, N = x.shape
    K = W.shape[1]
    g = np.asarray(out_bar, dtype=np.float64)
    lb = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    if inverse:
        xin, _ = rqs(W, H, D, np.asfortranarray(x), inverse=True)
        xin = np.asarray(xin, dtype=np.float64)
    else:
        xin = x
    out = np.empty_like(x)
    for i in range(dim):
        w, h, d = W[i], H[i], D[i]
        B = w[-1]
        for n in range(N):
            xv = xin[i, n]
            if not (-B < xv < B):                       # identity outside [-B, B] (:132)
                out[i, n] = g[i, n]
                continue
            k = int(np.searchsorted(w, xv, side="left")) - 1      # searchsortedfirst(widths, x) - 1 (:139)
            wk = -B if k < 0 else w[k]
            hk = -h[-1] if k < 0 else h[k]
            wd = w[k + 1] - wk
            dy = h[k + 1] - hk
            s = dy / wd
            dk = 1.0 if k < 0 else d[k]
            dk1 = 1.0 if k + 1 == K - 1 else d[k + 1]
            xi = (xv - wk) / wd
            p = xi * (1 - xi)
            ds = dk1 + dk - 2 * s
            den = s + ds * p
            nj = dk1 * xi * xi + 2 * s * p + dk * (1 - xi) ** 2
            J = s * s * nj / (den * den)
            dnj = 2 * dk1 * xi + 2 * s * (1 - 2 * xi) - 2 * dk * (1 - xi)
            dl = (dnj / nj - 2 * ds * (1 - 2 * xi) / den) / wd
            out[i, n] = g[i, n] * J + lb[n] * dl if not inverse else (g[i, n] - lb[n] * dl) / J
    return out


def rqs_vjp_knots(widths, heights, derivs, x, out_bar, ladj_bar=None, inverse=False):
    """Parameter side of `rqs_vjp`: cotangents of the knot arrays (dim, K), summed over the batch, of
    with_logabsdet_jacobian for the elementwise RationalQuadraticSpline (inverse=False: x = input, out_bar = ȳ) and its
    inverse (inverse=True: x = y, out_bar = x̄; implicit function theorem at f⁻¹(y)).  Closed-form partials of
    rational_quadratic_spline.jl:128-164 (value) and :266-297 (logjac); test infrastructure, pinned by finite differences of
    `rqs` in tests/test_oracle_golden.py.  The derivative at the last knot is not read by the spline (constant 1)."""
    W = np.asarray(widths, dtype=np.float64)
    H = np.asarray(heights, dtype=np.float64)
    D = np.asarray(derivs, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    dim, N = x.shape
    K = W.shape[1]
    g_all = np.asarray(out_bar, dtype=np.float64)
    lb_all = np.zeros(N) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=np.float64), (N,))
    if inverse:
        xin, _ = rqs(W, H, D, np.asfortranarray(x), inverse=True)
        xin = np.asarray(xin, dtype=np.float64)
    else:
        xin = x
    Wb, Hb, Db = np.zeros_like(W), np.zeros_like(H), np.zeros_like(D)
    for i in range(dim):
        w, h, d = W[i], H[i], D[i]
        B = w[-1]
        for n in range(N):
            xv = xin[i, n]
            if not (-B < xv < B):
                continue
            k = int(np.searchsorted(w, xv, side="left")) - 1      # bin between knots k and k+1 (0-based), k = -1: knot "0" = -knot K
            wk = -B if k < 0 else w[k]
            hk = -h[-1] if k < 0 else h[k]
            wd = w[k + 1] - wk
            dy = h[k + 1] - hk
            s = dy / wd
            dk = 1.0 if k < 0 else d[k]
            dk1 = 1.0 if k + 1 == K - 1 else d[k + 1]
            xi = (xv - wk) / wd
            p = xi * (1 - xi)
            om = 1 - 2 * xi
            ds = dk1 + dk - 2 * s
            den = s + ds * p
            M = dk1 * xi * xi + 2 * s * p + dk * (1 - xi) ** 2
            Nn = s * xi * xi + dk * p
            g, lb = g_all[i, n], lb_all[n]
            l_xi = (2 * dk1 * xi + 2 * s * om - 2 * dk * (1 - xi)) / M - 2 * ds * om / den
            if inverse:
                J = s * s * M / (den * den)
                g = -(g - lb * l_xi / wd) / J
                lb = -lb
            y_xi = dy * ((2 * s * xi + dk * om) * den - Nn * ds * om) / den ** 2
            y_s = dy * (xi * xi * den - Nn * (1 - 2 * p)) / den ** 2
            y_dh = Nn / den
            y_dk = dy * p * (den - Nn) / den ** 2
            y_dk1 = -dy * Nn * p / den ** 2
            l_s = 2 / s + 2 * p / M - 2 * (1 - 2 * p) / den
            l_dk = (1 - xi) ** 2 / M - 2 * p / den
            l_dk1 = xi * xi / M - 2 * p / den
            Gxi, Gs, Gdh = g * y_xi + lb * l_xi, g * y_s + lb * l_s, g * y_dh
            gw_k, gw_k1 = (Gxi * (xi - 1) + Gs * s) / wd, -(Gxi * xi + Gs * s) / wd
            gh_k, gh_k1 = g - Gdh - Gs / wd, Gdh + Gs / wd
            if k < 0:
                Wb[i, K - 1] -= gw_k
                Hb[i, K - 1] -= gh_k
            else:
                Wb[i, k] += gw_k
                Hb[i, k] += gh_k
                Db[i, k] += g * y_dk + lb * l_dk
            Wb[i, k + 1] += gw_k1
            Hb[i, k + 1] += gh_k1
            if k + 1 != K - 1:
                Db[i, k + 1] += g * y_dk1 + lb * l_dk1
    return Wb, Hb, Db


def rqs_params_vjp(raw_w, raw_h, raw_d, B, w_bar, h_bar, d_bar):
    """Pullback of `rqs_params` (the B constructor, rational_quadratic_spline.jl:109-123): knot cotangents (dim, K+1) ->
    cotangents of the raw parameters (dim, K), (dim, K), (dim, K-1).  numpy, float64; pinned by finite differences."""
    outs = []
    for raw, cb in ((raw_w, w_bar), (raw_h, h_bar)):
        r = np.asarray(raw, dtype=np.float64)
        cb = np.asarray(cb, dtype=np.float64)
        e = np.exp(r - r.max(axis=1, keepdims=True))
        p = e / e.sum(axis=1, keepdims=True)
        pbar = 2 * B * np.cumsum(cb[:, 1:][:, ::-1], axis=1)[:, ::-1]      # p̄_i = 2B Σ_{j>=i} c̄_{j+1}
        outs.append(p * (pbar - (p * pbar).sum(axis=1, keepdims=True)))
    rd = np.asarray(raw_d, dtype=np.float64)
    outs.append(np.asarray(d_bar, dtype=np.float64)[:, 1:-1] / (1 + np.exp(-rd)))
    return tuple(outs)


# ------------------------------------------------------------------ pullbacks of the matrix-variate bijectors (SURVEY.md §8f f-1 x f-4)
def _free_matrix(kind, y, K):
    """Unconstrained side -> (K, K, N) matrix of free parameters: corr kinds: strict upper triangle Y[i, j], i < j (VecCorr packing:
    column-major strict upper, src/utils.jl:99-108); pd kinds: lower triangle with the log-diagonal (PDVec packing:
    triu_to_vec(Y'), pd.jl:41)."""
    y = np.asarray(y)
    if kind in ("corr", "pd"):
        return np.array(y, copy=True)
    N = y.shape[1]
    Y = np.zeros((K, K, N), dtype=y.dtype)
    idx = 0
    for j in range(K):
        for i in range(j if kind == "vec_corr" else j + 1):
            if kind == "vec_corr":
                Y[i, j] = y[idx]
            else:
                Y[j, i] = y[idx]          # (Y')[i, j] = Y[j, i]
            idx += 1
    return Y


def _pack_free(kind, G, K):
    if kind == "corr":
        return G * np.triu(np.ones((K, K), bool), 1)[:, :, None]
    if kind == "pd":
        return G * np.tril(np.ones((K, K), bool))[:, :, None]
    rows = []
    for j in range(K):
        for i in range(j if kind == "vec_corr" else j + 1):
            rows.append(G[i, j] if kind == "vec_corr" else G[j, i])
    return np.stack(rows, axis=0) if rows else np.zeros((0, G.shape[2]), dtype=G.dtype)


def _chol_lower_batch(A):
    """lower Cholesky factor of the (K, K, N) symmetric matrices"""
    return np.transpose(np.linalg.cholesky(np.transpose(A, (2, 0, 1))), (1, 2, 0))


def _chol_reverse(L, Lb):
    """Reverse mode of the lower Cholesky factorisation A = L L' READ FROM ONE TRIANGLE (entries A[i][j], j <= i, each read once —
    cholesky(Hermitian(X, uplo)), src/utils.jl:37,50): cotangent of those entries given the cotangent Lb of the factor.  Unblocked,
    column by column from the last (forward: s_jj = A_jj - Σ_m L_jm², L_jj = √s_jj; s_ij = A_ij - Σ_m L_im L_jm, L_ij = s_ij / L_jj)."""
    K = L.shape[0]
    Lb = np.array(Lb, copy=True)
    Ab = np.zeros_like(L)
    for j in range(K - 1, -1, -1):
        for i in range(j + 1, K):
            Lb[j, j] = Lb[j, j] - Lb[i, j] * L[i, j] / L[j, j]
        sjj = Lb[j, j] / (2 * L[j, j])
        Ab[j, j] = sjj
        for m in range(j):
            Lb[j, m] = Lb[j, m] - 2 * sjj * L[j, m]
        for i in range(j + 1, K):
            sij = Lb[i, j] / L[j, j]
            Ab[i, j] = sij
            for m in range(j):
                Lb[i, m] = Lb[i, m] - sij * L[j, m]
                Lb[j, m] = Lb[j, m] - sij * L[i, m]
    return Ab


def matrix_bijector_vjp(kind, inp, out_bar, ladj_bar=None, inverse=False):
    """Pullback of with_logabsdet_jacobian for VecCorrBijector / CorrBijector / PDBijector / PDVecBijector (forward) and their
    inverses, a batch along the LAST axis: in_bar = J(inp)' out_bar + ladj_bar ∇ logabsdetjac(inp).

    inverse=True (unconstrained -> matrix, what a log-density evaluation differentiates): the rules the reference ships, chained —
    pd_from_upper / pd_from_lower (ext/BijectorsChainRulesCoreExt.jl:324-331, ext/BijectorsReverseDiffExt.jl:160-168: the factor's
    cotangent is the triangle of (X̄ + X̄') L), then replace_diag(exp) (ext/BijectorsReverseDiffExt.jl:153-158) + the log-det
    weights of pd.jl:27-31, or the reverse sweep of _inv_link_chol_lkj (corr.jl:402-451, with (1 - z²)·exp(log_remainder) for its
    (inv(z) - z)·W) + the (K - j) log U[j,j] terms of corr.jl:77-79.
    inverse=False (matrix -> unconstrained): cotangent of the factor from the link (corr.jl:299-335 differentiated: asinh(w/√R),
    logcosh = ½ log(1 + w²/R), R the running remainder; atanh on the first row of the vector form, :322; pd.jl:11: replace_diag(log)
    and the weights of :27-31), then the reverse of cholesky(Hermitian(X)) — the cotangent lands on the triangle the reference
    READS (upper for the correlation bijectors, src/utils.jl:50; lower for PD, :37), the other triangle gets zeros."""
    inp = np.asarray(inp)
    dt = inp.dtype
    corr_kind = kind in ("vec_corr", "corr")
    if inverse:
        K = out_bar.shape[0]
        N = out_bar.shape[2]
        Y = _free_matrix(kind, inp, K)
    else:
        K, N = inp.shape[0], inp.shape[2]
    dl = np.zeros(N, dtype=dt) if ladj_bar is None else np.broadcast_to(np.asarray(ladj_bar, dtype=dt), (N,))
    if inverse:
        Xb = np.asarray(out_bar, dtype=dt)
        S = Xb + np.transpose(Xb, (1, 0, 2))
        L = np.zeros((K, K, N), dtype=dt)
        if corr_kind:
            z, lc = np.tanh(Y), np.abs(Y) + np.log1p(np.exp(-2 * np.abs(Y))) - np.log(2.0)
            E = np.zeros((K, K, N), dtype=dt)
            for c in range(K):
                lr = np.zeros(N, dtype=dt)
                for i in range(c):
                    E[i, c] = np.exp(lr)
                    L[c, i] = z[i, c] * E[i, c]
                    lr = lr - lc[i, c]
                L[c, c] = np.exp(lr)
        else:
            L = Y * np.tril(np.ones((K, K), bool), -1)[:, :, None]
            for i in range(K):
                L[i, i] = np.exp(Y[i, i])
        Lb = np.einsum("imn,mjn->ijn", S, L) * np.tril(np.ones((K, K), bool))[:, :, None]
        G = np.zeros((K, K, N), dtype=dt)
        if corr_kind:
            for c in range(K):
                dlr = L[c, c] * Lb[c, c] + 2 * dl + ((K - 1 - c) * dl if 1 <= c <= K - 2 else 0.0)
                for i in range(c - 1, -1, -1):
                    G[i, c] = (1 - z[i, c] ** 2) * E[i, c] * Lb[c, i] - z[i, c] * dlr
                    dlr = dlr + dl + L[c, i] * Lb[c, i]
        else:
            G = Lb * np.tril(np.ones((K, K), bool), -1)[:, :, None]
            for i in range(K):
                G[i, i] = Lb[i, i] * L[i, i] + dl * (K + 1 - i)
        return _pack_free(kind, G, K)
    X = inp
    Gy = _free_matrix(kind, np.asarray(out_bar, dtype=dt), K)           # cotangent of the free parameters, same arrangement
    if corr_kind:
        A = np.triu(np.ones((K, K), bool))[:, :, None] * X
        A = A + np.transpose(A, (1, 0, 2)) - np.eye(K)[:, :, None] * X     # Hermitian(X): the upper triangle mirrored
    else:
        A = np.tril(np.ones((K, K), bool))[:, :, None] * X
        A = A + np.transpose(A, (1, 0, 2)) - np.eye(K)[:, :, None] * X
    L = _chol_lower_batch(A)
    Lb = np.zeros((K, K, N), dtype=dt)
    if corr_kind:
        for c in range(K):
            # row c of L = column c of U: w_i = L[c, i] (i < c), d = L[c, c]; R_i = d² + Σ_{m > i} w_m²
            R = np.zeros((max(c, 1), N), dtype=dt)
            rem = L[c, c] ** 2
            for i in range(c - 1, -1, -1):
                R[i] = rem
                rem = rem + L[c, i] ** 2
            Gsum = np.zeros(N, dtype=dt)                                  # Σ_{i < m} ∂F/∂R_i
            for m in range(c):
                w, yb, wt = L[c, m], Gy[m, c], (K - m) * dl
                if kind == "vec_corr" and m == 0:                        # y = atanh(w), logcosh = -½ log(1 - w²): no remainder
                    Lb[c, m] = (yb + wt * w) / (1 - w * w)
                    continue
                S2 = R[m] + w * w
                Sq = np.sqrt(S2)
                Lb[c, m] = yb / Sq + wt * w / S2 + 2 * w * Gsum
                Gsum = Gsum + yb * (-w / (2 * R[m] * Sq)) + wt * (-(w * w) / (2 * R[m] * S2))
            Lb[c, c] = 2 * L[c, c] * Gsum
    else:
        Lb = Gy * np.tril(np.ones((K, K), bool), -1)[:, :, None]
        for i in range(K):
            Lb[i, i] = (Gy[i, i] - dl * (K + 1 - i)) / L[i, i]
    Ab = _chol_reverse(L, Lb)                                             # cotangent of A[i][j], j <= i
    return np.transpose(Ab, (1, 0, 2)) if corr_kind else Ab               # corr kinds read X[j, i] (upper); PD reads X[i, j] (lower)
