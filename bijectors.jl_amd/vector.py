"""the VectorBijectors homogeneous-product path, batched over chains.

`product_distribution(fill(d, size...))` links every component with the SAME scalar (or per-slice vector)
transform (src/vector/product/fill.jl:111-219); DynamicPPL calls it on every log-density evaluation.  With one
column per chain a (prod(size), n_chains) parameter matrix is one instance of the elementwise / Simplex kernels:
the log-det of a chain is the per-column log-det.  Host mirror of

    src/vector/univariate/positive.jl:11-50     Exp(bound, sign), Log(bound, sign)
    src/vector/univariate/truncated.jl:17-103   Truncate(a, b), Untruncate(a, b)
    src/vector/multivariate/mvlognormal.jl      MapExp / MapLog
    src/vector/multivariate/simplex.jl          SimplexBijector for Dirichlet-like slices
    src/vector/product/fill.jl:111-219          ProductVecTransform / ProductVecInvTransform

No arithmetic happens here: every transform is an op list of `bjx_chain` or a call of `bjx_simplex`.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch

from . import _lib as L
from . import interface as I

__all__ = ["Exp", "Log", "Truncate", "Untruncate", "TypedIdentity", "scalar_to_scalar_bijector", "ProductVecTransform",
           "ProductVecInvTransform", "to_linked_vec", "from_linked_vec", "to_linked_vec_product", "from_linked_vec_product",
           "MapLog", "MapExp", "JointOrderWrap", "InverseJointOrderWrap", "is_monotonically_decreasing"]


class ScalarToScalarBijector(I.Bijector):
    """A scalar map applied to every element; evaluated by the fused chain kernel (alone, inside a composition, or as a segment of a
    `Stacked`: interface._stage_ops reads `_elementwise_stage`)."""

    _elementwise_stage = True

    def _ops(self, inv=False):
        raise NotImplementedError

    def _wlj(self, x, per_sample, want_ladj=True):
        return I._run_chain(self._ops(False), x, per_sample, want_ladj)

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return I._run_chain(self._ops(True), x, per_sample, want_ladj)


class TypedIdentity(ScalarToScalarBijector):
    def _key(self):
        return ()

    def _ops(self, inv=False):
        return [(L.OP_IDENTITY, None, None)]


class Exp(ScalarToScalarBijector):
    """positive.jl:11-25: y -> sign * exp(y) + bound, log-det = y."""

    def __init__(self, bound=0.0, sign: int = 1):
        self.bound, self.sign = float(bound), int(sign)

    def _key(self):
        return (self.bound, self.sign)

    def _ops(self, inv=False):
        if inv:
            return Log(self.bound, self.sign)._ops(False)
        ops = [(L.OP_EXP, None, None)]
        if self.sign < 0:
            ops.append((L.OP_SIGNFLIP, None, None))
        if self.bound != 0.0:
            ops.append((L.OP_SHIFT, self.bound, None))
        return ops


class Log(ScalarToScalarBijector):
    """positive.jl:27-50: x -> log(sign * (x - bound)), log-det = -log(sign * (x - bound))."""

    def __init__(self, bound=0.0, sign: int = 1):
        self.bound, self.sign = float(bound), int(sign)

    def _key(self):
        return (self.bound, self.sign)

    def _ops(self, inv=False):
        if inv:
            return Exp(self.bound, self.sign)._ops(False)
        ops = []
        if self.bound != 0.0:
            ops.append((L.OP_SHIFT, -self.bound, None))
        if self.sign < 0:
            ops.append((L.OP_SIGNFLIP, None, None))
        ops.append((L.OP_LOG, None, None))
        return ops


class Truncate(ScalarToScalarBijector):
    """truncated.jl:17-56: (-inf, inf) -> (a, b); the four branches on the finiteness of the bounds are the ones of
    Inverse{TruncatedBijector} (its closed-form log-det log(b-a) - |y| - 2 log1pexp(-|y|) equals
    log(b-a) + y - 2 log1pexp(y), :44-48)."""

    def __init__(self, lower, upper):
        self.lower, self.upper = float(lower), float(upper)

    def _key(self):
        return (self.lower, self.upper)

    def _ops(self, inv=False):
        return [(L.OP_TRUNCATED if inv else L.OP_TRUNCATED_INV, self.lower, self.upper)]


class Untruncate(ScalarToScalarBijector):
    """truncated.jl:59-103: (a, b) -> (-inf, inf)."""

    def __init__(self, lower, upper):
        self.lower, self.upper = float(lower), float(upper)

    def _key(self):
        return (self.lower, self.upper)

    def _ops(self, inv=False):
        return [(L.OP_TRUNCATED_INV if inv else L.OP_TRUNCATED, self.lower, self.upper)]


class MapLog(ScalarToScalarBijector):
    """src/vector/multivariate/mvlognormal.jl:1-8 (to_linked_vec of an MvLogNormal): log of every element, log-det −Σ log x."""

    def _key(self):
        return ()

    def _ops(self, inv=False):
        return [(L.OP_EXP if inv else L.OP_LOG, None, None)]


class MapExp(ScalarToScalarBijector):
    """mvlognormal.jl:9-15 (from_linked_vec): exp of every element, log-det Σ x."""

    def _key(self):
        return ()

    def _ops(self, inv=False):
        return [(L.OP_LOG if inv else L.OP_EXP, None, None)]


def is_monotonically_decreasing(t) -> bool:
    """Bijectors.is_monotonically_decreasing for the scalar links (truncated.jl:21-22, 72-73; positive.jl: sign = -1; common.jl:29)."""
    if isinstance(t, (Truncate, Untruncate)):
        return math.isinf(t.lower) and not math.isinf(t.upper)
    if isinstance(t, (Exp, Log)):
        return t.sign < 0
    return False


class JointOrderWrap(I.Bijector):
    """src/vector/order/order.jl:14-46 (to_linked_vec of JointOrderStatistics): the parent's scalar link over every element of the
    ORDERED sample (the sign flipped back when the link is decreasing), then ordered -> unordered: y₁, log(yᵢ − yᵢ₋₁) — which is
    inverse(OrderedBijector) (ordered.jl:50-80).  One bjx_chain launch + one bjx_ordered launch; per-column log-det for a matrix of chains."""

    def __init__(self, bijector):
        self.bijector = bijector

    def _key(self):
        return (self.bijector,)

    def _wlj(self, x, per_sample, want_ladj=True):
        ops = list(self.bijector._ops(False)) + ([(L.OP_SIGNFLIP, None, None)] if is_monotonically_decreasing(self.bijector) else [])
        y, l1 = I._run_chain(ops, x, True if want_ladj else False, want_ladj)
        z, l2 = I.inverse(I.OrderedBijector())._wlj(y, per_sample=True, want_ladj=want_ladj)
        if not want_ladj:
            return z, None
        l = l1 + l2
        return z, (l if (per_sample or x.dim() == 2) else l.reshape(()))

    def _wlj_inv(self, y, per_sample, want_ladj=True):
        return InverseJointOrderWrap(_inverse_scalar(self.bijector))._wlj(y, per_sample, want_ladj)


class InverseJointOrderWrap(I.Bijector):
    """order.jl:48-76 (from_linked_vec): unordered -> ordered (xᵢ = exp(yᵢ) + xᵢ₋₁ = OrderedBijector), the sign flip, then the
    inverse link over every element."""

    def __init__(self, bijector):
        self.bijector = bijector

    def _key(self):
        return (self.bijector,)

    def _wlj(self, y, per_sample, want_ladj=True):
        x, l1 = I.OrderedBijector()._wlj(y, per_sample=True, want_ladj=want_ladj)
        ops = ([(L.OP_SIGNFLIP, None, None)] if is_monotonically_decreasing(self.bijector) else []) + list(self.bijector._ops(False))
        z, l2 = I._run_chain(ops, x, True if want_ladj else False, want_ladj)
        if not want_ladj:
            return z, None
        l = l1 + l2
        return z, (l if (per_sample or y.dim() == 2) else l.reshape(()))

    def _wlj_inv(self, x, per_sample, want_ladj=True):
        return JointOrderWrap(_inverse_scalar(self.bijector))._wlj(x, per_sample, want_ladj)


def _inverse_scalar(t):
    if isinstance(t, Exp):
        return Log(t.bound, t.sign)
    if isinstance(t, Log):
        return Exp(t.bound, t.sign)
    if isinstance(t, Truncate):
        return Untruncate(t.lower, t.upper)
    if isinstance(t, Untruncate):
        return Truncate(t.lower, t.upper)
    if isinstance(t, TypedIdentity):
        return t
    return I.inverse(t)


def scalar_to_scalar_bijector(minimum: float, maximum: float, positive_family: bool = False):
    """The link of a continuous univariate distribution from its support (truncated.jl:105-108; positive.jl:74 for
    the distributions whose support is [minimum, inf) by type)."""
    if positive_family:
        return Log(minimum, 1)
    if math.isinf(minimum) and math.isinf(maximum):
        return TypedIdentity()
    return Untruncate(minimum, maximum)


class ProductVecTransform(I.Transform):
    """fill.jl:111-159: `vec` of the transform applied to every component.  `base_size` = () for univariate
    components (one scalar map over all elements) or (K,) for vector components (e.g. Dirichlet: SimplexBijector on
    every length-K slice).  Batched: X is (prod(base_size) * prod(size), n_chains), one column per chain."""

    def __init__(self, trf, size: Sequence[int], base_size: Tuple[int, ...] = ()):
        self.trf, self.size, self.base_size = trf, tuple(int(s) for s in size), tuple(int(s) for s in base_size)

    def _key(self):
        return (self.trf, self.size, self.base_size)

    def _n_components(self):
        return int(math.prod(self.size)) if self.size else 1

    def _wlj(self, x, per_sample, want_ladj=True):
        return _product_apply(self.trf, self, x, per_sample, want_ladj)


class ProductVecInvTransform(ProductVecTransform):
    """fill.jl:161-219: the inverse (linked vector -> components)."""


def _product_apply(trf, t: ProductVecTransform, x, per_sample, want_ladj):
    vec = x.dim() == 1
    X = x.reshape(-1, 1) if vec else x
    if X.dim() != 2:
        raise ValueError("DimensionMismatch: expected the vectorised parameters, one column per chain")
    n_chains = X.shape[1]
    m = t._n_components()
    if not t.base_size:                                   # univariate components: one elementwise launch
        if X.shape[0] != m:
            raise ValueError(f"DimensionMismatch: expected {m} rows, got {X.shape[0]}")
        r = I._fast_chain(t, lambda: trf._ops(False), X, True) if want_ladj else None      # cached launch plan (include/bjx.h "plans")
        y, l = r if r is not None else I._run_chain(trf._ops(False), X, True if want_ladj else False, want_ladj)
    else:                                                 # vector components: slices become extra columns (zero-copy)
        k_in = X.shape[0] // m
        if k_in * m != X.shape[0]:
            raise ValueError(f"DimensionMismatch: {X.shape[0]} rows do not split into {m} components")
        Xc = I.colmajor(X)
        slices = Xc.T.reshape(n_chains * m, k_in).T       # (k_in, m * n_chains), column-major view of the same memory
        ys, ls = trf._wlj(slices, per_sample=True, want_ladj=want_ladj)
        k_out = ys.shape[0]
        y = ys.T.reshape(n_chains, m * k_out).T
        l = ls.reshape(n_chains, m).sum(dim=1) if want_ladj else None
    if vec:
        y = y.reshape(-1)
    if not want_ladj:
        return y, None
    if per_sample is True:
        return y, l
    return y, (l[0] if vec else l)                        # a vector input returns the scalar, like the reference


def to_linked_vec(trf, size: Sequence[int], base_size: Tuple[int, ...] = ()):
    """`to_linked_vec(product_distribution(fill(d, size...)))` given the component link `trf`
    (e.g. `scalar_to_scalar_bijector(0, 1)` for Beta, `SimplexBijector()` with base_size=(K,) for Dirichlet)."""
    return ProductVecTransform(trf, size, base_size)


def from_linked_vec(trf, size: Sequence[int], base_size: Tuple[int, ...] = ()):
    """`from_linked_vec(...)`: `trf` is the component LINK; its inverse is applied (fill.jl:161-219)."""
    inv = _inverse_scalar(trf) if isinstance(trf, ScalarToScalarBijector) else I.inverse(trf)
    return ProductVecInvTransform(inv, size, base_size)


def _stack(components, inverse_links: bool):
    """Heterogeneous product `product_distribution((a = d1, b = d2, ...))` (src/vector/interface.jl:86-129 examples):
    the linked vector is the concatenation of the components' linked vectors, i.e. a `Stacked` with one segment per
    component — ONE launch of bjx_stacked over all chains when every link is elementwise.
    components: [(link, n_rows)] with `link` a scalar link (applied to n_rows rows) in the order of the fields."""
    bs, ranges, lo = [], [], 1
    for link, n in components:
        t = link
        if inverse_links:
            t = _inverse_scalar(link) if isinstance(link, ScalarToScalarBijector) else I.inverse(link)
        bs.append(I.identity if isinstance(t, TypedIdentity) else t)
        ranges.append((lo, lo + int(n) - 1))
        lo += int(n)
    return I.Stacked(bs, ranges)


def to_linked_vec_product(components):
    return _stack(components, False)


def from_linked_vec_product(components):
    return _stack(components, True)
