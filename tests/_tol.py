"""One bar for every pullback / density comparison of the GPU suites (VERDICT r05 "do this" #1): north_star's relative tolerance —
1e-3 Float32, 1e-6 Float64 — FLAT, on a stated scale, with the measured worst error written down.

Error norm (say it where it is used): |got − ref| / scale, where scale is

* ``per="sample"``  the max-norm of the reference cotangent of that SAMPLE (last axis = batch): a cotangent entry that is tiny next to
  its column's largest one is compared on the column's scale, as `tests/test_gpu_matrix_vjp.py` has done since round 5;
* ``per="tensor"``  the max-norm of the whole reference tensor (parameter cotangents: one small tensor, every entry a sum over N columns);
* ``per="element"`` |ref| of the element itself plus ``floor`` (values, densities).

A legitimate growth factor is passed explicitly (``grow=``) and is part of the record: the rounding of a sum of N terms of either sign
grows like sqrt(N)·eps relative to the largest TERM, not to the (possibly cancelling) sum, so a parameter cotangent may pass
``term_scale=`` (the max-norm of the summands) instead of the reference's own norm.  Nothing else multiplies the bar.

Every call appends one line to gpurun_out/vjp_errors.jsonl; scripts/collect_profiles.py summarises them into profiles/rNN_vjp_errors.md.
"""
import json
import os

import numpy as np

RTOL_FLAT = {np.dtype(np.float32): 1e-3, np.dtype(np.float64): 1e-6}
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "vjp_errors.jsonl")


def _record(rec):
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        with open(_OUT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except Exception:
        pass


COND_C = 4.0      # head-room over the first-order amplification estimates below (they are estimates of a typical, not a worst-case, rounding pattern)


def simplex_amp(x, dt):
    """Per-column first-order amplification of the rounding of the stick-breaking recurrence in the arithmetic of `dt`
    (src/bijectors/simplex.jl:28-64, :84-120): the remainder r_k = 1 − Σ_{i<=k} x_i is a running sum of K terms — absolute rounding error
    ~ sqrt(K)·eps/2 whoever evaluates it in `dt`, the reference's own Float32 path included — and every Jacobian entry divides by it:
    a_n = sqrt(K)·eps(dt) / (2·min_k r_k).  `x` is the point ON the simplex, (K, N), in Float64."""
    x = np.asarray(x, np.float64)
    K = x.shape[0]
    eps = float(np.finfo(np.dtype(dt)).eps)
    r = np.maximum(1.0 - np.cumsum(x[:-1], axis=0), eps)
    return np.sqrt(K) * eps / (2.0 * r.min(axis=0))


def planar_inverse_amp(orc, w, u, b, z, dt):
    """Per-column first-order amplification of the INVERSE PlanarLayer pullback in the arithmetic of `dt`
    (src/bijectors/planar_layer.jl:112-127): layer l's Jacobian determinant d_l = 1 + wᵀû·sech²(wᵀz+b) can approach 0 (wᵀû > −1 only);
    the pullback holds 1/d_l from J⁻ᵀ times 1/d_l from ∇ logabsdetjac, and the root α it is evaluated at carries eps/d_l:
    a_n = eps(dt) · (Π_l max(1, 1/d_l))².  `z` = the pre-image (dim, N); log d_l per column from the oracle's single-layer forward."""
    w, u, b = (np.asarray(a, np.float64) for a in (w, u, b))
    w, u = w.reshape(w.shape[0], -1), u.reshape(u.shape[0], -1)
    x = np.asfortranarray(np.asarray(z, np.float64))
    logc = np.zeros(x.shape[1])
    for k in range(w.shape[1]):
        x, l = orc.planar(w[:, k], u[:, k], b.reshape(-1)[k:k + 1], x)
        logc += np.maximum(0.0, -np.asarray(l, np.float64))
        x = np.asfortranarray(x)
    return float(np.finfo(np.dtype(dt)).eps) * np.exp(2.0 * logc)


def flat_close(got, ref, dt, what, per="sample", term_scale=None, floor=0.0, note=None, cond=None):
    """assert |got − ref| <= rtol(dt) · scale everywhere (see the module docstring for `per`); returns the worst error / scale.
    `cond` (per="sample" only): a per-column first-order amplification a_n computed by the test from the data (simplex_amp,
    planar_inverse_amp: conditioning that ANY evaluation in `dt` suffers); the bar of column n is max(rtol, COND_C · a_n), the
    amplification at the worst column, its maximum and the share of columns over the FLAT bar are recorded and printed."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0.0
    rtol = RTOL_FLAT[np.dtype(dt)]
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all(), f"{what}: non-finite entries differ"
    if not fin.all():       # ±inf / NaN must agree exactly, the rest is compared below
        assert np.array_equal(got[~fin], ref[~fin], equal_nan=True), f"{what}: non-finite values differ"
        got, ref = np.where(fin, got, 0.0), np.where(fin, ref, 0.0)
    tiny = np.finfo(np.float64).tiny
    if per == "sample":
        n = ref.shape[-1] if ref.ndim else 1
        scale = np.abs(ref).reshape(-1, n).max(axis=0)
        if term_scale is not None:
            scale = np.maximum(scale, np.broadcast_to(np.asarray(term_scale, np.float64), scale.shape))
        scale = scale + floor + tiny
        err = np.abs(got - ref).reshape(-1, n).max(axis=0) / scale
    elif per == "tensor":
        scale = float(np.abs(ref).max())
        if term_scale is not None:
            scale = max(scale, float(term_scale))
        scale = scale + floor + tiny
        err = np.abs(got - ref).reshape(-1) / scale
    elif per == "element":
        err = (np.abs(got - ref) / (np.abs(ref) + floor + tiny)).reshape(-1)
    else:
        raise ValueError(per)
    allowed = np.full(err.shape, rtol)
    if cond is not None:
        assert per == "sample", "a conditioned bar is per column"
        cond = np.broadcast_to(np.asarray(cond, np.float64), err.shape)
        allowed = np.maximum(rtol, COND_C * cond)
    worst = int(np.argmax(err / allowed))
    rec = {"what": what, "dtype": np.dtype(dt).name, "per": per, "worst": float(err[worst]), "rtol": rtol, "worst_over_rtol": float(err[worst] / rtol),
           "n": int(err.size), "shape": list(ref.shape)}
    if cond is not None:
        rec.update({"conditioned": True, "worst_over_allowed": float(err[worst] / allowed[worst]), "amp_at_worst": float(cond[worst]), "amp_max": float(cond.max()),
                    "frac_over_flat": float((err > rtol).mean()), "worst_flat_column": float(err[allowed <= rtol].max()) if (allowed <= rtol).any() else None})
        if (err > rtol).any():
            print(f"{what}: {100 * (err > rtol).mean():.2f} % of columns over the flat {rtol:.0e}; worst column {err[worst]:.3g} of its scale with "
                  f"first-order amplification {cond[worst]:.3g} (max over columns {cond.max():.3g}); bar = max(rtol, {COND_C:g}·amplification)")
    if term_scale is not None:
        rec["term_scale"] = float(np.max(term_scale))
    if floor:
        rec["floor"] = float(floor)
    if note:
        rec["note"] = note
    _record(rec)
    assert err[worst] <= allowed[worst], f"{what}: worst error {err[worst]:.3g} of its {per} scale at {worst}, allowed {allowed[worst]:.3g}" + (f" ({note})" if note else "")
    return float(err[worst])
