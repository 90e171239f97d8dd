#!/usr/bin/env python3
"""bench.py — throughput of the batched transform + log-abs-det-Jacobian hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4|c5a|c5b] [--scaling weak|strong]
                    [--collective torch|bjx] [--no-rows]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = ONE pass of the hot path over one batch of synthetic input that is already resident in
HBM.  Default workload = BASELINE.json configs[1] ("c2"): with_logabsdet_jacobian of
exp ∘ Shift(b) ∘ Scale(a), Float32, dim = 64, batch = 2^24 PER GPU (weak scaling: every rank
owns an independent column block; the only collective is the 8-byte all-reduce of Σ logabsdetjac).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (algorithmic
bytes / HIP-event time of the dominant kernel vs the 8 TB/s HBM peak) and `cpu_baseline`
(the CPU oracle = a port of the reference algorithm, timed on the host cores of the same box on a
bounded sample of the same workload).

The same line carries `rows`: one measured sub-line per OTHER BASELINE.json config — c1 (configs[0], a Float64 vector
of 2^20 through elementwise(exp): a launch-latency case, with its CPU leg), c3, c4, c5a, c5b — each with `kernel`,
`kernel_ms`, `roofline.frac` and (N = 1) `cpu_baseline`, measured exactly like the headline (same barriers, hipEvent
pairs on the context stream), so every BASELINE config is witnessed by whoever runs this file.  With N > 1 ranks the
line also carries `strong_scaling`: configs[1] and configs[3] at their FIXED global batch split into contiguous column
blocks (`shard_columns`) — the "1 vs 8 GPUs on the sharded batch" reading of north_star — next to the weak-scaling
headline (`--scaling strong` makes the strong form the headline).
"""
import argparse
import ctypes as C
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:35)
METRIC = "M samples/sec for with_logabsdet_jacobian (named bijector, dim×batch); % HBM roofline"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="c2", choices=["c1", "c2", "c2v", "copy", "c3", "c4", "c5a", "c5b", "vcorr", "pdvec", "c2_f64", "c4_f64"])
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: the BASELINE batch PER GPU; strong: the BASELINE batch in total, split by shard_columns")
    p.add_argument("--collective", default="torch", choices=["torch", "bjx"],
                   help="who all-reduces the Float64 partial Σ logabsdetjac: torch.distributed (RCCL) or the library's own "
                        "communicator (bjx_comm_init + bjx_allreduce_sum_f64: the path a Julia host takes)")
    p.add_argument("--no-rows", action="store_true", help="only the headline workload (no per-config sub-lines)")
    p.add_argument("--rows", default="c1,c3,c3_uncached,c4,c5a,c5b,c2_f64,c4_f64")
    p.add_argument("--no-cache-params", action="store_true",
                   help="do not opt into bj.cache_params: tables derived from parameters (the spline's LDS blob, the layer-major Planar tables) are rebuilt on every call")
    p.add_argument("--log2-batch", type=int, default=None, help="override the batch (testing)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-log2-batch", type=int, default=None)
    return p.parse_args()


_ARENA_MB = int(os.environ.get("BJX_BENCH_ARENA_MB", "0"))     # experiment: carve the large buffers from ONE allocation
_arena = {}


def colmajor_empty(torch, rows, batch, dtype, device):
    n = rows * batch
    item = torch.empty((), dtype=dtype).element_size()
    if _ARENA_MB <= 0 or n * item < (8 << 20):
        return torch.empty((batch, rows), dtype=dtype, device=device).T
    if "buf" not in _arena:
        _arena["buf"] = torch.empty(_ARENA_MB << 20, dtype=torch.uint8, device=device)
        _arena["off"] = 0
    off = _arena["off"]
    _arena["off"] = off + (n * item + 255) // 256 * 256
    return _arena["buf"][off:off + n * item].view(dtype).view(batch, rows).T


def fill_normal(bj, torch, t, col0, seed, mean=0.0, std=1.0):
    """Counter-based N(mean, std) fill keyed by the GLOBAL element index -> shard-count invariant."""
    L = bj._lib
    ctx = bj.context(t.device)
    rows, batch = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.shape[0], 1)
    dt = L.BJX_F32 if t.dtype == torch.float32 else L.BJX_F64
    rc = L.load().bjx_fill_normal(ctx.h, dt, t.data_ptr(), rows, batch, col0, seed, mean, std)
    L.check(ctx.h, rc, "bjx_fill_normal")


def compose_planar(bj, w, u, b):
    """The flow written the way the reference writes it (docs/src/flows.md:115): one PlanarLayer object per layer, composed with
    `∘` (`@` here), l_nl ∘ … ∘ l_1 — NOT a stacked-parameter constructor.  The composition planner turns the run into one launch."""
    flow = None
    for k in range(w.shape[1]):
        layer = bj.PlanarLayer(w[:, k].contiguous(), u[:, k].contiguous(), b[k:k + 1].clone())
        flow = layer if flow is None else layer @ flow
    return flow


PREROLL_MS = float(os.environ.get("BJX_BENCH_PREROLL_MS", "60"))     # untimed clock-settling pre-roll before the timed steps (0 = off)

# ---------------------------------------------------------------------------------- workloads
DEFAULT_LOG2 = {"c1": 0, "c2": 24, "c2v": 24, "copy": 24, "c3": 22, "c4": 22, "c5a": 20, "c5b": 20, "vcorr": 18, "pdvec": 18, "c2_f64": 24, "c4_f64": 22}


def make_workload(name, bj, torch, device, rank, world, log2_batch, scaling="weak"):
    """-> dict(step=callable, samples=int on this rank, total=int over all ranks, bytes_per_sample, label, kernel, dtype, cfg)"""
    f32 = torch.float32
    lb = DEFAULT_LOG2[name] if log2_batch is None else log2_batch
    NG = 1 << lb                                   # the BASELINE batch
    if scaling == "strong":                        # fixed global batch, contiguous column blocks
        lo, hi = bj.shard.shard_columns(NG, world, rank)
        N, col0, total = hi - lo, lo, NG
        where = f"batch=2^{lb} split over {world} GPU(s)"
    else:
        N, col0, total = NG, rank * NG, NG * world
        where = f"batch=2^{lb}/GPU"
    sharded = bj.shard.with_logabsdet_jacobian_sharded
    if name == "c1":
        # configs[0]: elementwise(exp) on ONE Float64 vector of 2^20 — the reference's everyday call shape; on the GPU it
        # is a launch-latency case (16.8 MB of traffic), not a bandwidth one
        n = 1 << 20
        x = torch.empty(n, dtype=torch.float64, device=device)
        y = torch.empty_like(x)
        fill_normal(bj, torch, x, 0, seed=0)
        b = bj.elementwise(bj.exp)

        def step():
            return sharded(b, x, out=y, per_sample=False)[2]

        return dict(step=step, samples=1, total=world, elements=n, bytes_per_sample=16 * n, kernel="chain_flat_kernel<double>", dtype="f64",
                    label="with_logabsdet_jacobian(elementwise(exp)) Float64 vector of 2^20 (one call = one 'sample'; BASELINE configs[0])",
                    cfg={"workload": "elementwise(exp) + logabsdetjac on a Float64 Vector of 2^20 (BASELINE configs[0])", "length": n})
    if name in ("c2", "c2v", "copy"):
        dim = 64
        x = colmajor_empty(torch, dim, N, f32, device)
        y = colmajor_empty(torch, dim, N, f32, device)
        fill_normal(bj, torch, x, col0, seed=0)
        if name == "c2":
            b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
        elif name == "copy":  # streaming ceiling of the same kernel skeleton (one trivial stage)
            b = bj.Shift(0.0)
        else:  # per-row vector parameters
            b = bj.elementwise(bj.exp) @ bj.Shift(torch.full((dim,), 0.1, device=device)) @ bj.Scale(torch.linspace(0.5, 1.5, dim, device=device))

        def step():
            return sharded(b, x, out=y, per_sample=False)[2]

        return dict(step=step, samples=N, total=total, bytes_per_sample=2 * dim * 4, kernel="chain_flat_kernel", dtype="f32",
                    label=f"with_logabsdet_jacobian(exp∘Shift∘Scale) Float32 dim={dim} {where}",
                    cfg={"workload": "Composed(Shift,Scale,Exp) fused fwd+logabsdetjac (BASELINE configs[1])", "dim": dim,
                         "batch_per_gpu": N, "params": "scalar" if name == "c2" else "per-row vectors"})
    if name == "c3":
        dim, K = 32, 16
        x = colmajor_empty(torch, dim, N, f32, device)
        y = colmajor_empty(torch, dim, N, f32, device)
        xb = colmajor_empty(torch, dim, N, f32, device)
        fill_normal(bj, torch, x, col0, seed=0)
        raw = [colmajor_empty(torch, dim, k, f32, device) for k in (K, K, K - 1)]
        for i, r in enumerate(raw):
            fill_normal(bj, torch, r, 0, seed=100 + i)
        b = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
        ib = bj.inverse(b)

        def step():
            y_, _, s1 = sharded(b, x, out=y)
            _, _, s2 = sharded(ib, y_, out=xb)
            return s1

        return dict(step=step, samples=N, total=total, bytes_per_sample=(2 * dim * 4 + 4) * 2, kernel="rqs_lds_kernel (forward + inverse launch)", dtype="f32",
                    label=f"RationalQuadraticSpline K=16 fwd+inverse+logabsdetjac Float32 dim={dim} {where}",
                    cfg={"workload": "RationalQuadraticSpline K=16 fwd+inv+logabsdetjac (BASELINE configs[2])", "dim": dim, "batch_per_gpu": N})
    if name in ("c2_f64", "c4_f64"):
        # Float64 sub-lines (Float64 is Turing's default and the dtype of the reference's own tests): the C2 chain
        # and the C4 flow in Float64 at HALF the Float32 batch (the same bytes per launch)
        f64 = torch.float64
        N64 = max(1, N // 2)
        if name == "c2_f64":
            dim = 64
            x = colmajor_empty(torch, dim, N64, f64, device)
            y = colmajor_empty(torch, dim, N64, f64, device)
            fill_normal(bj, torch, x, col0 // 2, seed=0)
            b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)

            def step():
                return sharded(b, x, out=y, per_sample=False)[2]

            return dict(step=step, samples=N64, total=total // 2, bytes_per_sample=2 * dim * 8, kernel="chain_flat_kernel<double>", dtype="f64",
                        label=f"with_logabsdet_jacobian(exp∘Shift∘Scale) Float64 dim={dim} batch=2^{lb - 1}/GPU",
                        cfg={"workload": "Composed(Shift,Scale,Exp) fused fwd+logabsdetjac, Float64 (configs[1] in the reference's test dtype)", "dim": dim, "batch_per_gpu": N64})
        dim, nl = 128, 8
        x = colmajor_empty(torch, dim, N64, f64, device)
        y = colmajor_empty(torch, dim, N64, f64, device)
        fill_normal(bj, torch, x, col0 // 2, seed=0)
        w, u = colmajor_empty(torch, dim, nl, f64, device), colmajor_empty(torch, dim, nl, f64, device)
        bb = torch.empty(nl, dtype=f64, device=device)
        fill_normal(bj, torch, w, 0, seed=200, std=1.0 / math.sqrt(dim))
        fill_normal(bj, torch, u, 0, seed=201, std=1.0 / math.sqrt(dim))
        fill_normal(bj, torch, bb, 0, seed=202)
        flow = compose_planar(bj, w, u, bb)

        def step():
            return sharded(flow, x, out=y)[2]

        return dict(step=step, samples=N64, total=total // 2, bytes_per_sample=2 * dim * 8 + 8, kernel="planar_mfma64_kernel", dtype="f64",
                    label=f"8-layer PlanarLayer flow fused fwd+logabsdetjac Float64 dim={dim} batch=2^{lb - 1}/GPU",
                    cfg={"workload": "l8∘…∘l1 of PlanarLayers (planner-fused), Float64 (configs[3] in the reference's test dtype)", "dim": dim, "layers": nl, "batch_per_gpu": N64})
    if name == "c4":
        dim, nl = 128, 8
        x = colmajor_empty(torch, dim, N, f32, device)
        y = colmajor_empty(torch, dim, N, f32, device)
        fill_normal(bj, torch, x, col0, seed=0)
        w = colmajor_empty(torch, dim, nl, f32, device)
        u = colmajor_empty(torch, dim, nl, f32, device)
        bb = torch.empty(nl, dtype=f32, device=device)
        fill_normal(bj, torch, w, 0, seed=200, std=1.0 / math.sqrt(dim))
        fill_normal(bj, torch, u, 0, seed=201, std=1.0 / math.sqrt(dim))
        fill_normal(bj, torch, bb, 0, seed=202)
        flow = compose_planar(bj, w, u, bb)

        def step():
            return sharded(flow, x, out=y)[2]

        return dict(step=step, samples=N, total=total, bytes_per_sample=2 * dim * 4 + 4, kernel="planar_reg2_kernel", dtype="f32",
                    label=f"8-layer PlanarLayer flow l8∘…∘l1 (planner: one fused launch) fwd+logabsdetjac Float32 dim={dim} {where}",
                    cfg={"workload": "l8∘…∘l1 of PlanarLayers, fused by the composition planner (BASELINE configs[3])", "dim": dim, "layers": nl, "batch_per_gpu": N})
    if name == "c5a":
        K = 64
        x = colmajor_empty(torch, K, N, f32, device)
        fill_normal(bj, torch, x, col0, seed=0)
        x = torch.softmax(x.T, dim=1).T  # synthetic-input preparation (outside the timed region)
        b = bj.SimplexBijector()

        def step():
            return sharded(b, x)[2]

        return dict(step=step, samples=N, total=total, bytes_per_sample=K * 4 + (K - 1) * 4 + 4, kernel="quad_stream_kernel<QSimplexFwd>", dtype="f32",
                    label=f"SimplexBijector fwd+logabsdetjac Float32 K={K} {where}",
                    cfg={"workload": "SimplexBijector (BASELINE configs[4], first half)", "K": K, "batch_per_gpu": N})
    if name == "c5b":
        K = 64   # BASELINE configs[4]: batch 2^20 (8.5 GB of y + 17 GB of dense W per GPU)
        n = K * (K - 1) // 2
        yv = colmajor_empty(torch, n, N, f32, device)
        fill_normal(bj, torch, yv, col0, seed=0, std=0.5)
        ib = bj.inverse(bj.VecCholeskyBijector("U"))

        def step():
            return sharded(ib, yv)[2]

        return dict(step=step, samples=N, total=total, bytes_per_sample=n * 4 + K * K * 4 + 4, kernel="chol_inv_chunk_kernel", dtype="f32",
                    label=f"inverse VecCholeskyBijector (y->W dense + logJ) Float32 K={K} {where}",
                    cfg={"workload": "VecCholeskyBijector inverse, dense W (BASELINE configs[4], second half)", "K": K, "batch_per_gpu": N})
    if name in ("vcorr", "pdvec"):
        # the matrix-variate constraint bijectors that need a per-sample Cholesky factorisation
        # (corr.jl:128-162, pd.jl:34-60).  Input = K x K correlation / covariance matrices X = U'U built on the device.
        K = 32
        n = K * (K - 1) // 2 if name == "vcorr" else K * (K + 1) // 2
        yv = colmajor_empty(torch, n, N, f32, device)
        fill_normal(bj, torch, yv, col0, seed=0, std=0.3)
        b = bj.VecCorrBijector() if name == "vcorr" else bj.PDVecBijector()
        X = bj.transform(bj.inverse(b), yv)        # preparation, outside the timed region
        del yv

        def step():
            return sharded(b, X)[2]

        return dict(step=step, samples=N, total=total, bytes_per_sample=K * K * 4 + n * 4 + 4, kernel="chol_link_kernel", dtype="f32",
                    label=f"{'VecCorrBijector' if name == 'vcorr' else 'PDVecBijector'} (X -> Cholesky -> y + logabsdetjac) Float32 K={K} {where}",
                    cfg={"workload": f"{'VecCorrBijector' if name == 'vcorr' else 'PDVecBijector'} forward (SURVEY.md §8f-4)", "K": K, "batch_per_gpu": N})
    raise ValueError(name)


# ---------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(name, log2_batch, budget=8.0, variants=True):
    """The oracle (a C++ port of the reference algorithm, reference-structured: one allocating pass
    per composed stage, single thread — the reference is single-threaded) on a bounded sample."""
    import numpy as np

    from oracle import oracle as orc

    rng = np.random.default_rng(0)
    if name == "copy":
        name = "c2"
    unit = "M samples/s"
    if name == "c1":
        n = 1 << 20
        N = 1
        x = rng.standard_normal(n)
        ops = [(orc.OP_EXP, None, None)]
        fn = lambda: orc.chain(ops, x)
        sample = "oracle elementwise(exp) + Σ log-det on the whole Float64 vector of 2^20 (1 thread)"
    elif name in ("c2", "c2v"):
        lb = 20 if log2_batch is None else log2_batch
        N, dim = 1 << lb, 64
        x = np.asfortranarray(rng.standard_normal((dim, N), dtype=np.float32))
        if name == "c2":
            ops = [(orc.OP_SCALE, 0.5, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)]
        else:
            ops = [(orc.OP_SCALE, np.linspace(0.5, 1.5, dim), None), (orc.OP_SHIFT, np.full(dim, 0.1), None), (orc.OP_EXP, None, None)]
        fn = lambda: orc.chain(ops, x)
        sample = f"oracle chain (3 allocating passes, 1 thread) on Float32 64 x 2^{lb}"
    elif name == "c2_f64":
        lb = 19 if log2_batch is None else log2_batch
        N, dim = 1 << lb, 64
        x = np.asfortranarray(rng.standard_normal((dim, N)))
        ops = [(orc.OP_SCALE, 0.5, None), (orc.OP_SHIFT, 0.1, None), (orc.OP_EXP, None, None)]
        fn = lambda: orc.chain(ops, x)
        sample = f"oracle chain (3 allocating passes, 1 thread) on Float64 64 x 2^{lb}"
    elif name == "c4_f64":
        lb = 15 if log2_batch is None else log2_batch
        N, dim, nl = 1 << lb, 128, 8
        w, u, b = rng.standard_normal((dim, nl)) / math.sqrt(dim), rng.standard_normal((dim, nl)) / math.sqrt(dim), rng.standard_normal(nl)
        x = np.asfortranarray(rng.standard_normal((dim, N)))
        fn = lambda: orc.planar(w, u, b, x)
        sample = f"oracle 8 Planar layers (layer-by-layer passes, 1 thread) on Float64 128 x 2^{lb}"
    elif name == "c3":
        lb = 16 if log2_batch is None else log2_batch
        N, dim, K = 1 << lb, 32, 16
        w, h, d = orc.rqs_params(rng.standard_normal((dim, K), dtype=np.float32), rng.standard_normal((dim, K), dtype=np.float32), rng.standard_normal((dim, K - 1), dtype=np.float32), 3.0)
        x = np.asfortranarray(rng.standard_normal((dim, N), dtype=np.float32))

        def fn():
            y, _ = orc.rqs(w, h, d, x)
            orc.rqs(w, h, d, y, inverse=True)

        sample = f"oracle RQS fwd+inv (per-element knot search, 1 thread) on Float32 32 x 2^{lb}"
    elif name == "c4":
        lb = 16 if log2_batch is None else log2_batch
        N, dim, nl = 1 << lb, 128, 8
        w = (rng.standard_normal((dim, nl)) / math.sqrt(dim)).astype(np.float32)
        u = (rng.standard_normal((dim, nl)) / math.sqrt(dim)).astype(np.float32)
        b = rng.standard_normal(nl).astype(np.float32)
        x = np.asfortranarray(rng.standard_normal((dim, N), dtype=np.float32))
        fn = lambda: orc.planar(w, u, b, x)
        sample = f"oracle 8 Planar layers (layer-by-layer passes, 1 thread) on Float32 128 x 2^{lb}"
    elif name == "c5a":
        lb = 18 if log2_batch is None else log2_batch
        N, K = 1 << lb, 64
        x = np.asfortranarray(rng.dirichlet(np.ones(K), size=N).T.astype(np.float32))
        fn = lambda: orc.simplex(x)
        sample = f"oracle Simplex transform + logabsdetjac (2 passes, 1 thread) on Float32 64 x 2^{lb}"
    elif name in ("vcorr", "pdvec"):
        lb = 12 if log2_batch is None else log2_batch
        N, K = 1 << lb, 32
        n = K * (K - 1) // 2 if name == "vcorr" else K * (K + 1) // 2
        y = np.asfortranarray((0.3 * rng.standard_normal((n, N))).astype(np.float32))
        f = orc.vec_corr if name == "vcorr" else orc.pd_vec
        X, _ = f(y, inverse=True)
        fn = lambda: f(X)
        sample = f"oracle {'VecCorrBijector' if name == 'vcorr' else 'PDVecBijector'} per sample (Cholesky + link, 1 thread) on Float32 {K}x{K} x 2^{lb}"
    else:
        lb = 11 if log2_batch is None else log2_batch
        N, K = 1 << lb, 64
        y = np.asfortranarray((0.5 * rng.standard_normal((K * (K - 1) // 2, N))).astype(np.float32))
        fn = lambda: orc.vec_cholesky(y, inverse=True)
        sample = f"oracle _inv_link_chol_lkj per sample (1 thread) on Float32 2016 x 2^{lb}"

    def best_of(f, budget):
        f()  # warm (page faults, libm init)
        best, reps, t_all = float("inf"), 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t_all < budget and reps < 10):
            t0 = time.perf_counter()
            f()
            best = min(best, time.perf_counter() - t0)
            reps += 1
        return best, reps

    best, reps = best_of(fn, budget)
    out = {"value": N / best / 1e6, "unit": unit, "cores": 1, "kind": "port",
           "sample": f"{sample}; best of {reps} runs, {best * 1e3:.1f} ms each; host has {os.cpu_count()} cores"}
    if name == "c1":
        out["us_per_call"] = best * 1e6
        out["M_elements_per_s"] = (1 << 20) / best / 1e6
    if name in ("c2", "c2v") and variants:
        # the best a CPU can do with the same arithmetic: the chain fused into ONE pass, on one
        # core and on all host cores (column blocks on a thread pool; the oracle call releases the GIL)
        from concurrent.futures import ThreadPoolExecutor

        b1, _ = best_of(lambda: orc.chain(ops, x, fused=True), 3.0)
        ncores = os.cpu_count() or 1
        nthr = min(ncores, 64)
        blocks = [np.asfortranarray(x[:, (N * i) // nthr:(N * (i + 1)) // nthr]) for i in range(nthr)]
        with ThreadPoolExecutor(nthr) as pool:
            bN, _ = best_of(lambda: list(pool.map(lambda blk: orc.chain(ops, blk, fused=True), blocks)), 3.0)
        out["variants"] = {"fused_single_pass_1_core": {"value": N / b1 / 1e6, "cores": 1},
                           "fused_single_pass_threads": {"value": N / bN / 1e6, "cores": nthr}}
    return out


def traffic_from_profiles(workload):
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE), stored by the profiling recipe in profiles/; None when not collected."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def traffic_source(workload):
    """`roofline.traffic` is NOT a quantity of this run: it is the PMC measurement stored in profiles/traffic.json by the
    profiling recipe (scripts/collect_profiles.py) under the profile tag named there."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tag = json.load(f).get(workload, {}).get("tag")
        return None if tag is None else f"profiles/traffic.json[{workload}], profile set {tag} (rocprofv3 PMC passes; not measured in this run)"
    except Exception:
        return None


# ---------------------------------------------------------------------------------- one measured workload
class Env:
    pass


def measure_small_calls(env, calls=400, reps=5):
    torch, bj = env.torch, env.bj
    V = bj.vector
    out = []
    for label, t, shape in (("from_linked_vec(positive reals)", V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, float("inf")), (64,)), (64, 256)),
                            ("from_linked_vec(unit interval)", V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, 1.0), (1000,)), (1000, 16))):
        x = torch.randn(shape[1], shape[0], device=env.device).T
        res = {"bijector": label, "rows_x_chains": f"{shape[0]} x {shape[1]}", "dtype": "f32", "log_det": "per chain"}
        for key, on in (("us_per_call", True), ("us_per_call_general_path", False)):
            bj._fast_plans(on)
            try:
                for _ in range(50):
                    bj.with_logabsdet_jacobian(t, x, per_sample=True)
                torch.cuda.synchronize()
                best = float("inf")
                for _ in range(reps):
                    t0 = time.perf_counter()
                    for _ in range(calls):
                        bj.with_logabsdet_jacobian(t, x, per_sample=True)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / calls * 1e6)
                res[key] = round(best, 2)
            finally:
                bj._fast_plans(True)
        out.append(res)
    return out


def measure(env, name, steps, warmup, scaling, log2_batch=None, want_cold=True):
    """Warm up, then time EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks.
    -> dict on every rank (only rank 0 uses it)."""
    torch, bj, dist = env.torch, env.bj, env.dist
    wl = make_workload(name, bj, torch, env.device, env.rank, env.world, log2_batch, scaling)
    ctx = bj.context(env.device)
    lib = bj._lib.load()

    def barrier():
        if getattr(env, "library_collective", False):          # the library's stream first, under its watchdog
            bj._lib.check(ctx.h, lib.bjx_synchronize(ctx.h), "bjx_synchronize")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    barrier()
    t_w = time.perf_counter()
    for _ in range(warmup):
        last = wl["step"]()
    barrier()
    per_warm = (time.perf_counter() - t_w) / max(warmup, 1)

    launches_all = [None]

    def timed_pass(instrument):
        """EXACTLY `steps` steps between barrier + synchronize on both sides, max over ranks.  THREE kinds of pass, never mixed
        (the per-launch hipEvent pairs cost 30-55 % on launch-bound steps and must not sit inside the region
        whose wall clock becomes `value`):
          "wall"   — nothing of ours on the stream: -> wall seconds                     (`value`, `ms_per_step`, `cold`)
          "region" — ONE hipEvent pair around the whole region: -> stream-region ms     (`stream_region_ms_per_step`)
          "kernel" — one hipEvent pair around every dominant-kernel launch: -> Σ ms, n  (`kernel_ms`, `roofline`)
        -> (wall s, stream-region ms, Σ dominant-kernel ms, launches)"""
        nonlocal last
        if instrument == "kernel":
            lib.bjx_kernel_time_begin(ctx.h)      # prof_on = 1 (context stream)
        if instrument == "region":
            lib.bjx_time_begin(ctx.h)
        n_launch0 = lib.bjx_launch_count()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = wl["step"]()
        launches_all[0] = (lib.bjx_launch_count() - n_launch0) / max(steps, 1)      # EVERY kernel launch of the library in the region (helpers included)
        ev_ms = C.c_float(0.0)
        if instrument == "region":
            lib.bjx_time_end(ctx.h, C.byref(ev_ms))
        barrier()
        dt = time.perf_counter() - t0
        k_ms, k_n = C.c_float(0.0), C.c_int(0)
        if instrument == "kernel":
            bj._lib.check(ctx.h, lib.bjx_kernel_time_end(ctx.h, C.byref(k_ms), C.byref(k_n)), "bjx_kernel_time_end")
        if dist is not None:
            tt = torch.tensor([dt, ev_ms.value, k_ms.value], dtype=torch.float64, device=env.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt[0]), float(tt[1]), float(tt[2]), k_n.value
        return dt, ev_ms.value, k_ms.value, k_n.value

    # COLD pass: the W warm-up steps as asked, then K timed steps — what a sampler's first calls or any burst shorter than ~30 ms
    # see (profiles/r03_warmup.md: a fresh process reaches its steady clocks only after 20-30 ms of load; C3 reads 0.54 of the
    # peak like this, 0.61-0.63 after 50-300 steps).  Reported as `cold` (wall clock only, no events on the stream).
    cold = None
    if PREROLL_MS > 0 and want_cold:
        cdt, _, _, _ = timed_pass("wall")
        cold = {"value": wl["total"] / (cdt / steps) / 1e6, "ms_per_step": cdt / steps * 1e3}
    # STEADY passes: MORE untimed steps of the same workload until PREROLL_MS of it have run, then K timed steps three times over
    # (wall, region, kernel).  The count comes from the all-reduced warm-up time, so every rank issues the same number of steps
    # (they contain the collective).
    pre = 0
    if PREROLL_MS > 0:
        # per-step time for the count: K more steps timed AFTER the warm-up (the warm-up's own clock contains one-time costs — code
        # objects, allocator growth — and under-counted the pre-roll by 10x for short steps: C5a's "steady" pass of round 4 started
        # 5 ms into the load, inside the clock transient it was meant to skip, and read 113 µs where 93 µs is the steady state)
        barrier()
        t_p = time.perf_counter()
        for _ in range(steps):
            last = wl["step"]()
        barrier()
        per = (time.perf_counter() - t_p) / steps
        if dist is not None:
            tt = torch.tensor([per], dtype=torch.float64, device=env.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            per = float(tt[0])
        pre = 0 if per <= 0 else max(0, min(4000, int(math.ceil(PREROLL_MS * 1e-3 / per))))
        done = 0
        while done < pre:                        # in bursts of <= 256 steps: the host never runs far ahead of the stream
            nb = min(256, pre - done)
            for _ in range(nb):
                last = wl["step"]()
            done += nb
            torch.cuda.synchronize()
        barrier()
    dt, _, _, _ = timed_pass("wall")
    _, ev, _, _ = timed_pass("region")
    dt_k, _, kern_total, k_launches = timed_pass("kernel")
    ladj_total = float(last[0]) if last is not None else float("nan")
    ms_per_step = dt / steps * 1e3
    # dominant kernel(s) of ONE step: per-launch hipEvent pairs summed, / steps (a step of c3 has two
    # launches, forward + inverse, and `bytes_per_sample` counts both)
    kern_ms = kern_total / steps
    alg_bytes = wl["samples"] * wl["bytes_per_sample"]       # per step, one GPU
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else float("nan")
    res = {
        "workload": name, "label": wl["label"], "value": wl["total"] / (dt / steps) / 1e6, "unit": "M samples/s", "dtype": wl["dtype"],
        "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup, "preroll_steps": pre, "scaling": scaling, "config": wl["cfg"],
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic_from_profiles(name), "traffic_source": traffic_source(name), "kernel": wl["kernel"], "kernel_ms": kern_ms,
                     "kernel_launches_per_step": k_launches / max(steps, 1), "launches_per_step_all": launches_all[0], "stream_region_ms_per_step": ev / steps,
                     "algorithmic_bytes_per_launch": alg_bytes, "frac_of_measured_copy_ceiling_6290": achieved / 6290.0},
        "sum_logabsdetjac": ladj_total, "cold": cold,
        "passes": {"wall_ms_per_step": ms_per_step, "kernel_pass_wall_ms_per_step": dt_k / steps * 1e3,
                   "note": "value / ms_per_step / cold: pass without any event on the stream; kernel_ms: a separate pass with one hipEvent pair per launch"},
    }
    if name == "c1":
        res["us_per_call"] = ms_per_step * 1e3
        res["M_elements_per_s"] = wl["elements"] * env.world / (dt / steps) / 1e6
    del wl, last
    gc.collect()
    torch.cuda.empty_cache()
    return res


def measure_graph(env, name, log2_batch, steps, warmup, scaling):
    """Launch-bound shards: the same step (kernel + finalize + the library's all-reduce) issued call by call and as ONE
    captured hipGraph (bjx_graph_begin/_end/_launch through bj.CapturedStep), `steps` steps each, max over ranks."""
    torch, bj, dist = env.torch, env.bj, env.dist
    wl = make_workload(name, bj, torch, env.device, env.rank, env.world, log2_batch, scaling)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(run):
        barrier()
        t0 = time.perf_counter()
        run()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=env.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        return dt / steps * 1e3

    for _ in range(warmup):
        wl["step"]()
    eager = timed(lambda: [wl["step"]() for _ in range(steps)])
    cs = bj.CapturedStep(wl["step"], env.device)
    cs.replay(warmup)
    graph = timed(lambda: cs.replay(steps))
    s_graph = float(cs.result[0]) if cs.result is not None else float("nan")
    cs.close()
    s_eager = float(wl["step"]()[0])
    res = {"workload": name, "label": wl["label"], "steps": steps, "eager_ms_per_step": eager, "graph_ms_per_step": graph,
           "speedup": eager / graph if graph > 0 else None, "M_samples_per_s_graph": wl["total"] / (graph * 1e-3) / 1e6 if graph > 0 else None,
           "sum_logabsdetjac_eager": s_eager, "sum_logabsdetjac_graph": s_graph}
    del wl, cs
    gc.collect()
    torch.cuda.empty_cache()
    return res


def _sig(v, n=6):
    """n significant digits (a c1 'sample' is one call of 2^20 elements: 4e-5 M samples/s must not round to 0)"""
    return v if not isinstance(v, float) or v == 0 or v != v else float(f"{v:.{n}g}")


def compact_row(r):
    """One BASELINE config as the driver's 8 KB window can carry it: the numbers, no prose (the full record goes to the side file)."""
    if "error" in r:
        return {"workload": r["workload"], "error": r["error"][:120]}
    rf = r["roofline"]
    out = {"workload": r["workload"], "config": {"workload": r["config"]["workload"]}, "dtype": r["dtype"],
           "value": _sig(r["value"]), "ms_per_step": _sig(r["ms_per_step"], 5), "frac": round(rf["frac"], 4),
           "kernel_ms": _sig(rf["kernel_ms"], 5), "stream_ms": _sig(rf["stream_region_ms_per_step"], 5), "kernel": rf["kernel"],
           "launches": rf["kernel_launches_per_step"], "scaling": r["scaling"]}
    if r.get("passes"):        # wall ms/step of the separate pass that carried the per-launch events (NOT what `value` comes from)
        out["ms_with_events"] = _sig(r["passes"]["kernel_pass_wall_ms_per_step"], 5)
    if r.get("cold"):
        out["cold"] = {"value": _sig(r["cold"]["value"]), "ms_per_step": _sig(r["cold"]["ms_per_step"], 5)}
    cb = r.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": None if cb.get("value") is None else _sig(cb["value"], 4), "cores": cb.get("cores", 1)}
    for k in ("us_per_call",):
        if k in r:
            out[k] = round(r[k], 2)
    return out


def build_line(a, world, head, rows, graph_rows, strong, cpu):
    """The ONE JSON line of the contract.  Everything a reader needs to judge the number is in it; explanatory prose and the
    per-row detail (labels, CPU sample descriptions, traffic sources) live in the side file named by `detail`."""
    rf = head["roofline"]
    out = {
        "metric": METRIC, "value": head["value"], "unit": "M samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": head["dtype"], "data": "synthetic (Philox N(0,1), shard-invariant)",
        "config": dict(head["config"], parallelism=f"batch-sharded x{world}, one f64 all-reduce of Σlogabsdetjac ({a.collective})",
                       cache_params=not a.no_cache_params),
        "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms",
                                            "kernel_launches_per_step", "launches_per_step_all", "stream_region_ms_per_step", "algorithmic_bytes_per_launch")},
        "cpu_baseline": cpu,
        "preroll": {"ms": PREROLL_MS, "steps": head.get("preroll_steps", 0)},
        "passes": head.get("passes"),
        "cold": head.get("cold"),
        "sum_logabsdetjac": head["sum_logabsdetjac"],
    }
    for k in ("us_per_call", "M_elements_per_s"):
        if k in head:
            out[k] = head[k]
    if rows:
        out["rows"] = [compact_row(r) for r in rows]
    if graph_rows:
        out["graph_step"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in g.items()
                              if k in ("workload", "log2_batch_per_gpu", "eager_ms_per_step", "graph_ms_per_step", "speedup", "error")} for g in graph_rows]
    if strong:
        out["strong_scaling"] = [compact_row(r) for r in strong]
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher (WORLD_SIZE unset): start the N ranks here, one process per GPU, the way
    `torch.distributed.run` would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relay rank 0's line and fail if any rank fails —
    a `--gpus 8` call must never quietly measure one GPU."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        print(f"bench.py --gpus {a.gpus}: rank exit codes {rcs}", file=sys.stderr)
        sys.exit(1)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a)
    import torch

    env = Env()
    env.torch = torch
    env.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    env.rank = rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        # the line says n_gpus = WORLD_SIZE; a mismatch with --gpus means the launcher and the flag disagree: refuse
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BJX_BENCH_BACKEND=gloo BJX_BENCH_ONE_DEVICE=1: N ranks on ONE GPU over gloo — exercises the multi-rank code path
        # (sharding, rows, strong-scaling sub-lines, max-over-ranks timing) on the single-GPU boxes this build has; the
        # driver's multi-GPU runs use the defaults (one GPU per rank, "nccl" = RCCL over xGMI)
        backend = os.environ.get("BJX_BENCH_BACKEND", "nccl")
        if os.environ.get("BJX_BENCH_ONE_DEVICE"):
            local = 0
        elif torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} needs cuda:{local}, the node has {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        dist = None
    env.dist = dist
    env.device = torch.device("cuda", torch.cuda.current_device())

    import bijectors_amd as bj

    env.bj = bj
    # The workloads call their bijectors with FROZEN parameters, step after step: the case `bj.cache_params` (Julia: `param_epoch!`)
    # exists for — tables derived from parameter arrays are kept while torch's version counters stand still.  The library's default is
    # OFF (a write the host cannot see must never meet a kept table), so the line says `cache_params: true`, and the row
    # `c3_uncached` is the spline with the default setting (its table rebuilt by a helper launch on every call).
    env.cache_params = not a.no_cache_params
    if env.cache_params:
        bj.cache_params(True)
    if a.collective == "bjx" and world > 1:
        bj.shard.init_comm(env.device, timeout_ms=60000)   # ncclUniqueId broadcast through torch.distributed, then RCCL inside the library;
        bj.shard.use_library_collective(True)               # watchdog: a stuck collective is an error after 60 s, not a hang
        env.library_collective = True

    head = measure(env, a.workload, a.steps, a.warmup, a.scaling, a.log2_batch)
    rows, strong = [], []
    want_rows = (not a.no_rows) and a.workload == "c2" and a.log2_batch is None
    if want_rows:
        rsteps, rwarm = max(3, min(a.steps, 20)), max(1, min(a.warmup, 5))      # the sub-lines run as warm as the headline
        for r in [r for r in a.rows.split(",") if r]:
            try:
                if r.endswith("_uncached"):        # the same workload with the library's default: no reuse of parameter tables
                    with bj.cache_params(False):
                        row = measure(env, r[:-len("_uncached")], rsteps, rwarm, a.scaling)
                    row["workload"] = r
                    row["config"] = dict(row["config"], workload=row["config"]["workload"] + " — cache_params OFF (library default)")
                    rows.append(row)
                else:
                    rows.append(measure(env, r, rsteps, rwarm, a.scaling))
            except Exception as e:                 # a sub-line must never cost the headline
                rows.append({"workload": r, "error": repr(e)})
        if world > 1 and a.scaling == "weak":
            for r in ("c2", "c4"):
                try:
                    strong.append(measure(env, r, rsteps, rwarm, "strong", want_cold=False))
                except Exception as e:
                    strong.append({"workload": r, "error": repr(e)})

    small_rows = []
    if want_rows and world == 1:
        # f-2 shapes (src/vector/product/fill.jl:146-165, 192-213): what a sampler calls on every log-density evaluation — the linked vector of a
        # product distribution, one column per chain.  Wall time per call of a burst of 400 calls (host issue + stream, drained at the end), through
        # the cached launch plan (include/bjx.h "plans") and through the general path of rounds 1-5.
        try:
            small_rows = measure_small_calls(env)
        except Exception as e:
            small_rows = [{"error": repr(e)}]

    graph_rows = []
    if want_rows and (world == 1 or a.collective == "bjx"):
        # shards of 2^20 columns and below are launch-bound: the step as a captured hipGraph next to call-by-call issue.
        # (multi-rank only with --collective bjx: the all-reduce must be recorded on the library's stream)
        for wl_, lb in (("c1", None), ("c2", 20), ("c2", 16)):      # c1 = BASELINE configs[0]: one Float64 vector of 2^20, a launch-latency case
            try:
                graph_rows.append(dict(measure_graph(env, wl_, lb, 50, 5, a.scaling), log2_batch_per_gpu=lb))
            except Exception as e:
                graph_rows.append({"workload": wl_, "log2_batch_per_gpu": lb, "error": repr(e)})

    if rank == 0:
        cpu_ok = not a.no_cpu_baseline and world == 1
        cpu = None
        if cpu_ok:
            try:
                cpu = cpu_baseline(a.workload, a.cpu_log2_batch)
            except Exception as e:  # the baseline is informational; never lose the GPU number
                cpu = {"value": None, "unit": "M samples/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
            for r in rows:
                if "error" not in r and not r["workload"].endswith("_uncached"):       # (the same CPU leg as the cached row)
                    try:
                        r["cpu_baseline"] = cpu_baseline(r["workload"], None, budget=2.5, variants=False)
                    except Exception as e:
                        r["cpu_baseline"] = {"value": None, "unit": "M samples/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        out = build_line(a, world, head, rows if want_rows else [], graph_rows, strong, cpu)
        if small_rows:
            out["small_calls"] = small_rows
        # the full record (labels, CPU sample descriptions, traffic sources, graph-step sums): a side file, best effort
        detail = {"line": out, "head": head, "rows": rows, "graph_step": graph_rows, "strong_scaling": strong,
                  "notes": {"preroll": "untimed steps of the same workload after the W warm-up steps, until the GPU has been under load for `ms` (steady clocks); the timed region is exactly K steps",
                            "cold": "K timed steps right after the W warm-up steps, no pre-roll: what a burst shorter than ~30 ms sees",
                            "traffic": "roofline.traffic is the rocprofv3 PMC measurement stored in profiles/traffic.json (not measured in this run)"}}
        for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
            try:
                os.makedirs(d, exist_ok=True)
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(detail, f, indent=1)
                out["detail"] = os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
                break
            except Exception:
                continue
        print(json.dumps(out, separators=(",", ":")))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
