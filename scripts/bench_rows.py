#!/usr/bin/env python3
"""One measured line per SURVEY.md §8(a) row (forward and inverse where the row has both): dominant
kernel time from per-launch hipEvent pairs (bjx_kernel_time_begin/_end), algorithmic bytes
(read input once + write output once + 4 B/sample when a per-sample log-det is written) and the
fraction of the 8 TB/s HBM peak.  Float32, inputs resident in HBM.  Output: a markdown table.

    python scripts/bench_rows.py [--log2-batch 22] [--steps 10] [--only name,...]
"""
import argparse
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402
from _timing import kernel_and_region_ms  # noqa: E402

PEAK = 8000.0


def cm(rows, batch, dev, dt=torch.float32):
    return torch.empty((batch, rows), dtype=dt, device=dev).T


def randn(rows, batch, dev, seed, std=1.0, mean=0.0):
    t = cm(rows, batch, dev)
    L, ctx = bj._lib, bj.context(dev)
    L.check(ctx.h, L.load().bjx_fill_normal(ctx.h, L.BJX_F32, t.data_ptr(), rows, batch, 0, seed, mean, std), "fill")
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-batch", type=int, default=22)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N = 1 << a.log2_batch
    d = 64
    f32 = torch.float32
    x = randn(d, N, dev, 0)
    xpos = torch.exp(0.5 * x.T).T
    xunit = torch.sigmoid(x.T).T
    rows = []  # (name, reference, callable, bytes_per_sample, samples)

    def add(name, ref, b, inp, out_rows=None, per_sample=True, samples=None):
        n = inp.shape[-1] if samples is None else samples
        in_rows = inp.numel() // n
        orows = in_rows if out_rows is None else out_rows
        y = cm(orows, n, dev) if inp.dim() == 2 and orows * n < (1 << 34) else None
        bps = 4 * (in_rows + orows) + (4 if per_sample else 0)

        def step():
            return bj.shard.with_logabsdet_jacobian_sharded(b, inp, out=y, per_sample=per_sample)

        rows.append((name, ref, step, bps, n))

    e = bj.elementwise
    add("exp∘Shift∘Scale (sum)", "a1,a3,a4,a5", e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), x, per_sample=False)
    add("exp∘Shift∘Scale (per-sample ladj)", "a1,a3,a4,a5", e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), x)
    add("elementwise(log)", "a2", e(bj.log), xpos, per_sample=False)
    add("Logit(0,1)", "a6", bj.Logit(0.0, 1.0), xunit, per_sample=False)
    add("inverse(Logit(0,1))", "a6", bj.inverse(bj.Logit(0.0, 1.0)), x, per_sample=False)
    add("LeakyReLU(0.1)", "a7", bj.LeakyReLU(0.1), x, per_sample=False)
    add("TruncatedBijector(0,1)", "a8", bj.TruncatedBijector(0.0, 1.0), xunit, per_sample=False)
    add("inverse(TruncatedBijector(0,1))", "a8", bj.inverse(bj.TruncatedBijector(0.0, 1.0)), x, per_sample=False)
    add("OrderedBijector", "a9", bj.OrderedBijector(), x)
    xo = bj.transform(bj.OrderedBijector(), x)
    add("inverse(OrderedBijector)", "a9", bj.inverse(bj.OrderedBijector()), xo)
    Ns = 1 << min(a.log2_batch, 22)
    xs = torch.softmax(randn(d, Ns, dev, 1).T, dim=1).T
    add("SimplexBijector K=64", "a10,a12", bj.SimplexBijector(), xs, out_rows=d - 1)
    ys = randn(d - 1, Ns, dev, 2)
    add("inverse(SimplexBijector) K=64", "a11", bj.inverse(bj.SimplexBijector()), ys, out_rows=d)
    K = 64
    Nc = 1 << min(a.log2_batch, 16)
    n = K * (K - 1) // 2
    yv = randn(n, Nc, dev, 3, std=0.5)
    icb = bj.inverse(bj.VecCholeskyBijector("U"))
    rows.append(("inverse(VecCholesky) K=64", "a14", lambda: bj.shard.with_logabsdet_jacobian_sharded(icb, yv), 4 * (n + K * K) + 4, Nc))
    Wd, _, _ = bj.shard.with_logabsdet_jacobian_sharded(icb, yv)
    cb = bj.VecCholeskyBijector("U")
    rows.append(("VecCholesky K=64 (W→y)", "a13", lambda: bj.shard.with_logabsdet_jacobian_sharded(cb, Wd), 4 * (n + K * K) + 4, Nc))
    dp, nl = 128, 8
    Np = 1 << min(a.log2_batch, 22)
    z = randn(dp, Np, dev, 4)
    w = randn(dp, nl, dev, 5, std=1 / math.sqrt(dp))
    u = randn(dp, nl, dev, 6, std=1 / math.sqrt(dp))
    bb = randn(nl, 1, dev, 7).reshape(-1).contiguous()
    flow = bj.PlanarLayer(w, u, bb)
    add("8×PlanarLayer d=128", "a15", flow, z)
    zf = bj.transform(flow, z)
    add("inverse(8×PlanarLayer) d=128", "a16", bj.inverse(flow), zf)
    one = bj.PlanarLayer(w[:, 0].contiguous(), u[:, 0].contiguous(), bb[:1].contiguous())
    add("1×PlanarLayer d=128", "a15", one, z)
    rad = bj.RadialLayer(torch.tensor([0.3], device=dev), torch.tensor([0.5], device=dev), randn(dp, 1, dev, 8).reshape(-1).contiguous())
    add("RadialLayer d=128", "a17", rad, z)
    add("inverse(RadialLayer) d=128", "a17", bj.inverse(rad), z)
    bn = bj.InvertibleBatchNorm(torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.ones(d, device=dev))
    add("InvertibleBatchNorm (eval) d=64", "a18", bn, x)
    bnt = bj.InvertibleBatchNorm(torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.ones(d, device=dev))

    def bn_train_step():
        with bj.training():
            return bj.shard.with_logabsdet_jacobian_sharded(bnt, x, out=y_bn)

    y_bn = cm(d, N, dev)
    rows.append(("InvertibleBatchNorm (training mode: statistics pass + apply pass) d=64", "a18", bn_train_step, 3 * 4 * d + 4, N))   # x is read twice
    dr, Kb = 32, 16
    raw = [randn(dr, k, dev, 100 + i) for i, k in enumerate((Kb, Kb, Kb - 1))]
    rqs = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
    xr = randn(dr, N, dev, 9)
    add("RQS K=16 d=32 forward", "a19", rqs, xr)
    yr = bj.transform(rqs, xr)
    add("RQS K=16 d=32 inverse", "a19", bj.inverse(rqs), yr)
    gr = randn(dr, N, dev, 31)
    lbr = randn(N, 1, dev, 32).reshape(-1).contiguous()
    rows.append(("vjp(RQS K=16 d=32)", "f-1", lambda: bj.vjp(rqs, xr, gr, lbr), 4 * 3 * dr + 4, N))
    rows.append(("vjp(inverse(RQS K=16 d=32))", "f-1", lambda: bj.vjp(bj.inverse(rqs), yr, gr, lbr), 4 * 3 * dr + 4, N))
    rows.append(("vjp_params(RQS K=16 d=32): input pullback + knot and raw-parameter cotangents (one pass over x, ȳ, ℓ̄)", "f-1",
                 lambda: bj.vjp_params(rqs, xr, gr, lbr), 4 * 3 * dr + 4, N))
    perm = bj.Permute(list(torch.randperm(d, generator=torch.Generator().manual_seed(0)).add(1).tolist()))
    add("Permute d=64", "a21", perm, x, per_sample=False)
    mask = bj.PartitionMask(d, list(range(1, d // 2 + 1)), list(range(d // 2 + 1, d + 1)))
    sc = torch.full((d // 2,), 1.5, device=dev)
    cpl = bj.Coupling(lambda x2: bj.Shift(0.25) @ bj.Scale(sc), mask)
    add("Coupling(Shift∘Scale) d=64, θ = [n1,batch] scale + shift arrays", "a20", cpl, x)
    rows[-1] = rows[-1][:3] + (rows[-1][3] + 2 * (d // 2) * 4,) + rows[-1][4:]     # the per-sample θ arrays are read too

    stk = bj.Stacked([e(bj.exp), bj.Logit(0.0, 1.0), bj.identity, e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)], [(1, 16), (17, 32), (33, 48), (49, 64)])
    xst = x.clone()
    xst[16:32] = xunit[16:32]
    add("Stacked(exp|Logit|identity|exp∘Shift∘Scale) d=64", "f-4", stk, xst)

    gb = randn(d, N, dev, 11)
    lbar = randn(N, 1, dev, 12).reshape(-1).contiguous()
    ob = bj.OrderedBijector()
    # pullbacks read the primal input + the output cotangent (+ 4 B/sample of log-det cotangent) and write the input cotangent
    rows.append(("vjp(OrderedBijector) d=64", "f-1", lambda: bj.vjp(ob, x, gb, lbar), 3 * d * 4 + 4, N))
    rows.append(("vjp(inverse(OrderedBijector)) d=64", "f-1", lambda: bj.vjp(bj.inverse(ob), xo, gb, lbar), 3 * d * 4 + 4, N))

    Wb = randn(K * K, Nc, dev, 13).T.reshape(Nc, K, K).permute(2, 1, 0)      # column-major (K, K, batch): strides (1, K, K²) — a plain reshape gives (K, 1, K²) and the call a layout copy (0.8 ms)
    lb2 = randn(Nc, 1, dev, 14).reshape(-1).contiguous()
    rows.append(("vjp(inverse(VecCholesky)) K=64", "f-1", lambda: bj.vjp(icb, yv, Wb, lb2), 4 * (2 * n + K * K) + 4, Nc))

    gyv = randn(n, Nc, dev, 23, std=1.0)
    rows.append(("vjp(VecCholesky forward link) K=64", "f-1", lambda: bj.vjp(cb, Wd, gyv), 4 * (n + 2 * K * K), Nc))

    gxs = randn(d, Ns, dev, 15)
    gys = randn(d - 1, Ns, dev, 16)
    lbs = randn(Ns, 1, dev, 17).reshape(-1).contiguous()
    sb_ = bj.SimplexBijector()
    rows.append(("vjp(SimplexBijector) K=64", "f-1", lambda: bj.vjp(sb_, xs, gys, lbs), 4 * (d + d - 1 + d) + 4, Ns))
    rows.append(("vjp(inverse(SimplexBijector)) K=64", "f-1", lambda: bj.vjp(bj.inverse(sb_), ys, gxs, lbs), 4 * (d - 1 + d + d - 1) + 4, Ns))
    rows.append(("vjp(Stacked(exp|Logit|identity|exp∘Shift∘Scale)) d=64", "f-1", lambda: bj.vjp(stk, xst, gb, lbar), 3 * d * 4 + 4, N))

    gz = randn(dp, Np, dev, 21)
    lbz = randn(Np, 1, dev, 22).reshape(-1).contiguous()
    rows.append(("vjp(8×PlanarLayer) d=128", "f-1", lambda: bj.vjp(flow, z, gz, lbz), 4 * 3 * dp + 4, Np))

    rows.append(("vjp(inverse(8×PlanarLayer)) d=128", "f-1", lambda: bj.vjp(bj.inverse(flow), zf, gz, lbz), 4 * 3 * dp + 4, Np))

    rows.append(("vjp(RadialLayer) d=128", "f-1", lambda: bj.vjp(rad, z, gz, lbz), 4 * 3 * dp + 4, Np))
    rows.append(("vjp_params(RadialLayer) d=128 (input pullback with the row sums for z̄₀ in the same pass)", "f-1", lambda: bj.vjp_params(rad, z, gz, lbz), 4 * 3 * dp + 4 + 8 + 64, Np))
    rows.append(("vjp_params(8×PlanarLayer) d=128 (input + w̄, ū, b̄; two passes)", "f-1", lambda: bj.vjp_params(flow, z, gz, lbz), 4 * 5 * dp + 4 + 4 * 4 * nl, Np))

    rows.append(("vjp_params(inverse(8×PlanarLayer)) d=128 (inverse transform + inverse input pullback + forward parameter pullback at the pre-image)", "f-1",
                 lambda: bj.vjp_params(bj.inverse(flow), zf, gz, lbz), 4 * 2 * dp + 4 * 3 * dp + 4 + 4 * 5 * dp + 4 + 4 * 4 * nl, Np))

    # §8(f) f-3: logpdf(td, Y) fused into the inverting kernel — Y is read once, x is never stored
    td_pl = bj.transformed(bj.MvNormal(dp), flow)
    rows.append(("logpdf(transformed(MvNormal(128), 8×PlanarLayer)) d=128", "f-3", lambda: bj.logpdf(td_pl, zf), 4 * dp + 4, Np))
    td_ch = bj.transformed(bj.MvNormal(torch.zeros(d, device=dev), torch.ones(d, device=dev)), e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5))
    rows.append(("logpdf(transformed(MvNormal(μ,σ), exp∘Shift∘Scale)) d=64", "f-3", lambda: bj.logpdf(td_ch, xpos), 4 * d + 4, N))

    rows.append(("rand(transformed(MvNormal(μ,σ), exp∘Shift∘Scale), 2^22) d=64 (samples drawn in the kernel)", "f-3", lambda: bj.rand(td_ch, N, seed=1), 4 * d, N))
    mf = e(bj.exp) @ bj.Shift(torch.zeros(d, device=dev)) @ bj.Scale(torch.ones(d, device=dev))
    rows.append(("vjp_params(exp∘Shift(μ)∘Scale(σ)) d=64 (input pullback with the row moments in the same pass)", "f-1", lambda: bj.vjp_params(mf, x, gb, lbar), 4 * 3 * d + 4 + 16, N))
    c2b = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    rows.append(("logabsdetjac(exp∘Shift∘Scale) alone (values not stored)", "a1,a5", lambda: bj.logabsdetjac(c2b, x), 4 * d, N))

    # §8(f) f-4: matrix-variate constraint bijectors (per-sample Cholesky), Scale with a matrix, Stacked with structured blocks
    # K = 4 and 8: the sizes LKJ / Wishart blocks have in models (one lane per sample); K = 32: lanes along the rows
    for Km, lbm in ((4, 22), (8, 20), (32, 18)):
        Nm = 1 << min(a.log2_batch, lbm)
        for nm, cls in (("VecCorrBijector", bj.VecCorrBijector), ("PDVecBijector", bj.PDVecBijector)):
            bm = cls()
            nvm = bm._n(Km)
            ym = randn(nvm, Nm, dev, 40, std=0.3)
            Xm = bj.transform(bj.inverse(bm), ym)
            bpsm = 4 * (Km * Km + nvm) + 4
            rows.append((f"{nm} K={Km} (X → Cholesky → y)", "f-4", (lambda b_=bm, X_=Xm: bj.shard.with_logabsdet_jacobian_sharded(b_, X_)), bpsm, Nm))
            rows.append((f"inverse({nm}) K={Km} (y → X = U'U)", "f-4", (lambda b_=bm, y_=ym: bj.shard.with_logabsdet_jacobian_sharded(bj.inverse(b_), y_)), bpsm, Nm))
    # §8(f) f-1 x f-4: pullbacks of the matrix bijectors (bjx_*_vjp).  Algorithmic bytes: primal input + output cotangent read, input
    # cotangent written, + the log-det cotangent
    for Km, lbm in ((3, 22), (4, 22), (8, 20), (12, 19), (16, 18), (24, 16), (32, 16), (48, 14), (64, 14)):
        Nm = 1 << min(a.log2_batch, lbm)
        for nm, cls in (("VecCorrBijector", bj.VecCorrBijector), ("PDVecBijector", bj.PDVecBijector)):
            bm = cls()
            nvm = bm._n(Km)
            ym = randn(nvm, Nm, dev, 40, std=0.3)
            Xm = bj.transform(bj.inverse(bm), ym)
            Xbar = randn(Km * Km, Nm, dev, 42).T.reshape(Nm, Km, Km).permute(2, 1, 0)      # column-major (K, K, batch): strides (1, K, K²)
            ybar = randn(nvm, Nm, dev, 43)
            lbm_ = randn(Nm, 1, dev, 44).reshape(-1).contiguous()
            rows.append((f"vjp(inverse({nm})) K={Km} (ȳ from X̄, ℓ̄)", "f-1", (lambda b_=bm, y_=ym, g_=Xbar, l_=lbm_: bj.vjp(bj.inverse(b_), y_, g_, l_)), 4 * (2 * nvm + Km * Km) + 4, Nm))
            rows.append((f"vjp({nm}) K={Km} (X̄ from ȳ, ℓ̄)", "f-1", (lambda b_=bm, X_=Xm, g_=ybar, l_=lbm_: bj.vjp(b_, X_, g_, l_)), 4 * (2 * Km * Km + nvm) + 4, Nm))
    Am = (randn(d, d, dev, 41, std=1 / math.sqrt(d)) + 1.5 * torch.eye(d, device=dev).T).T.contiguous().T
    add("Scale(64×64 matrix): a * x + logabsdet(a)", "f-4", bj.Scale(Am), x)
    add("inverse(Scale(64×64 matrix)): a \\ y", "f-4", bj.inverse(bj.Scale(Am)), x)
    # the parameter pullback of the matrix Scale (round 6): input pullback (x̄ = aᵀȳ: x, ȳ read, x̄ written) + ā = ȳxᵀ + Σℓ̄·a⁻ᵀ (x, ȳ read again by the
    # matrix-core outer-product kernel): 5 array passes + ℓ̄
    gsm = randn(d, N, dev, 45)
    lsm = randn(N, 1, dev, 46).reshape(-1).contiguous()
    rows.append(("vjp_params(Scale(64×64 matrix)): x̄ = aᵀȳ and ā = ȳxᵀ + Σℓ̄·a⁻ᵀ (outer-product sum on the matrix cores)", "f-1", lambda: bj.vjp_params(bj.Scale(Am), x, gsm, lsm), 4 * 5 * d + 4, N))
    # the same two calls with the reuse of parameter tables opted into (bj.cache_params: the factorisation of an unchanged matrix
    # is kept — no prep kernel in the steady state), and logpdf / rand with a FULL covariance (whitening = the matrix Scale)
    cached_rows = set()

    def cached(fn):          # measured inside ONE `with bj.cache_params():` region (entering and leaving per call would drop the tables each time)
        cached_rows.add(fn)
        return fn

    bsm, ibsm = bj.Scale(Am), bj.inverse(bj.Scale(Am))
    ysm = cm(d, N, dev)
    rows.append(("Scale(64×64 matrix) under cache_params (factorisation kept)", "f-4", cached(lambda: bj.shard.with_logabsdet_jacobian_sharded(bsm, x, out=ysm)), 4 * 2 * d + 4, N))
    rows.append(("inverse(Scale(64×64 matrix)) under cache_params", "f-4", cached(lambda: bj.shard.with_logabsdet_jacobian_sharded(ibsm, x, out=ysm)), 4 * 2 * d + 4, N))
    cov = (Am @ Am.T).T.contiguous().T
    td_cov = bj.transformed(bj.MvNormal(torch.zeros(d, device=dev), cov=cov), e(bj.exp))
    rows.append(("logpdf(transformed(MvNormal(μ, Σ full 64×64), exp)) d=64", "f-3", lambda: bj.logpdf(td_cov, xpos), 4 * d + 4, N))
    rows.append(("logpdf(… MvNormal(μ, Σ full)) under cache_params", "f-3", cached(lambda: bj.logpdf(td_cov, xpos)), 4 * d + 4, N))
    mix = bj.Stacked([e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), bj.SimplexBijector(), bj.Logit(0.0, 1.0), bj.OrderedBijector()], [(1, 16), (17, 32), (33, 48), (49, 64)])
    xmix = x.clone()
    xmix[16:32] = torch.softmax(x[16:32].T, dim=1).T
    xmix[32:48] = xunit[32:48]
    add("Stacked(exp∘Shift∘Scale | Simplex | Logit | Ordered) d=64 → 63: structured blocks in the same launch", "f-4", mix, xmix, out_rows=d - 1)

    only = [s for s in a.only.split(",") if s]
    L, ctx = bj._lib, bj.context(dev)
    lib = L.load()
    # "kernel": the launch(es) inside the library's profiling scope; "stream": one event pair around all the calls of a separate
    # pass — helper launches (parameter preparation, finalize) and gaps included
    print("| row | §8(a) | kernel ms | samples | alg. B/sample | GB/s | % of 8 TB/s | stream ms | % of 8 TB/s by stream region |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, ref, step, bps, n in rows:
        if only and not any(o in name for o in only):
            continue
        try:
            if step in cached_rows:
                with bj.cache_params():
                    k, reg = kernel_and_region_ms(bj, step, steps=a.steps, device=dev)
            else:
                k, reg = kernel_and_region_ms(bj, step, steps=a.steps, device=dev)
            gbs = bps * n / (k * 1e-3) / 1e9
            gbr = bps * n / (reg * 1e-3) / 1e9
            print(f"| {name} | {ref} | {k:.4f} | 2^{int(math.log2(n))} | {bps} | {gbs:.0f} | {100 * gbs / PEAK:.1f} | {reg:.4f} | {100 * gbr / PEAK:.1f} |", flush=True)
        except Exception as ex:  # keep the table going
            print(f"| {name} | {ref} | failed: {ex!r} | | | | | | |", flush=True)


if __name__ == "__main__":
    main()
