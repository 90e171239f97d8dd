"""Worker of tests/test_gpu_env_switches.py: runs a fixed battery of calls through the library under whatever BJX_* tuning switches the
environment sets (they are read once per process) and saves every result to the .npz given as argv[1]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bijectors_amd as bj  # noqa: E402


def dev(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if a.ndim == 2:
        return torch.from_numpy(np.ascontiguousarray(a.T)).cuda().T
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 1, 0))).cuda().permute(2, 1, 0)


def main(path):
    out = {}
    r = np.random.default_rng(2024)

    def put(tag, *ts):
        for i, t in enumerate(ts):
            if hasattr(t, "result"):
                put(tag + f".{i}", t.result, t.logabsdetjac)
            elif isinstance(t, (tuple, list)):
                put(tag + f".{i}", *t)
            elif isinstance(t, dict):
                put(tag + f".{i}", *[t[k] for k in sorted(t) if isinstance(t[k], torch.Tensor)])
            elif isinstance(t, torch.Tensor):
                out[f"{tag}.{i}"] = t.detach().double().cpu().numpy()
            elif t is not None:
                out[f"{tag}.{i}"] = np.asarray(float(t))

    f32, f64 = np.float32, np.float64
    e = bj.elementwise
    for dt, tg in ((f32, "f32"), (f64, "f64")):
        tdt = torch.float32 if dt == f32 else torch.float64
        for dim, N in ((64, 1500), (10, 777), (24, 300), (200, 130)):
            x = np.asfortranarray(r.normal(size=(dim, N)).astype(dt))
            av = torch.linspace(0.5, 1.5, dim, dtype=tdt).cuda()
            ch = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(av)
            put(f"chain.{tg}.{dim}", bj.with_logabsdet_jacobian(ch, dev(x), per_sample=True), bj.with_logabsdet_jacobian(ch, dev(x)))
            put(f"logit.{tg}.{dim}", bj.with_logabsdet_jacobian(bj.inverse(bj.Logit(-1.0, 2.0)), dev(x), per_sample=True))
            nl = 3
            w, u, b = r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=nl)
            fl = bj.PlanarLayer(torch.tensor(w, dtype=tdt), torch.tensor(u, dtype=tdt), torch.tensor(b, dtype=tdt))
            yf = bj.with_logabsdet_jacobian(fl, dev(x))
            put(f"planar.{tg}.{dim}", yf, bj.with_logabsdet_jacobian(bj.inverse(fl), yf.result))
            g, lb = np.asfortranarray(r.normal(size=(dim, N)).astype(dt)), r.normal(size=N).astype(dt)
            put(f"planar_vjp.{tg}.{dim}", bj.vjp(fl, dev(x), dev(g), dev(lb)), bj.vjp_params(fl, dev(x), dev(g), dev(lb)))
            rd = bj.RadialLayer(torch.tensor([0.3], dtype=tdt), torch.tensor([0.2], dtype=tdt), torch.tensor(r.normal(size=dim), dtype=tdt))
            put(f"radial.{tg}.{dim}", bj.with_logabsdet_jacobian(rd, dev(x)), bj.vjp(rd, dev(x), dev(g), dev(lb)))
            put(f"ordered.{tg}.{dim}", bj.with_logabsdet_jacobian(bj.OrderedBijector(), dev(x), per_sample=True), bj.vjp(bj.OrderedBijector(), dev(x), dev(g), dev(lb)))
            xs = np.asfortranarray(r.dirichlet(np.ones(dim), size=N).T.astype(dt))
            ys = bj.with_logabsdet_jacobian(bj.SimplexBijector(), dev(xs), per_sample=True)
            put(f"simplex.{tg}.{dim}", ys, bj.with_logabsdet_jacobian(bj.inverse(bj.SimplexBijector()), ys[0], per_sample=True))
            put(f"simplex_vjp.{tg}.{dim}", bj.vjp(bj.SimplexBijector(), dev(xs), dev(g[:dim - 1]), dev(lb)), bj.vjp(bj.inverse(bj.SimplexBijector()), ys[0], dev(g), dev(lb)))
            if dim in (24, 200):
                K = 8
                raw = [dev(r.normal(size=(dim, k)).astype(dt)) for k in (K, K, K - 1)]
                sp = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0)
                put(f"rqs.{tg}.{dim}", bj.with_logabsdet_jacobian(sp, dev(x), per_sample=True))
                if dim == 24:        # knot / raw-parameter cotangents
                    xbk, grk = bj.vjp_params(sp, dev(x), dev(g), dev(lb))
                    put(f"rqs_vjp_knots.{tg}.{dim}", xbk, grk["widths"], grk["heights"], grk["derivatives"])
            if dim == 64:
                st = bj.Stacked([ch_seg(bj, av), bj.SimplexBijector(), bj.Logit(0.0, 1.0), bj.OrderedBijector()], [(1, 16), (17, 32), (33, 48), (49, 64)])
                xm = x.copy()
                xm[16:32] = r.dirichlet(np.ones(16), size=N).T
                xm[32:48] = r.uniform(0.05, 0.95, size=(16, N))
                put(f"stacked.{tg}", bj.with_logabsdet_jacobian(st, dev(np.asfortranarray(xm.astype(dt))), per_sample=True))
                A = (r.normal(size=(dim, dim)) / np.sqrt(dim) + 1.5 * np.eye(dim)).astype(dt)
                put(f"scale_matrix.{tg}", bj.with_logabsdet_jacobian(bj.Scale(dev(A)), dev(x), per_sample=True), bj.with_logabsdet_jacobian(bj.inverse(bj.Scale(dev(A))), dev(x), per_sample=True))
        # heights that are not whole 16-byte packs, past the tile walkers (element-aligned packs, row slabs, the 8 / 16-wave Planar tile)
        for dim, N in ((101, 90), (333, 50), (601, 40)):
            x = np.asfortranarray(r.normal(size=(dim, N)).astype(dt))
            nl = 3
            w, u, b = r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=(dim, nl)) / np.sqrt(dim), r.normal(size=nl)
            fl = bj.PlanarLayer(torch.tensor(w, dtype=tdt), torch.tensor(u, dtype=tdt), torch.tensor(b, dtype=tdt))
            put(f"planar_odd.{tg}.{dim}", bj.with_logabsdet_jacobian(fl, dev(x)))
            g, lb = np.asfortranarray(r.normal(size=(dim, N)).astype(dt)), r.normal(size=N).astype(dt)
            put(f"planar_odd_vjp.{tg}.{dim}", bj.vjp(fl, dev(x), dev(g), dev(lb)), bj.vjp_params(fl, dev(x), dev(g), dev(lb)))
            rd = bj.RadialLayer(torch.tensor([0.3], dtype=tdt), torch.tensor([0.2], dtype=tdt), torch.tensor(r.normal(size=dim), dtype=tdt))
            put(f"radial_odd.{tg}.{dim}", bj.with_logabsdet_jacobian(rd, dev(x)))
            bn = bj.InvertibleBatchNorm(torch.tensor(r.normal(size=dim), dtype=tdt), torch.tensor(0.3 * r.normal(size=dim), dtype=tdt),
                                        torch.tensor(r.normal(size=dim), dtype=tdt), torch.tensor(r.uniform(0.5, 2, size=dim), dtype=tdt), eps=1e-5)
            put(f"bn_odd.{tg}.{dim}", bj.with_logabsdet_jacobian(bn, dev(x)))
            a_, b_ = dim // 3, 2 * (dim // 3)
            xs = x.copy()
            xs[a_:b_] = r.uniform(0.05, 0.95, size=(b_ - a_, N))
            av = torch.linspace(0.5, 1.5, a_, dtype=tdt).cuda()
            st = bj.Stacked([e(bj.exp) @ bj.Scale(av), bj.Logit(0.0, 1.0), bj.identity], [(1, a_), (a_ + 1, b_), (b_ + 1, dim)])
            put(f"stacked_odd.{tg}.{dim}", bj.with_logabsdet_jacobian(st, dev(np.asfortranarray(xs.astype(dt))), per_sample=True))
            g_, lb_ = np.asfortranarray(r.normal(size=(dim, N)).astype(dt)), r.normal(size=N).astype(dt)
            put(f"stacked_odd_vjp.{tg}.{dim}", bj.vjp(st, dev(np.asfortranarray(xs.astype(dt))), dev(g_), dev(lb_)))
        for K, N in ((4, 300), (9, 130), (16, 70), (64, 9)):
            nv = K * (K - 1) // 2
            y = np.asfortranarray((r.normal(size=(nv, N)) * min(0.6, 1.6 / np.sqrt(K))).astype(dt))
            ib = bj.inverse(bj.VecCholeskyBijector("U"))
            W = bj.with_logabsdet_jacobian(ib, dev(y), per_sample=True)
            put(f"chol.{tg}.{K}", W, bj.with_logabsdet_jacobian(bj.VecCholeskyBijector("U"), W[0], per_sample=True))
            gw, lb = np.asfortranarray(r.normal(size=(K, K, N)).astype(dt)), r.normal(size=N).astype(dt)
            put(f"chol_vjp.{tg}.{K}", bj.vjp(ib, dev(y), dev(gw), dev(lb)), bj.vjp(bj.VecCholeskyBijector("U"), W[0], dev(y)))
            vc = bj.VecCorrBijector()
            X = bj.with_logabsdet_jacobian(bj.inverse(vc), dev(y), per_sample=True)
            put(f"vcorr.{tg}.{K}", X, bj.with_logabsdet_jacobian(vc, X[0], per_sample=True))
            if K <= 16 or K == 64:          # pullbacks of the matrix bijectors (one lane / one group of lanes per sample / MFMA blocks)
                yb = np.asfortranarray(r.normal(size=(nv, N)).astype(dt))
                put(f"vcorr_vjp.{tg}.{K}", bj.vjp(bj.inverse(vc), dev(y), dev(gw), dev(lb)), bj.vjp(vc, X[0], dev(yb), dev(lb)))
                pv = bj.PDVecBijector()
                # (a random triangular factor is exponentially ill-conditioned in K: 0.4 N(0,1) below the diagonal gives cotangents of 2e6 at K = 64, where
                # every kernel — and every switch — is one rounding pattern of a problem conditioned 1e5: scripts/probe_matrix_vjp_cond.py)
                ypd = np.asfortranarray(((0.4 if K <= 16 else 1.6 / K) * r.normal(size=(K * (K + 1) // 2, N))).astype(dt))
                Xpd = bj.transform(bj.inverse(pv), dev(ypd))
                put(f"pdvec_vjp.{tg}.{K}", bj.vjp(bj.inverse(pv), dev(ypd), dev(gw), dev(lb)), bj.vjp(pv, Xpd, dev(ypd), dev(lb)))
    np.savez(path, **out)


def ch_seg(bj, av):
    return bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(av[:16])


if __name__ == "__main__":
    main(sys.argv[1])
