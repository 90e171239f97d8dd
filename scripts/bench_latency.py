#!/usr/bin/env python3
"""Launch-bound regime (the reference's everyday shapes: one vector, a handful of chains): time per
`with_logabsdet_jacobian` call issued eagerly through the host mirror vs replayed from a hipGraph that captured
the same calls (C1 is one 2²⁰-element vector; configs[0]).  Prints a markdown table.

usage: python scripts/bench_latency.py [--batch 8] [--reps 300]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bijectors_amd as bj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    r = np.random.default_rng(0)
    d, N = 64, a.batch
    dev = lambda x: torch.from_numpy(np.asfortranarray(x.astype(np.float32)).T.copy()).cuda().T
    x = dev(r.normal(size=(d, N)))
    p = dev(r.dirichlet(np.ones(d), size=N).T)
    av = torch.linspace(0.5, 1.5, d).cuda()
    chain = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(av)
    w = torch.from_numpy((r.normal(size=(d, 8)) / 8).astype(np.float32)).cuda()
    u = torch.from_numpy((r.normal(size=(d, 8)) / 8).astype(np.float32)).cuda()
    b = torch.from_numpy(r.normal(size=8).astype(np.float32)).cuda()
    flow = bj.PlanarLayer(w, u, b)
    stk = bj.Stacked([bj.elementwise(bj.exp), bj.Logit(-5.0, 5.0), bj.identity, chain_seg(av)], [(1, 16), (17, 32), (33, 48), (49, 64)])
    td = bj.transformed(bj.MvNormal(d), flow)
    cases = [
        ("exp∘Shift∘Scale", lambda: bj.with_logabsdet_jacobian(chain, x)),
        ("SimplexBijector K=64", lambda: bj.with_logabsdet_jacobian(bj.SimplexBijector(), p, per_sample=True)),
        ("8×PlanarLayer d=64", lambda: bj.with_logabsdet_jacobian(flow, x)),
        ("Stacked (4 segments)", lambda: bj.with_logabsdet_jacobian(stk, x, per_sample=True)),
        ("logpdf(transformed(MvNormal, 8×Planar))", lambda: bj.logpdf(td, x)),
        ("all five in sequence", None),
    ]
    fns = [c[1] for c in cases[:-1]]
    cases[-1] = (cases[-1][0], lambda: [f() for f in fns])
    s = torch.cuda.Stream()
    print(f"| call (dim 64 × batch {N}, Float32) | eager µs/call | hipGraph replay µs/call | ratio |")
    print("|---|---|---|---|")
    for name, fn in cases:
        with torch.cuda.stream(s):
            for _ in range(5):
                fn()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                fn()
            s.synchronize()
            eager = (time.perf_counter() - t0) / a.reps * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / a.reps * 1e6
        print(f"| {name} | {eager:.1f} | {rep:.1f} | {eager / rep:.1f}× |")


def chain_seg(av):
    return bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(av[:16])


if __name__ == "__main__":
    main()
