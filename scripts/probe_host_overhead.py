#!/usr/bin/env python3
"""Where the HOST time of a small call goes (32-45 us per call for kernels of 5-15 us): cProfile over N calls of
a few representative bijectors at 16 columns, top functions by internal time, and the wall time per call with the stream drained only
at the end.   python scripts/probe_host_overhead.py [--calls 3000]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bijectors_amd as bj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=3000)
    ap.add_argument("--top", type=int, default=18)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-plans", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    d, n = 1001, 16
    x = torch.randn(n, d, device=dev).T
    g = torch.randn(n, d, device=dev).T
    lb = torch.randn(n, device=dev)
    chain = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    mu, sig = torch.zeros(d, device=dev), torch.ones(d, device=dev)
    mf = bj.elementwise(bj.exp) @ bj.Shift(mu) @ bj.Scale(sig)
    x128 = torch.randn(n, 128, device=dev).T
    layers = [bj.PlanarLayer(torch.randn(128, device=dev) / 11, torch.randn(128, device=dev) / 11, torch.randn(1, device=dev)) for _ in range(8)]
    flow = layers[0]
    for l in layers[1:]:
        flow = l @ flow
    # f-2 shapes (src/vector/product/fill.jl:146-165, 192-213): what a sampler calls per log-density evaluation — the linked vector of a
    # product distribution, one column per chain
    V = bj.vector
    x64 = torch.randn(256, 64, device=dev).T                                   # 64 parameters x 256 chains
    link64 = V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, float("inf")), (64,))
    link1000 = V.from_linked_vec(V.scalar_to_scalar_bijector(0.0, 1.0), (1000,))
    x1000 = torch.randn(n, 1000, device=dev).T                                 # 1 000 parameters x 16 chains
    cases = {
        "chain fwd (scalar params) 1001 x 16": lambda: bj.with_logabsdet_jacobian(chain, x),
        "chain fwd per_sample (vector params)": lambda: bj.with_logabsdet_jacobian(mf, x, per_sample=True),
        "f-2 from_linked_vec(positive) 64 x 256 chains": lambda: bj.with_logabsdet_jacobian(link64, x64, per_sample=True),
        "f-2 from_linked_vec(unit interval) 1000 x 16 chains": lambda: bj.with_logabsdet_jacobian(link1000, x1000, per_sample=True),
        "chain vjp": lambda: bj.vjp(chain, x, g, lb),
        "vjp_params(mean-field chain)": lambda: bj.vjp_params(mf, x, g, lb),
        "8 x PlanarLayer composition 128 x 16": lambda: bj.with_logabsdet_jacobian(flow, x128),
    }
    if a.no_plans:
        bj._fast_plans(False)               # the general path of rounds 1-5 (op list walked, marshalled and hashed per call)
        print("(launch plans OFF: general path)")
    for name, fn in cases.items():
        if a.only and a.only not in name:
            continue
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.calls):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"\n== {name}: host issue {t_issue / a.calls * 1e6:.1f} us/call, with the stream drained {t_all / a.calls * 1e6:.1f} us/call")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.calls):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats("tottime")
        rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[: a.top]
        for (fname, line, func), (cc, nc, tt, ct, _) in rows:
            print(f"   {tt / a.calls * 1e6:7.2f} us self  {ct / a.calls * 1e6:7.2f} us cum  {nc / a.calls:5.1f} calls  {os.path.basename(fname)}:{line} {func}")


if __name__ == "__main__":
    main()
