#!/usr/bin/env python3
"""How a bench workload's step time moves with the time the GPU has been under load (cold vs steady).

    python scripts/probe_steady.py c5a [--chunks 40] [--steps 20] [--out]

Runs `chunks` back-to-back chunks of `steps` steps of the bench.py workload; every chunk is timed by the wall clock between two
synchronisations (no event on the stream) — and every 4th chunk a second time with the per-launch hipEvent pairs — and printed
with the milliseconds of load that preceded it.  `--out` passes a preallocated output (no allocator traffic in the step).
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import bijectors_amd as bj  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("workload")
    p.add_argument("--chunks", type=int, default=40)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--log2-batch", type=int, default=None)
    p.add_argument("--fin", default=None, help="comma list of BJX_OPT_INKERNEL_FINALIZE values cycled chunk by chunk (an A/B inside one process), e.g. 0,2")
    a = p.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.make_workload(a.workload, bj, torch, dev, 0, 1, a.log2_batch, "weak")
    ctx = bj.context(dev)
    lib = bj._lib.load()
    modes = [int(v) for v in a.fin.split(",")] if a.fin else None
    for _ in range(3):
        wl["step"]()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    print(f"| chunk | load before (ms) | wall µs/step | kernel µs/step | launches/step | finalize mode |")
    print("|---|---|---|---|---|---|")
    per_mode = {}
    for c in range(a.chunks):
        with_ev = (c % 4 == 3) and not modes
        mode = ""
        if modes:
            mode = modes[c % len(modes)]
            torch.cuda.synchronize()
            bj._lib.check(ctx.h, lib.bjx_set_option(ctx.h, bj._lib.BJX_OPT_INKERNEL_FINALIZE, mode), "bjx_set_option")
        if with_ev:
            lib.bjx_kernel_time_begin(ctx.h)
        torch.cuda.synchronize()
        before = (time.perf_counter() - t_start) * 1e3
        t0 = time.perf_counter()
        for _ in range(a.steps):
            wl["step"]()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k = ""
        n = ""
        if with_ev:
            ms, cnt = C.c_float(0), C.c_int(0)
            lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(cnt))
            k = f"{ms.value / a.steps * 1e3:.1f}"
            n = f"{cnt.value / a.steps:.1f}"
        print(f"| {c} | {before:.1f} | {dt / a.steps * 1e6:.1f}{' (events on)' if with_ev else ''} | {k} | {n} | {mode} |")
        if modes and c >= len(modes) * 2:
            per_mode.setdefault(mode, []).append(dt / a.steps * 1e6)
    for m, v in per_mode.items():
        v.sort()
        print(f"finalize mode {m}: median {v[len(v) // 2]:.2f} µs/step, min {v[0]:.2f} (n={len(v)})")


if __name__ == "__main__":
    main()
