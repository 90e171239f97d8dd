"""Shared timing for the row benchmarks: dominant-kernel time from the library's per-launch hipEvent pairs, AFTER a
clock-settling pre-roll.  Round 3 (profiles/r03_warmup.md): a kernel measured after 2-5 warm-up launches runs 8-15 % below
its steady state when those launches last less than ~20-30 ms (C3 0.54 -> 0.61-0.63 of the HBM peak on the same box with 50+
warm-up steps), so every row first runs untimed until PREROLL_MS of the same work have gone through the GPU."""
import ctypes as C
import os
import time

import torch

PREROLL_MS = float(os.environ.get("BJX_BENCH_PREROLL_MS", "60"))


def kernel_ms(bj, fn, steps=10, warm=3, device=None):
    """-> average summed dominant-kernel milliseconds of one fn() call."""
    lib = bj._lib.load()
    ctx = bj.context(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / max(warm, 1)
    if PREROLL_MS > 0 and per > 0:
        for _ in range(max(0, min(4000, int(PREROLL_MS * 1e-3 / per) + 1 - warm))):
            fn()
        torch.cuda.synchronize()
    lib.bjx_kernel_time_begin(ctx.h)
    for _ in range(steps):
        fn()
    ms, n = C.c_float(0), C.c_int(0)
    bj._lib.check(ctx.h, lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(n)), "bjx_kernel_time_end")
    return ms.value / steps


def kernel_and_region_ms(bj, fn, steps=10, warm=3, device=None):
    """-> (average summed dominant-kernel ms, average STREAM-REGION ms) of one fn() call, from two separate passes after ONE
    pre-roll: the region pass has a single hipEvent pair around all `steps` calls, so it contains every helper launch (parameter
    preparation, finalize) and every gap — the cost a caller sees on the stream (a row whose helper launch
    costs as much as its hot kernel must say so)."""
    k = kernel_ms(bj, fn, steps=steps, warm=warm, device=device)
    lib = bj._lib.load()
    ctx = bj.context(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()
    lib.bjx_time_begin(ctx.h)
    for _ in range(steps):
        fn()
    ms = C.c_float(0)
    lib.bjx_time_end(ctx.h, C.byref(ms))
    torch.cuda.synchronize()
    return k, ms.value / steps
