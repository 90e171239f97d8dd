"""Probe: where the time of the fused log-density chain goes (same-call variants)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bijectors_amd as bj
dev = torch.device("cuda:0")
d, N = 64, 1 << 22
g = torch.Generator(device=dev); g.manual_seed(1)
xpos = (torch.rand((N, d), device=dev, generator=g) + 0.6).T
e = bj.elementwise
L, ctx = bj._lib, bj.context(dev); lib = L.load()
def t(name, step, steps=10):
    for _ in range(3): step()
    torch.cuda.synchronize()
    lib.bjx_kernel_time_begin(ctx.h)
    for _ in range(steps): step()
    ms, cnt = C.c_float(0), C.c_int(0)
    L.check(ctx.h, lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(cnt)), "time_end")
    print(f"{name:70s} {ms.value/steps:.4f} ms  launches/step {cnt.value/steps:.1f}", flush=True)
c2 = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
mu, sg = torch.zeros(d, device=dev), torch.ones(d, device=dev)
t("logabsdetjac sum (no store)", lambda: bj.logabsdetjac(c2, xpos))
t("logabsdetjac per-sample (no store)", lambda: bj.interface._run_chain(bj.interface._fused_ops(c2), xpos, True, True, store=False))
t("inverse chain per-sample (no store)", lambda: bj.interface._run_chain(bj.interface._fused_ops(bj.inverse(c2)), xpos, True, True, store=False))
t("logpdf MvNormal(d) (no params)", lambda: bj.logpdf(bj.transformed(bj.MvNormal(d), c2), xpos))
t("logpdf MvNormal(mu,sigma)", lambda: bj.logpdf(bj.transformed(bj.MvNormal(mu, sg), c2), xpos))
t("logpdf MvNormal(mu,sigma) identity transform", lambda: bj.logpdf(bj.transformed(bj.MvNormal(mu, sg)), xpos))
