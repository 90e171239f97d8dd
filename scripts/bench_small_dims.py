"""Small row counts (dim = 2 ... 10: toy flows, low-dimensional posteriors): one line per bijector.  Markdown table."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
import bijectors_amd as bj

dev = torch.device("cuda", 0)
lib = bj._lib.load()
ctx = bj.context(dev)


def timed(fn, reps=10):
    from _timing import kernel_ms          # clock-settling pre-roll, then the per-launch event pairs
    return kernel_ms(bj, fn, steps=reps, device=dev)


print("| bijector | dim | kernel ms (2^22 columns) | alg. B/sample | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
N = 1 << 22
e = bj.elementwise
for d in [int(v) for v in os.environ.get("BJX_BENCH_DIMS", "2,3,8,10,24,100,200").split(",")]:
    x = torch.randn(N, d, device=dev).T
    cases = []
    w = torch.randn(d, device=dev) / math.sqrt(d)
    u = torch.randn(d, device=dev) / math.sqrt(d)
    cases.append(("PlanarLayer", bj.PlanarLayer(w, u, torch.randn(1, device=dev)), x))
    W8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    U8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    cases.append(("8×PlanarLayer", bj.PlanarLayer(W8, U8, torch.randn(8, device=dev)), x))
    cases.append(("RadialLayer", bj.RadialLayer(torch.tensor([0.5], device=dev), torch.tensor([0.3], device=dev), torch.randn(d, device=dev)), x))
    cases.append(("InvertibleBatchNorm (eval)", bj.InvertibleBatchNorm(torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev), torch.randn(d, device=dev),
                                                                       torch.rand(d, device=dev) + 0.5), x))
    cases.append(("Permute", bj.Permute(list(torch.randperm(d).add(1).tolist())), x))
    raw = [torch.randn(d, 8, device=dev), torch.randn(d, 8, device=dev), torch.randn(d, 7, device=dev)]
    cases.append(("RQS K=8", bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0), x))
    cases.append(("exp∘Shift∘Scale (per-sample ladj)", e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5), x))
    if d >= 2:
        m = bj.PartitionMask(d, list(range(1, d // 2 + 1)), list(range(d // 2 + 1, d + 1)))
        sc = torch.full((d // 2,), 1.5, device=dev)
        cases.append(("Coupling(Shift∘Scale)", bj.Coupling(lambda x2: bj.Shift(0.25) @ bj.Scale(sc), m), x))
    if d >= 3:
        a_, b_ = d // 3, 2 * (d // 3)
        xs = x.clone()
        xs[a_:b_] = torch.rand(N, b_ - a_, device=dev).T * 0.9 + 0.05
        cases.append(("Stacked(exp | Logit | identity)", bj.Stacked([e(bj.exp), bj.Logit(0.0, 1.0), bj.identity], [(1, a_), (a_ + 1, b_), (b_ + 1, d)]), xs))
    for name, b, xin in cases:
        if b is None:
            continue
        try:
            ms = timed(lambda: bj.with_logabsdet_jacobian(b, xin, per_sample=True) if not isinstance(b, bj.PlanarLayer) else bj.with_logabsdet_jacobian(b, xin))
        except Exception as ex:
            print(f"| {name} | {d} | error {ex!r} | | | |")
            continue
        bps = 2 * d * 4 + 4
        g = bps * N / (ms * 1e-3) / 1e9
        print(f"| {name} | {d} | {ms:.4f} | {bps} | {g:.0f} | {g / 80:.1f} |")
