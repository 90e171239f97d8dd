"""Probe: Simplex / Ordered (forward, inverse, pullbacks) at tall columns (Dirichlet dimensions of topic models)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
dev = torch.device("cuda", 0)
lib = bj._lib.load(); ctx = bj.context(dev)
def timed(fn, reps=5):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _timing import kernel_ms
    return kernel_ms(bj, fn, steps=reps, device=dev)
N = 1 << int(os.environ.get("BJX_BENCH_LOG2N", "20"))
TD = torch.float64 if os.environ.get("BJX_PROBE_DTYPE") == "f64" else torch.float32
ES = 8 if TD == torch.float64 else 4
print("| bijector | K | kernel ms (2^%d columns) | alg. B/sample | GB/s | %% of 8 TB/s |" % (N.bit_length() - 1))
print("|---|---|---|---|---|---|")
for K in tuple(int(v) for v in os.environ.get("BJX_BENCH_KS", "64,100,128,200,256,500").split(",")):
    x = torch.softmax(torch.randn(N, K, device=dev, dtype=TD), dim=1).T
    sb = bj.SimplexBijector()
    y = bj.transform(sb, x)
    xo = torch.randn(N, K, device=dev, dtype=TD).T
    gy, gx, lb = torch.randn(N, K - 1, device=dev, dtype=TD).T, torch.randn(N, K, device=dev, dtype=TD).T, torch.randn(N, device=dev, dtype=TD)
    rows = [("SimplexBijector", lambda: bj.with_logabsdet_jacobian(sb, x, per_sample=True), (2 * K - 1) * ES + ES),
            ("inverse(SimplexBijector)", lambda: bj.with_logabsdet_jacobian(bj.inverse(sb), y, per_sample=True), (2 * K - 1) * ES + ES),
            ("OrderedBijector", lambda: bj.with_logabsdet_jacobian(bj.OrderedBijector(), xo, per_sample=True), 2 * K * ES + ES),
            ("vjp(SimplexBijector)", lambda: bj.vjp(sb, x, gy, lb), (3 * K - 1) * ES + ES),
            ("vjp(inverse(SimplexBijector))", lambda: bj.vjp(bj.inverse(sb), y, gx, lb), (3 * K - 2) * ES + ES)]
    if os.environ.get("BJX_PROBE_ROWS") == "fwd": rows = rows[:3]
    for label, fn, bps in rows:
        ms = timed(fn)
        g = bps * N / (ms * 1e-3) / 1e9
        print(f"| {label} | {K} | {ms:.4f} | {bps} | {g:.0f} | {g / 80:.1f} |", flush=True)
