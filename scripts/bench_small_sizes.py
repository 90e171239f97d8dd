"""Small per-sample sizes (what constraint bijectors see in models: K = 3 ... 8): VecCholesky, Simplex, Ordered.  Markdown table."""
import os, sys
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


print("| bijector | K | samples | kernel ms | alg. B/sample | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
N = 1 << 22
for K in (3, 4, 8, 16):
    nv = K * (K - 1) // 2
    y = (0.4 * torch.randn(N, nv, device=dev)).T
    b = bj.VecCholeskyBijector("U")
    W = bj.transform(bj.inverse(b), y)
    for label, bb, xin in ((f"inverse(VecCholesky) (y → W)", bj.inverse(b), y), ("VecCholesky (W → y)", b, W)):
        ms = timed(lambda: bj.with_logabsdet_jacobian(bb, xin, per_sample=True))
        bps = (K * K + nv) * 4 + 4
        g = bps * N / (ms * 1e-3) / 1e9
        print(f"| {label} | {K} | 2^22 | {ms:.4f} | {bps} | {g:.0f} | {g / 80:.1f} |")
for K in (3, 4, 8):
    nv = K * (K - 1) // 2
    y = (0.4 * torch.randn(N, nv, device=dev)).T
    Wb = torch.randn(N, K * K, device=dev).T.reshape(K, K, N)
    lb = torch.randn(N, device=dev)
    b = bj.VecCholeskyBijector("U")
    ms = timed(lambda: bj.vjp(bj.inverse(b), y, Wb, lb))
    bps = (2 * nv + K * K) * 4 + 4
    g = bps * N / (ms * 1e-3) / 1e9
    print(f"| vjp(inverse(VecCholesky)) | {K} | 2^22 | {ms:.4f} | {bps} | {g:.0f} | {g / 80:.1f} |")
for K in (3, 4, 8, 16):
    x = torch.softmax(torch.randn(N, K, device=dev), dim=1).T
    for label, bb, xin in (("SimplexBijector", bj.SimplexBijector(), x), ("OrderedBijector", bj.OrderedBijector(), (torch.randn(N, K, device=dev)).T)):
        ms = timed(lambda: bj.with_logabsdet_jacobian(bb, xin, per_sample=True))
        out_rows = K - 1 if "Simplex" in label else K
        bps = (K + out_rows) * 4 + 4
        g = bps * N / (ms * 1e-3) / 1e9
        print(f"| {label} | {K} | 2^22 | {ms:.4f} | {bps} | {g:.0f} | {g / 80:.1f} |")
