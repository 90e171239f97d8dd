"""Small correlation / covariance blocks (K = 2 ... 8, the sizes LKJ / Wishart priors have in practice): one lane per sample
(matrix_lane_kernel) against lanes along the rows (BJX_MATRIX_LANE_MAX=0 in a second process).  Prints a markdown table."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
import ctypes as C

dev = torch.device("cuda", 0)
DT = torch.float64 if os.environ.get("BJX_BENCH_F64") else torch.float32
EB = 8 if DT == torch.float64 else 4
N = 1 << int(os.environ.get("BJX_BENCH_LOG2N", "20"))
lib = bj._lib.load()
ctx = bj.context(dev)
print("| bijector | K | kernel ms | alg. B/sample | GB/s | % of 8 TB/s |")
print("|---|---|---|---|---|---|")
for K in tuple(int(k) for k in os.environ.get("BJX_BENCH_KS", "2,3,4,5,8,9,12").split(",")):
    nv = K * (K - 1) // 2
    y = (0.4 * torch.randn(N, nv, device=dev, dtype=DT)).T if nv else torch.zeros(0, N, device=dev, dtype=DT)
    for name, b, nu in (("VecCorrBijector", bj.VecCorrBijector(), nv), ("PDVecBijector", bj.PDVecBijector(), K * (K + 1) // 2)):
        yy = (0.4 * torch.randn(N, nu, device=dev, dtype=DT)).T
        X = bj.transform(bj.inverse(b), yy)
        for label, bb, xin in ((name, b, X), (f"inverse({name})", bj.inverse(b), yy)):
            from _timing import kernel_ms
            kms = kernel_ms(bj, lambda: bj.with_logabsdet_jacobian(bb, xin, per_sample=True), steps=10, device=dev)
            bytes_ps = (K * K + nu) * EB + EB
            gbs = bytes_ps * N / (kms * 1e-3) / 1e9
            print(f"| {label} | {K} | {kms:.4f} | {bytes_ps} | {gbs:.0f} | {gbs / 80:.1f} |")
