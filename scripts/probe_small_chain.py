"""Probe: exp∘Shift∘Scale with a per-sample log-det on short columns (the mixed walker), for counter runs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
dev = torch.device("cuda", 0)
N = 1 << 22
e = bj.elementwise
for d in [int(v) for v in os.environ.get("BJX_BENCH_DIMS", "2,3,5").split(",")]:
    x = torch.randn(N, d, device=dev).T
    ch = e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
    for _ in range(int(os.environ.get("BJX_PROBE_REPS", "20"))):
        bj.with_logabsdet_jacobian(ch, x, per_sample=True)
    torch.cuda.synchronize()
