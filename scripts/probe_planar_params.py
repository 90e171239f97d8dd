"""Probe: vjp_params(8×PlanarLayer) on short columns, for rocprofv3 --stats."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj
dev = torch.device("cuda", 0)
N = 1 << 22
d = int(os.environ.get("BJX_BENCH_DIMS", "3"))
x = torch.randn(N, d, device=dev).T
g = torch.randn(N, d, device=dev).T
lb = torch.randn(N, device=dev)
W8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
U8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
fl = bj.PlanarLayer(W8, U8, torch.randn(8, device=dev))
for _ in range(10):
    bj.vjp_params(fl, x, g, lb)
torch.cuda.synchronize()
