"""8 x PlanarLayer forward at the heights given on the command line (2^21 columns, a few launches each): the workload of a counter
pass (rocprofv3 --pmc ... -- python scripts/probe_planar_odd.py 200 201)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj

dev = torch.device("cuda", 0)
N = 1 << 21
for d in [int(v) for v in sys.argv[1:]]:
    x = torch.randn(N, d, device=dev).T
    W8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    U8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    fl = bj.PlanarLayer(W8, U8, torch.randn(8, device=dev))
    for _ in range(4):
        bj.with_logabsdet_jacobian(fl, x)
    torch.cuda.synchronize()
