"""A few calls of the group kernels at a whole-pack height and the odd height next to it (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
does reading 16-byte packs on element-aligned addresses cost HBM traffic, or only pipeline time?  2^20 columns."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bijectors_amd as bj

dev = torch.device("cuda", 0)
N = 1 << 20
e = bj.elementwise
for d in [int(v) for v in os.environ.get("BJX_BENCH_DIMS", "100,101,200,201").split(",")]:
    x = torch.randn(N, d, device=dev).T
    W8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    U8 = torch.randn(d, 8, device=dev) / math.sqrt(d)
    cases = [
        bj.PlanarLayer(torch.randn(d, device=dev) / math.sqrt(d), torch.randn(d, device=dev) / math.sqrt(d), torch.randn(1, device=dev)),
        bj.PlanarLayer(W8, U8, torch.randn(8, device=dev)),
        bj.RadialLayer(torch.tensor([0.5], device=dev), torch.tensor([0.3], device=dev), torch.randn(d, device=dev)),
        bj.InvertibleBatchNorm(torch.randn(d, device=dev), 0.1 * torch.randn(d, device=dev), torch.randn(d, device=dev), torch.rand(d, device=dev) + 0.5),
        e(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5),
    ]
    a_, b_ = d // 3, 2 * (d // 3)
    xs = x.clone()
    xs[a_:b_] = torch.rand(N, b_ - a_, device=dev).T * 0.9 + 0.05
    st = bj.Stacked([e(bj.exp), bj.Logit(0.0, 1.0), bj.identity], [(1, a_), (a_ + 1, b_), (b_ + 1, d)])
    for b in cases:
        for _ in range(2):
            if isinstance(b, bj.PlanarLayer):
                bj.with_logabsdet_jacobian(b, x)
            else:
                bj.with_logabsdet_jacobian(b, x, per_sample=True)
    for _ in range(2):
        bj.with_logabsdet_jacobian(st, xs, per_sample=True)
    torch.cuda.synchronize()
    print("done", d, flush=True)
