import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch, bijectors_amd as bj
from _timing import kernel_and_region_ms
for dt in (torch.float32, torch.float64):
    es = 4 if dt == torch.float32 else 8
    for dim in (4, 8, 16, 32, 128, 200, 512, 1500):
        N = (1 << 29) // (dim * (es // 4)); nl = 8
        w = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5; u = torch.randn(dim, nl, device="cuda", dtype=dt) / dim ** 0.5
        b = torch.randn(nl, device="cuda", dtype=dt)
        layer = bj.PlanarLayer(w, u, b)
        x = torch.randn(N, dim, device="cuda", dtype=dt).T; g = torch.randn(N, dim, device="cuda", dtype=dt).T; lb = torch.randn(N, device="cuda", dtype=dt)
        y = bj.transform(layer, x)
        _, ms = kernel_and_region_ms(bj, lambda: bj.vjp(bj.inverse(layer), y, g, lb), steps=5, warm=2)
        print(f"{str(dt)[6:]} {dim}: vjp(inverse) {ms:.3f} ms  {3*dim*es*N/ms/1e6/80:.1f} %", flush=True)
