"""C3 diagnostic (round 3): is the forward launch slower than the inverse because of the kernel or because of the buffers?
Times each direction alone (kernel events), on each buffer pairing, and with the output buffer displaced by odd offsets."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bijectors_amd as bj
from bench import colmajor_empty, fill_normal

dev = torch.device("cuda:0"); f32 = torch.float32
dim, K, N = 32, 16, 1 << 22
ctx = bj.context(dev); lib = bj._lib.load()
big = torch.empty(4 * dim * N + (64 << 20) // 4, dtype=f32, device=dev)      # x | y | xb | slack, carved by hand
def view(off_elems): return big[off_elems:off_elems + dim * N].view(N, dim).T
raw = [colmajor_empty(torch, dim, k, f32, dev) for k in (K, K, K - 1)]
for i, r in enumerate(raw): fill_normal(bj, torch, r, 0, seed=100 + i)
b = bj.RationalQuadraticSpline(raw[0], raw[1], raw[2], 3.0); ib = bj.inverse(b)
sharded = bj.shard.with_logabsdet_jacobian_sharded

def timeit(fn, steps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    lib.bjx_kernel_time_begin(ctx.h)
    for _ in range(steps): fn()
    ms, n = C.c_float(0), C.c_int(0)
    lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(n))
    return ms.value / n.value * 1e3

E = dim * N
for pad in (0, 0, 1 << 10, 3 << 10, (1 << 20) + (3 << 10), (5 << 20) + (7 << 10)):     # displacement of y / xb in ELEMENTS
    x, y, xb = view(0), view(E + pad), view(2 * E + 2 * pad)
    fill_normal(bj, torch, x, 0, seed=0)
    sharded(b, x, out=y)
    t_f = timeit(lambda: sharded(b, x, out=y))
    t_i = timeit(lambda: sharded(ib, y, out=xb))
    t_f2 = timeit(lambda: sharded(b, xb, out=y))            # forward reading the buffer the inverse wrote
    t_i2 = timeit(lambda: sharded(ib, x, out=xb))           # inverse reading the N(0,1) input
    def both():
        sharded(b, x, out=y); sharded(ib, y, out=xb)
    t_b = timeit(both)
    print(f"pad {pad*4:>9d} B: fwd x->y {t_f:6.1f} us | inv y->xb {t_i:6.1f} | fwd xb->y {t_f2:6.1f} | inv x->xb {t_i2:6.1f} | alternating avg {t_b:6.1f}")
