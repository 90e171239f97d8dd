import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bijectors_amd as bj
dev = torch.device("cuda", 0)
K, Nc = 64, 1 << 16
n = K * (K - 1) // 2
L, ctx = bj._lib, bj.context(dev)
lib = L.load()
yv = torch.empty((Nc, n), device=dev).T
L.check(ctx.h, lib.bjx_fill_normal(ctx.h, L.BJX_F32, yv.data_ptr(), n, Nc, 0, 3, 0.0, 0.5), "fill")
icb = bj.inverse(bj.VecCholeskyBijector("U")); cb = bj.VecCholeskyBijector("U")
Wd = bj.transform(icb, yv)
def timeit(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.bjx_kernel_time_begin(ctx.h)
    for _ in range(10): fn()
    ms, cnt = C.c_float(0), C.c_int(0)
    lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(cnt))
    print(f"{name}: {ms.value/10:.4f} ms ({cnt.value} launches)")
timeit(lambda: bj.transform(cb, Wd), "fwd transform only")
timeit(lambda: bj.with_logabsdet_jacobian(cb, Wd), "fwd + ladj")
timeit(lambda: bj.transform(icb, yv), "inv transform only")
timeit(lambda: bj.logabsdetjac(icb, yv), "inv ladj only")
timeit(lambda: bj.with_logabsdet_jacobian(icb, yv), "inv + ladj")
