import sys, math, numpy as np, torch, ctypes as C
sys.path.insert(0, '/root/repo')
import bijectors_amd as bj
Lm = bj._lib; lib = Lm.load()
dim, N = 64, 1 << 22
g = torch.Generator().manual_seed(1)
y = torch.exp(0.5 * torch.randn(N, dim, generator=g)).cuda().T
A = torch.randn(dim, dim, generator=g) / 8
Lc = torch.linalg.cholesky(A @ A.T + 0.3 * torch.eye(dim)).contiguous().T.contiguous().T   # any layout: just timing
Ld = Lc.T.contiguous().T.cuda()
mu = torch.randn(dim, generator=g).cuda()
ctx = bj.context()
lp = torch.empty(N, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())
ops1 = (Lm.BjxOp * 2)(Lm.BjxOp(Lm.OP_LOG, 0, 0.0, 0.0, None, None), Lm.BjxOp(Lm.OP_SHIFT, dim, 0.0, 0.0, mu.data_ptr(), None))
ops0 = (Lm.BjxOp * 1)()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
lib.bjx_set_option(ctx.h, Lm.BJX_OPT_PARAM_EPOCH, 1) if hasattr(Lm, 'BJX_OPT_PARAM_EPOCH') else None
print('plain density only      ', t(lambda: lib.bjx_scale_matrix(ctx.h, Lm.BJX_F32, 1, p(Ld), p(y), None, p(lp), None, dim, N, Lm.BJX_BASE_STDNORMAL)))
print('chain n_ops=0 density   ', t(lambda: lib.bjx_scale_matrix_chain(ctx.h, Lm.BJX_F32, 1, p(Ld), ops0, 0, p(y), None, p(lp), dim, N, Lm.BJX_BASE_STDNORMAL)))
print('chain shift only        ', t(lambda: lib.bjx_scale_matrix_chain(ctx.h, Lm.BJX_F32, 1, p(Ld), (Lm.BjxOp * 1)(Lm.BjxOp(Lm.OP_SHIFT, dim, 0.0, 0.0, mu.data_ptr(), None)), 1, p(y), None, p(lp), dim, N, Lm.BJX_BASE_STDNORMAL)))
print('chain log+shift density ', t(lambda: lib.bjx_scale_matrix_chain(ctx.h, Lm.BJX_F32, 1, p(Ld), ops1, 2, p(y), None, p(lp), dim, N, Lm.BJX_BASE_STDNORMAL)))
out = torch.empty_like(y)
print('plain with store        ', t(lambda: lib.bjx_scale_matrix(ctx.h, Lm.BJX_F32, 1, p(Ld), p(y), p(out), None, None, dim, N, 0)))
