#!/usr/bin/env python3
"""dlopen, first-call (context + code-object load) and second-call latency of the library named by BJX_LIB_PATH (default: the one in
the tree) in a fresh process; `scripts/gpu_run.sh profile` keeps the output as <TAG>_first_call.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

t0 = time.perf_counter()
import bijectors_amd as bj  # noqa: E402

bj._lib.load()
t1 = time.perf_counter()
x = torch.randn(64, 1024, device="cuda").T.contiguous().T
torch.cuda.synchronize()
t2 = time.perf_counter()
b = bj.elementwise(bj.exp) @ bj.Shift(0.1) @ bj.Scale(0.5)
y, l = bj.with_logabsdet_jacobian(b, x)
torch.cuda.synchronize()
t3 = time.perf_counter()
bj.with_logabsdet_jacobian(b, x)
torch.cuda.synchronize()
t4 = time.perf_counter()
ok = bool(torch.allclose(y, torch.exp(0.5 * x + 0.1), rtol=1e-5))
lib = os.environ.get("BJX_LIB_PATH") or os.path.join(os.path.dirname(bj.__file__) if hasattr(bj, "__file__") else ".", "libbjx_hip.so")
size = os.path.getsize(lib) if os.path.exists(lib) else -1
print(f"library {size / 1e6:.1f} MB: dlopen {1e3 * (t1 - t0):.1f} ms; first call (context + code-object load of the kernel) {1e3 * (t3 - t2):.1f} ms; "
      f"second call {1e3 * (t4 - t3):.3f} ms; result ok = {ok}")
