#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2c; O=gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matrix_bij or batchnorm_training or scale_with or named_stacked or coupling" -p no:cacheprovider > $O/tests.txt 2>&1; tail -4 $O/tests.txt
grep -E "^FAILED|Mismatched|Max abs|Max rel" $O/tests.txt | head -30
b() { python bench.py --no-cpu-baseline --no-rows --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for wl in vcorr pdvec; do echo -n "$wl : "; b --workload $wl; done
python - <<'PY'
import torch, time, math, sys
sys.path.insert(0, '.')
import bijectors_amd as bj
import ctypes as C
dev = torch.device('cuda', 0)
L, ctx = bj._lib, bj.context(dev)
lib = L.load()
def cm(r, n, dt=torch.float32): return torch.empty((n, r), dtype=dt, device=dev).T
for dt in (torch.float32, torch.float64):
  for K, lb in ((8, 20), (16, 20), (32, 18), (64, 16)):
    N = 1 << lb
    for kind, cls in (('vec_corr', bj.VecCorrBijector), ('pd_vec', bj.PDVecBijector)):
        b = cls()
        n = b._n(K)
        y = cm(n, N, dt); 
        L.check(ctx.h, lib.bjx_fill_normal(ctx.h, 0 if dt == torch.float32 else 1, y.data_ptr(), n, N, 0, 1, 0.0, min(0.6, 1.6 / math.sqrt(K))), 'fill')
        X = bj.transform(bj.inverse(b), y)
        es = 4 if dt == torch.float32 else 8
        for name, fn, bps in (('fwd', lambda: bj.with_logabsdet_jacobian(b, X, per_sample=True), (K * K + n + 1) * es), ('inv', lambda: bj.with_logabsdet_jacobian(bj.inverse(b), y, per_sample=True), (K * K + n + 1) * es)):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            lib.bjx_kernel_time_begin(ctx.h)
            for _ in range(5): fn()
            ms, cnt = C.c_float(0), C.c_int(0)
            lib.bjx_kernel_time_end(ctx.h, C.byref(ms), C.byref(cnt))
            k = ms.value / 5
            print(f"{str(dt)[6:]:8s} {kind:9s} {name} K={K:3d} N=2^{lb}: {k:8.4f} ms  {N / k / 1e3:9.1f} Msamp/s  {bps * N / k / 1e6:8.1f} GB/s  {bps * N / k / 1e6 / 80:5.1f} %")
PY
exit 0
