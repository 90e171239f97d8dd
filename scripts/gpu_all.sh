#!/bin/bash
# full GPU parity suite + one bench line per workload (no CPU baseline)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_full.txt 2>&1; tail -3 gpurun_out/pytest_gpu_full.txt
grep -E "^FAILED|Mismatched|Max absolute|Max relative" gpurun_out/pytest_gpu_full.txt | head -30
b() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for wl in ${WLS:-c2 c3 c4 c5a c5b}; do echo -n "$wl : "; b --workload $wl; done
exit 0
