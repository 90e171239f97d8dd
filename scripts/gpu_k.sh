#!/bin/bash
# usage: gpu_k.sh "<pytest -k expr>" "<workloads>"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "$1" > gpurun_out/pytest_k.txt 2>&1; tail -3 gpurun_out/pytest_k.txt
grep -E "^FAILED|^ERROR|Mismatched|Max absolute|Max relative|Error" gpurun_out/pytest_k.txt | head -30
b() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for wl in $2; do echo -n "$wl : "; b --workload $wl; done
exit 0
