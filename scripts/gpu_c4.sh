#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "planar" > gpurun_out/pytest_planar.txt 2>&1; tail -3 gpurun_out/pytest_planar.txt
grep -E "^FAILED|Mismatched|Max absolute|Max relative" gpurun_out/pytest_planar.txt | head -30
b() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; }
echo -n "c4 reg: "; b --workload c4
echo -n "c4 tile: "; BJX_PLANAR_REG=0 b --workload c4
exit 0
