#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_full.txt 2>&1; tail -5 gpurun_out/pytest_gpu_full.txt
grep -E "Mismatched|Max absolute|Max relative|AssertionError|^E  " gpurun_out/pytest_gpu_full.txt | head -40
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; }
for wl in copy c2 c2v c3 c4 c5a c5b; do for nt in 1 0; do
  echo -n "$wl nt=$nt : "; BJX_NT=$nt b --workload $wl
done; done 2>&1 | tee gpurun_out/bench_all.txt
exit 0
