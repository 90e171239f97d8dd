#!/bin/bash
# round 3, call D: buffer-placement sweep (BJX_BENCH_PAD = displacement step of the k-th large buffer) over the BASELINE configs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3d; O=gpurun_out/r3d
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  frac %.3f  kernel_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['ms_per_step']))"; }
for wl in c3 c2 c4 c5a c5b; do
  for pad in 0 256 1024 4096 69632 0 4096; do
    echo -n "$wl pad=$pad : "; BJX_BENCH_PAD=$pad b --workload $wl
  done
done 2>&1 | tee $O/placement.txt
exit 0
