#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3e; O=gpurun_out/r3e
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  frac %.3f  kernel_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['ms_per_step']))"; }
{
for rep in 1 2; do
  echo -n "c3 separate : "; b --workload c3
  echo -n "c3 arena    : "; BJX_BENCH_ARENA_MB=2048 b --workload c3
  echo -n "c2 separate : "; b --workload c2
  echo -n "c2 arena    : "; BJX_BENCH_ARENA_MB=9000 b --workload c2
done
python scripts/probe_c3_dir.py 2>&1 | grep pad
} 2>&1 | tee $O/arena.txt
exit 0
