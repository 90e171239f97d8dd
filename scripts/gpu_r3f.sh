#!/bin/bash
# round 3, call F: does the 5-step warm-up leave the GPU below its steady clocks?  same binary, same box, warm-up sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3f; O=gpurun_out/r3f
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  frac %.3f  kernel_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['ms_per_step']))"; }
{
for wl in c3 c2 c4 c5a c5b; do
  for w in 5 50 300 5 300; do
    echo -n "$wl warmup=$w : "; b --workload $wl --warmup $w
  done
done
} 2>&1 | tee $O/warmup.txt
exit 0
