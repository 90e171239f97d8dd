#!/bin/bash
# usage: gpu_ab_libs.sh <workload> <reps> lib1.so lib2.so ...   (same-call A/B of several builds; "-" = the in-tree library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
WL=$1; REPS=$2; shift 2
for r in $(seq 1 $REPS); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset BJX_LIB_PATH; else export BJX_LIB_PATH=$PWD/$L; fi
    python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-rows 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$L', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms', d.get('roofline', {}).get('kernel_ms'), 'frac', d.get('roofline', {}).get('frac'))"
  done
done
