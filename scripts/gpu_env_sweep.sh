#!/bin/bash
# usage: gpu_env_sweep.sh <workload> <reps> <ENVVAR> v1 v2 ...   (same-call sweep of one environment switch)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
WL=$1; REPS=$2; VAR=$3; shift 3
for r in $(seq 1 $REPS); do
  for v in "$@"; do
    env $VAR=$v python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-rows 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms', d.get('roofline', {}).get('kernel_ms'), 'frac', d.get('roofline', {}).get('frac'))"
  done
done
