#!/bin/bash
# usage: gpu_env.sh <workload> "ENV1=.. ENV2=.." ["ENV..."]...   -> one bench line per env set
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
WL=$1; shift
b() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 --workload $WL 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for e in "$@"; do echo -n "$WL [$e] : "; ( export $e; b ); done
exit 0
