#!/bin/bash
# usage: gpu_ab_probe.sh <reps> lib1.so lib2.so ...  — same-call A/B of several builds on the chain probes and the headline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPS=$1; shift
for r in $(seq 1 $REPS); do
  for L in "$@"; do
    export BJX_LIB_PATH=$PWD/$L
    echo "== $L"
    python scripts/probe_logpdf.py 2>&1 | grep " ms "
    python scripts/bench_rows.py --only "exp∘Shift∘Scale (,Logit(0,1),vjp_params(exp,Stacked(exp|Logit" 2>&1 | grep "^| " | grep -v "^| row" | cut -c1-150
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rows 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('headline ms_per_step %.4f' % d['ms_per_step'], 'frac', d.get('roofline', {}).get('frac'))"
  done
done
