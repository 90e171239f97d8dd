#!/bin/bash
# round 3, call A: short-lived RQS blocks (BJX_RQS_SL=2|4) vs the looping kernel — parity, then same-box A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a; O=gpurun_out/r3a
for sl in 2 4; do
  BJX_RQS_SL=$sl timeout 900 python -m pytest tests -m gpu -q -x -k "rqs or spline or c3 or coupling" -p no:cacheprovider > $O/tests_sl$sl.txt 2>&1
  echo "SL=$sl: $(grep -E 'passed|failed' $O/tests_sl$sl.txt | tail -1)"; grep -E "^FAILED" $O/tests_sl$sl.txt | head -5
done
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for rep in 1 2 3; do
  echo -n "loop : "; b --workload c3
  echo -n "SL=2 : "; BJX_RQS_SL=2 b --workload c3
  echo -n "SL=4 : "; BJX_RQS_SL=4 b --workload c3
done
echo -n "c2   : "; b --workload c2
exit 0
