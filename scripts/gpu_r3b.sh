#!/bin/bash
# round 3, call B: rqs_lds_kernel without static LDS (5 blocks of 32 KiB per CU instead of 4) — parity + iters sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3b; O=gpurun_out/r3b
timeout 900 python -m pytest tests -m gpu -q -x -k "rqs or spline or c3 or coupling or finalize" -p no:cacheprovider > $O/tests.txt 2>&1
echo "tests: $(grep -E 'passed|failed' $O/tests.txt | tail -1)"; grep -E "^FAILED" $O/tests.txt | head -5
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for rep in 1 2; do
  for it in 0 16 24 32 48 64; do
    echo -n "iters=$it : "; BJX_RQS_ITERS=$it b --workload c3
  done
done
exit 0
