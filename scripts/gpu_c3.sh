#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/c3; O=gpurun_out/c3
timeout 900 python -m pytest tests -m gpu -q -k "rqs or spline or c3 or nan_inputs or coupling or shape_sweep" -p no:cacheprovider > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt | tail -1; grep -E "^FAILED" $O/tests.txt | head
b() { python bench.py --no-cpu-baseline --no-rows --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for i in 1 2; do echo -n "c3 : "; b --workload c3; done
python scripts/bench_rows.py --only "RQS" 2>/dev/null | grep "^|"
exit 0
