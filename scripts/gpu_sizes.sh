#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
b() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['ms_per_step']))"; }
for lb in 16 18 20; do echo -n "c5b 2^$lb : "; b --workload c5b --log2-batch $lb; done
for lb in 20 22; do echo -n "c5a 2^$lb : "; b --workload c5a --log2-batch $lb; done
