#!/bin/bash
# round 3, call T: full suite + default bench + tall columns table with the G-lane kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3t; O=gpurun_out/r3t
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3t/bench_default.json').read().strip().splitlines()[-1])
print('headline c2: %.0f Msamp/s frac %.3f ms/step %.4f' % (d['value'], d['roofline']['frac'], d['ms_per_step']))
for r in d.get('rows', []):
    if 'error' in r: print(r); continue
    print('  %-7s value %10.2f frac(kernel) %.3f' % (r['workload'], r['value'], r['roofline']['frac']))
PY
BJX_BENCH_KS=100,160,200,256,300,500,1000,2000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee $O/tall.md
exit 0
