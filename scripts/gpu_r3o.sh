#!/bin/bash
# round 3, call O: f64 sub-lines in the default bench line, lean f64 logit; full suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3o; O=gpurun_out/r3o
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3o/bench_default.json').read().strip().splitlines()[-1])
print('headline c2: %.0f Msamp/s frac %.3f ms/step %.4f' % (d['value'], d['roofline']['frac'], d['ms_per_step']))
for r in d.get('rows', []):
    if 'error' in r: print(r); continue
    print('  %-7s value %10.2f frac(kernel) %.3f kernel_ms %.4f step_ms %.4f cpu %s' % (r['workload'], r['value'], r['roofline']['frac'], r['roofline']['kernel_ms'], r['ms_per_step'], (r.get('cpu_baseline') or {}).get('value')))
PY
python scripts/bench_f64.py 2>/dev/null | grep "^|" | tee $O/f64_rows.md
exit 0
