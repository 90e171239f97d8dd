#!/bin/bash
# round 3, call G: full GPU suite on the new binary (fenced in-kernel finalize as the small-grid default, two-rank test, tighter
# matrix tolerances, C5b at 2^20), the default bench line, an old/new A/B of the chain kernels, latency rows, and the row tables
# re-measured with the clock-settling pre-roll.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3g; O=gpurun_out/r3g
( time timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -10; grep real $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3g/bench_default.json').read().strip().splitlines()[-1])
print('headline c2: %.0f Msamp/s frac %.3f ms/step %.4f preroll %s' % (d['value'], d['roofline']['frac'], d['ms_per_step'], d.get('preroll')))
for r in d.get('rows', []):
    if 'error' in r: print(r); continue
    print('  %-4s value %10.2f frac(kernel) %.3f kernel_ms %.4f step_ms %.4f wall-frac %.3f' % (r['workload'], r['value'], r['roofline']['frac'], r['roofline']['kernel_ms'], r['ms_per_step'], r['roofline']['algorithmic_bytes_per_launch']/ (r['ms_per_step']*1e-3)/8e12), r.get('us_per_call',''))
for g in d.get('graph_step', []): print('  graph', g.get('workload'), g.get('log2_batch_per_gpu'), 'eager %.4f ms graph %.4f ms' % (g.get('eager_ms_per_step',-1), g.get('graph_ms_per_step',-1)))
PY
echo "--- A/B old (two follow-up launches for every grid) vs new"
bash scripts/gpu_ab_libs.sh c2 2 tmp_ab/old.so - 2>/dev/null
bash scripts/gpu_ab_libs.sh c1 2 tmp_ab/old.so - 2>/dev/null
bash scripts/gpu_ab_libs.sh c5a 2 tmp_ab/old.so - 2>/dev/null
bash scripts/gpu_ab_libs.sh c4 1 tmp_ab/old.so - 2>/dev/null
echo "--- rows"
python scripts/bench_rows.py 2>/dev/null | grep "^|" > $O/rows.md; wc -l $O/rows.md
python scripts/bench_f64.py 2>/dev/null | grep "^|" > $O/f64_rows.md; wc -l $O/f64_rows.md
bash scripts/gpu_small_sizes.sh r3g > /dev/null 2>&1; wc -l $O/small_sizes.md
( BJX_BENCH_LOG2N=20 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^|" ) > $O/tall_simplex.md; wc -l $O/tall_simplex.md
python scripts/bench_latency.py 2>/dev/null | grep "^|" > $O/latency.md; cat $O/latency.md
exit 0
