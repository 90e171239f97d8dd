#!/bin/bash
# round 2, first GPU call: new tests first (fail fast), then the whole suite, then the default bench line (with rows)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2a; O=gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_bitexact.py tests/test_gpu_parity.py -m gpu -q -x -k "bit_exact or matrix_bij" -p no:cacheprovider > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "vector_scale" -p no:cacheprovider > $O/shard_test.txt 2>&1; tail -5 $O/shard_test.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt
grep -E "^FAILED" $O/pytest_gpu_full.txt | head -40
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2a/bench_default.json').read().strip().splitlines()[-1])
print('headline %.1f Msamp/s frac %.3f kernel_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']), 'cpu', d['cpu_baseline']['value'])
for r in d.get('rows', []):
    if 'error' in r: print(r); continue
    print('%-5s %10.1f Msamp/s frac %.3f kernel_ms %.4f step_ms %.4f cpu %s' % (r['workload'], r['value'], r['roofline']['frac'], r['roofline']['kernel_ms'], r['ms_per_step'], r.get('cpu_baseline',{}).get('value')))
PY
for wl in vcorr pdvec; do timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 > $O/bench_$wl.json 2> $O/bench_$wl.err; tail -c 300 $O/bench_$wl.err; python - $wl <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r2a/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], '%.1f Msamp/s frac %.3f kernel_ms %.4f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']), 'cpu', d['cpu_baseline']['value'])
except Exception as e: print('no line', e)
PY
done
exit 0
