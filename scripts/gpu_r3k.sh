#!/bin/bash
# round 3, call K: BN training pullback, f64 VecCholesky chunk kernels at K = 64, thresholds; two ranks on one GPU bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
echo "== f64 rows"; python scripts/bench_f64.py 2>/dev/null | grep "^|" | tee $O/f64_rows.md
echo "== two ranks on one GPU (gloo)"
BJX_BENCH_BACKEND=gloo BJX_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $O/two_ranks_one_gpu.json 2> $O/two_ranks.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3k/two_ranks_one_gpu.json').read().strip().splitlines()[-1])
    print('n_gpus', d['n_gpus'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'rows', [(r.get('workload'), round(r.get('value',0),1)) for r in d.get('rows',[])], 'strong', [(r.get('workload'), round(r.get('value',0),1)) for r in d.get('strong_scaling',[])])
except Exception as e:
    print('two-rank bench failed', e); print(open('gpurun_out/r3k/two_ranks.err').read()[-1500:])
PY
exit 0
