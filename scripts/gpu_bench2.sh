#!/bin/bash
# two ranks on one GPU over gloo: the multi-rank code path of bench.py (weak headline + rows + strong-scaling sub-lines)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/bench2
export BJX_BENCH_BACKEND=gloo BJX_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench2/line.json 2> gpurun_out/bench2/err.txt
echo "rc=$?"; tail -3 gpurun_out/bench2/err.txt | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench2/line.json').read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'value', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'cpu', d['cpu_baseline'])
for r in d.get('rows', []): print(' row', r.get('workload'), r.get('error') or (round(r['value'],1), round(r['roofline']['frac'],3)))
for r in d.get('strong_scaling', []): print(' strong', r.get('workload'), r.get('error') or (round(r['value'],1), round(r['ms_per_step'],4), r['config'].get('batch_per_gpu')))
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 5 --warmup 2 --scaling strong --no-rows 2>/dev/null | tail -1 | cut -c1-400
exit 0
