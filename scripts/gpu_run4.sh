#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider -k "chain or shard" > gpurun_out/pytest_gpu_chain.txt 2>&1; tail -3 gpurun_out/pytest_gpu_chain.txt
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; }
for wl in copy c2 c2v; do for u in 1 2 4; do for pf in 0 1; do
  echo -n "$wl U=$u PF=$pf : "; BJX_U=$u BJX_PF=$pf b --workload $wl
done; done; done 2>&1 | tee gpurun_out/chain_variants2.txt
exit 0
