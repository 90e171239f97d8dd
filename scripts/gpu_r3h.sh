#!/bin/bash
# round 3, call H: full GPU suite (general-size matrix kernels, fixed K = 1 case), then the in-kernel finalize hand-off A/B on the
# launch-bound shapes: two follow-up launches (mode 0) | sc1 store + drained flag (mode 1/2, no fences) | agent release/acquire fences
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3h; O=gpurun_out/r3h
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
b() { python bench.py --no-cpu-baseline --no-rows --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.2f us  kernel %.2f us  region %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['stream_region_ms_per_step']*1e3))"; }
{
for rep in 1 2; do
for wl in "c1" "c2 --log2-batch 16" "c2 --log2-batch 20" "c5a --log2-batch 16" "c4 --log2-batch 16"; do
  echo -n "$wl | two-pass      : "; BJX_INKERNEL_FIN=0 b --workload $wl
  echo -n "$wl | sc1 + drained : "; BJX_INKERNEL_FIN=2 b --workload $wl
  echo -n "$wl | agent fences  : "; BJX_INKERNEL_FIN=2 BJX_INKERNEL_FIN_FENCES=1 b --workload $wl
done
done
} 2>&1 | tee $O/finalize_ab.txt
exit 0
