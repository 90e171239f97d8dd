#!/bin/bash
# Second GPU pass: full parity log, chain-kernel tuning matrix, rocprof stats + PMC traffic.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_full.txt 2>&1; tail -8 gpurun_out/pytest_gpu_full.txt
grep -E "Mismatched|Max absolute|Max relative|AssertionError|^E  " gpurun_out/pytest_gpu_full.txt | head -40
b() { python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; }
echo "== tuning matrix (workload NT BPC U)"
for wl in copy c2; do for nt in 0 1; do for bpc in 4 8 16; do for u in 2 4 8; do
  echo -n "$wl nt=$nt bpc=$bpc u=$u : "; BJX_NT=$nt BJX_BPC=$bpc BJX_U=$u b --workload $wl
done; done; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tuning_c2.txt
echo "== rocprof kernel stats (c2)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rocprof_c2.log 2>&1 )
ls gpurun_out/prof_c2 | head; f=$(ls gpurun_out/prof_c2/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f"
echo "== rocprof PMC (c2)"
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_pmc_$c.log 2>&1 )
f=$(ls gpurun_out/pmc_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 "$f"; grep chain_flat "$f" | head -3)
done
exit 0
