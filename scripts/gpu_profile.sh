#!/bin/bash
# Collects the round's evidence: bench lines for every workload, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE PMC passes for the headline kernel.  Output under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r01}
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu_full.txt 2>&1; tail -3 gpurun_out/pytest_gpu_full.txt
echo "== bench (default = c2)"; python bench.py 2>/dev/null | tee gpurun_out/${TAG}_bench_c2.json
for wl in c2v copy c3 c4 c5a c5b; do echo "== bench $wl"; python bench.py --workload $wl --steps 10 --warmup 3 2>/dev/null | tee gpurun_out/${TAG}_bench_$wl.json; done
for wl in c2 c3 c4 c5a c5b; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_$wl.log 2>&1 )
  f=$(ls gpurun_out/${TAG}_prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { echo "-- kernel stats $wl"; head -6 "$f" | cut -c1-220; }
done
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_pmc_$c.log 2>&1 )
  f=$(ls gpurun_out/${TAG}_pmc_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && grep chain_flat "$f" | head -2 | cut -c1-400
done
exit 0
