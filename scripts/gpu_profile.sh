#!/bin/bash
# Collects one round's evidence on the GPU box (everything under gpurun_out/<TAG>/):
#   pytest -m gpu, one bench line per workload (with the CPU baseline), rocprofv3 --kernel-trace --stats
#   per workload, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only) for the
#   dominant kernel of every workload.  scripts/collect_profiles.py turns it into profiles/.
# usage: gpu_profile.sh <TAG> [workloads...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r01}; shift
WLS=${@:-c2 c2v c3 c4 c5a c5b}
O=$R/gpurun_out/$TAG
mkdir -p $O
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
for wl in $WLS; do
  echo "== bench $wl"; timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json; cut -c1-200 $O/bench_$wl.json
done
for wl in $WLS; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $O/rocprof_$wl.log 2>&1 )
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp "$f" $O/${wl}_kernel_stats.csv; echo "-- kernel stats $wl"; head -4 "$f" | cut -c1-160; }
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${wl}_$c -o $wl -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $O/rocprof_pmc_${wl}_$c.log 2>&1 )
    f=$(ls $O/pmc_${wl}_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${wl}_pmc_$c.csv
  done
  rm -rf $O/prof_$wl $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE
done
ls -la $O | head -60
exit 0
