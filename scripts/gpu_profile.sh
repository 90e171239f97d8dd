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
WLS=${@:-c2 c2v c3 c4 c5a c5b vcorr pdvec}
O=$R/gpurun_out/$TAG
mkdir -p $O
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
echo "== default bench line (headline + rows, what the driver runs)"; timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; cut -c1-300 $O/bench_default.json
for wl in $WLS; do
  echo "== bench $wl"; timeout 600 python bench.py --workload $wl --no-rows --steps 20 --warmup 5 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json; cut -c1-200 $O/bench_$wl.json
done
echo "== rows table under rocprofv3 (the table and the kernel stats come from the SAME process)"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows -o rows -- python $R/scripts/bench_rows.py > $O/rows_raw.txt 2>&1 )
grep "^|" $O/rows_raw.txt > $O/rows.md; f=$(ls $O/prof_rows/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -60 "$f" > $O/rows_kernel_stats.csv; rm -rf $O/prof_rows; wc -l $O/rows.md
python scripts/bench_f64.py 2>/dev/null | grep "^|" > $O/f64_rows.md; wc -l $O/f64_rows.md
[ -x scripts/f64math_bench.co ] && ./scripts/f64math_bench.co > $O/f64math_bench.txt 2>&1
echo "== small per-sample sizes (K = 2 ... 16 matrix / Cholesky blocks, dim = 2 ... 10 columns)"
( python scripts/bench_small_sizes.py 2>/dev/null | grep "^|"; echo; python scripts/bench_matrix_small.py 2>/dev/null | grep "^|"; echo; python scripts/bench_small_dims.py 2>/dev/null | grep "^|"; echo; python scripts/bench_small_vjp.py 2>/dev/null | grep "^|" ) > $O/small_sizes.md; wc -l $O/small_sizes.md
( for m in 0 2 1 4 0; do echo -n "c4 BJX_PLANAR_MFMA=$m : "; BJX_PLANAR_MFMA=$m python bench.py --workload c4 --no-rows --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"; done ) > $O/planar_mfma_ab.txt 2>&1; cat $O/planar_mfma_ab.txt
for wl in $WLS; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $O/rocprof_$wl.log 2>&1 )
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp "$f" $O/${wl}_kernel_stats.csv; echo "-- kernel stats $wl"; head -4 "$f" | cut -c1-160; }
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${wl}_$c -o $wl -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-rows > $O/rocprof_pmc_${wl}_$c.log 2>&1 )
    f=$(ls $O/pmc_${wl}_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${wl}_pmc_$c.csv
  done
  rm -rf $O/prof_$wl $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE
done
ls -la $O | head -60
exit 0
