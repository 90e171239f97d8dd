#!/bin/bash
# round 3, last call: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel stats of the group kernels at 100 / 101 / 200 / 201 rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3bj; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o odd -- python $R/scripts/probe_odd_traffic.py > $O/pmc_$c.log 2>&1 )
  f=$(ls $O/pmc_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/odd_pmc_$c.csv; rm -rf $O/pmc_$c
done
( cd /tmp && BJX_BENCH_DIMS=101,201,500,1000 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o flows -- python $R/scripts/bench_small_dims.py > $O/flows_raw.txt 2>&1 )
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -80 "$f" > $O/tall_flows_kernel_stats.csv; rm -rf $O/prof
grep "^|" $O/flows_raw.txt > $O/tall_flows_under_rocprof.md
ls -la $O
exit 0
