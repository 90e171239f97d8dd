#!/bin/bash
# usage: gpu_prof_one.sh <TAG> <workload> -> rocprofv3 kernel stats of bench.py --workload <wl> (no rows, no CPU leg)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; R=$PWD; TAG=$1; WL=$2; O=$R/gpurun_out/$TAG; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$WL -o $WL -- python $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $O/rocprof_$WL.log 2>&1 )
f=$(ls $O/prof_$WL/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp "$f" $O/${WL}_kernel_stats.csv; head -6 "$f" | cut -c1-200; }
tail -1 $O/rocprof_$WL.log | cut -c1-300
rm -rf $O/prof_$WL
exit 0
