#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3aj; O=$PWD/gpurun_out/r3aj; R=$PWD; export TMPDIR=/tmp
for d in 3 10; do
( cd /tmp && BJX_BENCH_DIMS=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$d -o p -- python $R/scripts/probe_planar_params.py > $O/log_$d.txt 2>&1 )
f=$(ls $O/prof_$d/*kernel_stats.csv 2>/dev/null | head -1); echo "--- dim $d"; [ -n "$f" ] && head -12 "$f" | cut -c1-220
rm -rf $O/prof_$d
done
exit 0
