#!/bin/bash
# usage: gpu_stats.sh <workload> [steps]  -> rocprofv3 kernel stats (top rows)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD; WL=$1; ST=${2:-10}
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$WL -o $WL -- python $R/bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline > $R/gpurun_out/stats_$WL.log 2>&1 )
f=$(ls gpurun_out/stats_$WL/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('%-90s calls %5s  avg %10.1f ns  min %10s  max %10s  %5s%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs'], r['Percentage']))
PY
exit 0
