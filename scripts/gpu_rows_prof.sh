#!/bin/bash
# usage: gpu_rows_prof.sh "<--only filter>"  -> rocprofv3 kernel stats of scripts/bench_rows.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rows_prof -o rows -- python $R/scripts/bench_rows.py --only "$1" > $R/gpurun_out/rows_prof.log 2>&1 )
tail -3 gpurun_out/rows_prof.log
f=$(ls gpurun_out/rows_prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print('%-100s calls %5s avg %10.1f us' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
exit 0
