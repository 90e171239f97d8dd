#!/bin/bash
# usage: gpu_pmc_rows.sh "<--only filter>" <kernel-substring> "<counters...>" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; ONLY=$1; KS=$2; shift 2
i=0
for set in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcr_$i -o p -- python $R/scripts/bench_rows.py --only "$ONLY" --steps 2 > $R/gpurun_out/pmcr_$i.log 2>&1 )
  f=$(ls gpurun_out/pmcr_$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" "$KS" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
byk=collections.defaultdict(list)
for r in rows: byk[r['Kernel_Name']].append(r)
for kn, rows in byk.items():
  agg=collections.defaultdict(list)
  for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
  print('kernel:', rows[0]['Kernel_Name'][:150], 'VGPR', rows[0]['VGPR_Count'], 'LDS', rows[0]['LDS_Block_Size'], 'grid', rows[0]['Grid_Size'], 'wg', rows[0]['Workgroup_Size'])
  for k,v in agg.items(): print('  %-28s %16.1f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
done
exit 0
