#!/bin/bash
# round 3, call AA: counters of the mixed walker on short columns
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3aa; O=$PWD/gpurun_out/r3aa; R=$PWD; export TMPDIR=/tmp
for d in 2 5; do
export BJX_BENCH_DIMS=$d
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python $R/scripts/probe_small_chain.py > $O/pmc_$i.log 2>&1 )
  f=$(ls $O/pmc_$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" $d <<'PY' | tee -a $O/counters.txt
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'stacked_mixed' in r['Kernel_Name']]
agg=collections.defaultdict(list)
for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
if rows: print('dim', sys.argv[2], 'VGPR', rows[0]['VGPR_Count'], 'SGPR', rows[0]['SGPR_Count'], 'LDS', rows[0]['LDS_Block_Size'], 'grid', rows[0]['Grid_Size'], '|', '  '.join('%s %.4g' % (k.replace('SQ_',''), sum(v)/len(v)) for k, v in agg.items()))
PY
  rm -rf $O/pmc_$i
done
done
exit 0
