#!/bin/bash
# round 3, call P: counters of the chunked wave walker on tall columns, one column height per pass
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3p; O=$PWD/gpurun_out/r3p; R=$PWD; export TMPDIR=/tmp
export BJX_PROBE_ROWS=fwd BJX_SEQ_CHUNK_MIN=20480
rm -f $O/counters.txt
for K in ${KS:-200 256 1000}; do
export BJX_BENCH_KS=$K
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python $R/scripts/probe_tall_simplex.py > $O/pmc_$i.log 2>&1 )
  f=$(ls $O/pmc_$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" $K <<'PY' | tee -a $O/counters.txt
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'seq_chunk' in r['Kernel_Name'] or 'quad_stream' in r['Kernel_Name'] or 'seq_wave' in r['Kernel_Name']]
byk=collections.defaultdict(list)
for r in rows: byk[r['Kernel_Name']].append(r)
for kn, rows in byk.items():
  agg=collections.defaultdict(list)
  for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
  short = kn.split('namespace)::')[1][:18] + ' ' + kn.split('namespace)::')[2][:28] if kn.count('namespace)::') >= 2 else kn[:60]
  print('K', sys.argv[2], short, 'VGPR', rows[0]['VGPR_Count'], '|', '  '.join('%s %.4g' % (k.replace('SQ_',''), sum(v)/len(v)) for k, v in agg.items()))
PY
  rm -rf $O/pmc_$i
done
done
exit 0
