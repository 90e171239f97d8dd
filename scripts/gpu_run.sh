#!/bin/bash
# The ONE GPU-side script (rounds 1-3 had 86 one-off gpu_*.sh files: archived in profiles/lab_scripts_r01_r03.md).
# Run it on the GPU box through gpurun, e.g.   gpurun --timeout 1500 -- 'scripts/gpu_run.sh profile r04'
#
#   gpu_run.sh tests [pytest args]            pytest -m gpu (whole suite by default)                      -> gpurun_out/pytest_gpu.txt
#   gpu_run.sh bench [bench.py args]          the driver's line                                          -> gpurun_out/bench.json
#   gpu_run.sh rows <filter> [ENV=V ...]      scripts/bench_rows.py --only <filter>, once per env setting (same box, same call: an A/B)
#   gpu_run.sh ab <filter> <old.so> <new.so>  the same rows with two builds of the library (BJX_LIB_PATH)
#   gpu_run.sh pmc <workload> <kernel-substring> "<counters>" ["<counters>" ...]   rocprofv3 --pmc passes (kernel-trace only), per-kernel means
#   gpu_run.sh rowprof <filter> ["<counters>" ...]  rocprofv3 kernel stats (+ --pmc passes, kernel-trace only) of scripts/bench_rows.py --only <filter>
#   gpu_run.sh stats <workload>               rocprofv3 --kernel-trace --stats of one bench workload      -> gpurun_out/<wl>_kernel_stats.csv
#   gpu_run.sh profile <TAG> [workloads...]   the round's evidence, everything under gpurun_out/<TAG>/ (scripts/collect_profiles.py <TAG>
#                                             turns it into profiles/): pytest -m gpu, the default bench line, one bench line + kernel
#                                             stats + FETCH_SIZE / WRITE_SIZE passes per workload, the rows tables, .so size and first-call latency
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
MODE=${1:-tests}; shift

pmc_summary() {   # <counter_collection.csv> <kernel-substring>
python - "$1" "$2" <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
byk=collections.defaultdict(list)
for r in rows: byk[r['Kernel_Name']].append(r)
for kn, rows in byk.items():
  agg=collections.defaultdict(list)
  for r in rows: agg[r['Counter_Name']].append(float(r['Counter_Value']))
  print('kernel:', rows[0]['Kernel_Name'][:150], 'VGPR', rows[0]['VGPR_Count'], 'SGPR', rows[0]['SGPR_Count'], 'LDS', rows[0]['LDS_Block_Size'], 'grid', rows[0]['Grid_Size'], 'wg', rows[0]['Workgroup_Size'])
  for k,v in agg.items(): print('  %-28s %16.1f  (n=%d)' % (k, sum(v)/len(v), len(v)))
PY
}

case $MODE in
tests)
  [ $# -eq 0 ] && set -- tests
  rm -f gpurun_out/vjp_errors.jsonl gpurun_out/matrix_vjp_errors.jsonl
  ( time timeout 1500 python -m pytest --maxfail=25 "$@" -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.txt 2>&1; tail -8 gpurun_out/pytest_gpu.txt ;;
bench)
  timeout 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; wc -c gpurun_out/bench.json; cut -c1-600 gpurun_out/bench.json ;;
rows)
  F=$1; shift
  [ $# -eq 0 ] && set -- "BJX_NOOP=0"
  for e in "$@"; do
    echo "== [$e]"; env $e python scripts/bench_rows.py --only "$F" --steps 10 2>/dev/null | grep "^| " | grep -v "^| row\|^|---" | cut -d'|' -f2,4,8,10
  done ;;
ab)
  F=$1; OLD=$2; NEW=$3
  for so in $OLD $NEW; do
    echo "== $so"; BJX_LIB_PATH=$R/$so python scripts/bench_rows.py --only "$F" --steps 10 2>/dev/null | grep "^| " | grep -v "^| row\|^|---" | cut -d'|' -f2,4,8,10
  done ;;
pmc)
  WL=$1; KS=$2; shift 2; i=0
  for set in "$@"; do
    i=$((i+1))
    ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${WL}_$i -o p -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-rows > $R/gpurun_out/pmc_${WL}_$i.log 2>&1 )
    f=$(ls gpurun_out/pmc_${WL}_$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && pmc_summary "$f" "$KS"
  done ;;
rowprof)
  F=$1; shift; i=0
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rowprof_0 -o p -- python $R/scripts/bench_rows.py --only "$F" --steps 10 > $R/gpurun_out/rowprof_0.txt 2>&1 )
  grep "^| " gpurun_out/rowprof_0.txt | grep -v "^| row\|^|---" | cut -d'|' -f2,4,8,10
  f=$(ls gpurun_out/rowprof_0/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-260
  for set in "$@"; do
    i=$((i+1))
    ( cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/rowprof_$i -o p -- python $R/scripts/bench_rows.py --only "$F" --steps 3 > /dev/null 2>&1 )
    f=$(ls gpurun_out/rowprof_$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && pmc_summary "$f" "kernel"
  done ;;
stats)
  WL=$1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -o $WL -- python $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $R/gpurun_out/rocprof_$WL.log 2>&1 )
  f=$(ls gpurun_out/prof_$WL/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp "$f" gpurun_out/${WL}_kernel_stats.csv; head -8 "$f" | cut -c1-200; }; rm -rf gpurun_out/prof_$WL ;;
profile)
  TAG=${1:-r06}; shift
  WLS=${@:-c2 c2v c3 c4 c5a c5b vcorr pdvec c2_f64 c4_f64}
  O=$R/gpurun_out/$TAG; mkdir -p $O
  # the binary this evidence belongs to (tests/test_profiles_fresh.py compares kernel names with the .so in the tree)
  sha256sum bijectors.jl_amd/libbjx_hip.so | cut -c1-16 > $O/lib_sha16.txt; stat -c %s bijectors.jl_amd/libbjx_hip.so > $O/lib_bytes.txt
  python scripts/probe_first_call.py > $O/first_call.txt 2>&1
  cat $O/first_call.txt
  rm -f gpurun_out/matrix_vjp_errors.jsonl gpurun_out/vjp_errors.jsonl $O/kernel_trace_timed.jsonl
  echo "== default bench line (what the driver runs: first on a fresh box)"; timeout 900 python bench.py > $O/bench_default.json 2>$O/bench_default.err; wc -c $O/bench_default.json; cp gpurun_out/bench_detail.json $O/bench_default_detail.json 2>/dev/null
  echo "== pytest -m gpu"; ( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
  cp gpurun_out/matrix_vjp_errors.jsonl gpurun_out/vjp_errors.jsonl $O/ 2>/dev/null
  for wl in $WLS; do
    timeout 600 python bench.py --workload $wl --no-rows --steps 20 --warmup 5 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json; echo "== bench $wl: $(cut -c1-120 $O/bench_$wl.json)"
  done
  echo "== rows tables (table and kernel stats from the SAME process)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows -o rows -- python $R/scripts/bench_rows.py > $O/rows_raw.txt 2>&1 )
  grep "^|" $O/rows_raw.txt > $O/rows.md; f=$(ls $O/prof_rows/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -60 "$f" > $O/rows_kernel_stats.csv; rm -rf $O/prof_rows; wc -l $O/rows.md
  timeout 600 python scripts/bench_f64.py 2>/dev/null | grep "^|" > $O/f64_rows.md; wc -l $O/f64_rows.md
  # (round 5's height sweeps — ordered_tall, planar_heights, small_dims — stay in profiles/ as r05_*: those kernels did not change in round 6)
  { echo "# host time per call, launch plans ON (default)"; timeout 300 python scripts/probe_host_overhead.py --calls 2000 --top 8 2>/dev/null | grep -v amdgpu.ids;
    echo; echo "# the same with the plans OFF (--no-plans: the general path of rounds 1-5)"; timeout 300 python scripts/probe_host_overhead.py --calls 2000 --top 4 --no-plans 2>/dev/null | grep -v amdgpu.ids; } > $O/host_overhead.txt
  timeout 300 python scripts/probe_small_calls.py 2>/dev/null | grep "^|" > $O/small_calls.md
  { hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_lds scripts/probe_lds_atomics.hip 2>/dev/null && /tmp/probe_lds; } > $O/lds_atomics.txt 2>&1
  bash scripts/ab_c3.sh c3 2 2>/dev/null > $O/c3_table_policy.txt
  # issue-side counters of the two BASELINE kernels under their roofline, ON THE SHIPPED LIBRARY (VERDICT r05 missing #7): three passes each
  for wl in c3 c4_f64; do
    ks=rqs_lds_kernel; [ $wl = c4_f64 ] && ks=planar_mfma64_kernel
    i=0
    for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"; do
      i=$((i+1))
      ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmcq_${wl}_$i -o p -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-rows > $O/rocprof_pmcq_${wl}_$i.log 2>&1 )
      f=$(ls $O/pmcq_${wl}_$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && pmc_summary "$f" "$ks" > $O/${wl}_sq_pmc_$i.txt
      rm -rf $O/pmcq_${wl}_$i
    done
  done
  for wl in $WLS; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $O/rocprof_$wl.log 2>&1 )
    f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp "$f" $O/${wl}_kernel_stats.csv; echo "-- kernel stats $wl"; head -3 "$f" | cut -c1-160; }
    # the same trace, timed launches only (after bench.py's pre-roll): median / min / mean — what the line's kernel_ms must agree with
    f=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python scripts/trace_stats.py "$f" $wl 60 >> $O/kernel_trace_timed.jsonl
    for c in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${wl}_$c -o $wl -- python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-rows > $O/rocprof_pmc_${wl}_$c.log 2>&1 )
      f=$(ls $O/pmc_${wl}_$c/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/${wl}_pmc_$c.csv
    done
    rm -rf $O/prof_$wl $O/pmc_${wl}_FETCH_SIZE $O/pmc_${wl}_WRITE_SIZE
  done
  ls $O | head -80 ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
exit 0
