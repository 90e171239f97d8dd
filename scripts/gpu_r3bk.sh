#!/bin/bash
# round 3, closing call: ordinary against streaming stores in the odd-height kernels (time and WRITE_SIZE), and the whole GPU suite on the final library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3bk; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
for nt in 0 1; do echo "BJX_UNAL_NT=$nt"; BJX_UNAL_NT=$nt BJX_BENCH_DIMS=101,201 timeout 300 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Planar\|Batch\|Stacked\|Coupling"; done | tee $O/nt_ab.md
for nt in 0 1; do
  ( cd /tmp && BJX_UNAL_NT=$nt BJX_BENCH_DIMS=101,201 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$nt -o odd -- python $R/scripts/probe_odd_traffic.py > $O/pmc_$nt.log 2>&1 )
  f=$(ls $O/pmc_$nt/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/odd_pmc_WRITE_SIZE_nt$nt.csv; rm -rf $O/pmc_$nt
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2>$O/bench_default.err; cut -c1-200 $O/bench_default.json
exit 0
