#!/bin/bash
# round 3, call W: tall columns, final thresholds: Float32 and Float64 tables, walkers beside (one box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3w; O=gpurun_out/r3w
export BJX_BENCH_KS=100,160,200,256,300,500,1000,2000
echo "## Float32, 2^20 columns — shipped dispatch" > $O/tall.md
python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall.md
echo "## Float32 — walkers (BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0)" >> $O/tall.md
BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall.md
export BJX_BENCH_KS=100,200,500,1000 BJX_PROBE_DTYPE=f64 BJX_BENCH_LOG2N=19
echo "## Float64, 2^19 columns — shipped dispatch" >> $O/tall.md
python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall.md
exit 0
