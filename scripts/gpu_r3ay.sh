#!/bin/bash
# round 3, call AY: per-sample chains / Stacked at odd column heights beyond the short-column kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ay
BJX_BENCH_DIMS=17,33,63,65,127,129,255,257,333,1001 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘\|Stacked\|Planar\|BatchNorm" | tee gpurun_out/r3ay/odd_dims.md
exit 0
