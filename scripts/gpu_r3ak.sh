#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export BJX_BENCH_DIMS=3,10
for e in "X=1" "BJX_PLANAR_PARAM_MFMA=0" "BJX_PLANAR_PARAM_BLOCKS=16384" "BJX_PLANAR_PARAM_BLOCKS=65536" "BJX_PLANAR_PARAM_BLOCKS=256"; do echo "--- $e"; env $e python scripts/bench_small_vjp.py 2>&1 | grep "^|" | grep "vjp_params"; done
exit 0
