#!/bin/bash
# round 3, call AB: sweep of groups per block in the mixed walker
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export BJX_BENCH_DIMS=2,3,5,10
for g in 1 2 3 4 6 8; do echo "--- BJX_MIXED_GPB=$g"; BJX_MIXED_GPB=$g python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘"; done
exit 0
