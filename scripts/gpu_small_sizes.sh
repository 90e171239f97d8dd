#!/bin/bash
# Regenerates gpurun_out/<TAG>/small_sizes.md only (the small-shape tables of gpu_profile.sh).  usage: gpu_small_sizes.sh <TAG>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
( python scripts/bench_small_sizes.py 2>/dev/null | grep "^|"; echo; python scripts/bench_matrix_small.py 2>/dev/null | grep "^|"; echo; python scripts/bench_small_dims.py 2>/dev/null | grep "^|"; echo; python scripts/bench_small_vjp.py 2>/dev/null | grep "^|" ) > $O/small_sizes.md; wc -l $O/small_sizes.md
