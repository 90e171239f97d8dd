#!/bin/bash
# round 3, call U: full suite (-x removed) + pullbacks at K = 100 with the G-lane kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3u; O=gpurun_out/r3u
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
echo "--- K = 100, 128 with BJX_SIMPLEX_VJP_TALL_MIN=65 BJX_SEQ_TALL_MIN=65"
BJX_SIMPLEX_VJP_TALL_MIN=65 BJX_SEQ_TALL_MIN=65 BJX_BENCH_KS=80,100,128 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
echo "--- default"
BJX_BENCH_KS=80,100,128 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
exit 0
