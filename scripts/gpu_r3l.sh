#!/bin/bash
# round 3, call L: full suite (full-covariance base, watchdog, BN pullback), rows re-measured (knot pullback look-ahead)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3l; O=gpurun_out/r3l
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
python scripts/bench_rows.py --only "RQS,rand,logpdf,Stacked,Scale(64" 2>/dev/null | grep "^|" | tee $O/rows_some.md
exit 0
