#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2d; O=gpurun_out/r2d
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt
grep -E "^FAILED" $O/pytest_gpu_full.txt | head -40
bash scripts/gpu_r2c.sh 2>/dev/null | grep -v "^\.\|passed"
exit 0
