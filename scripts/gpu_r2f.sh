#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2f; O=gpurun_out/r2f
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu_full.txt | tail -1
grep -E "^FAILED" $O/pytest_gpu_full.txt | head -40
exit 0
