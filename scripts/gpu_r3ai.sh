#!/bin/bash
# round 3, call AI: full suite + the small-size tables with the short-column kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ai; O=gpurun_out/r3ai
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
bash scripts/gpu_small_sizes.sh r3ai > /dev/null 2>&1; wc -l $O/small_sizes.md
exit 0
