#!/bin/bash
# round 3, call M: full suite incl. the env-switch battery
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -40; grep real $O/pytest_gpu.txt
grep -E "AssertionError: BJX|differs from the default|worker failed" $O/pytest_gpu.txt | head -40
exit 0
