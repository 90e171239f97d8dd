#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3c
timeout 600 python scripts/probe_c3_dir.py 2>&1 | tee gpurun_out/r3c/probe_c3_dir.txt | tail -12
exit 0
