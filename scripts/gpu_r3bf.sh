#!/bin/bash
# round 3, call BF: Stacked slabs of 64 packs (one pack per lane, four columns in flight), element-aligned packs inside row windows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bf
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "batchnorm or coupling or stacked or COL_UNALIGNED or STACKED_SLAB or covers" 2>&1 | tail -30 | tee gpurun_out/r3bf/pytest.txt
BJX_BENCH_DIMS=257,300,333,500,1000,1001 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Stacked\|Batch" | tee gpurun_out/r3bf/tall_flows.md
exit 0
