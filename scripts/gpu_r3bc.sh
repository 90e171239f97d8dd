#!/bin/bash
# round 3, call BC: tail rows behind a template flag, Stacked slabs from 257 rows, RQS slabs at odd heights
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "radial or batchnorm or coupling or rqs or permute or stacked or COL_UNALIGNED or STACKED_SLAB or RQS_SLAB or covers" 2>&1 | tail -30 | tee gpurun_out/r3bc/pytest.txt
BJX_BENCH_DIMS=63,101,201,252,333,500,1000,1001 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | tee gpurun_out/r3bc/tall_flows.md
BJX_STACKED_SLAB=128 BJX_BENCH_DIMS=333,500,1000 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep Stacked | tee gpurun_out/r3bc/stacked_slab128.md
exit 0
