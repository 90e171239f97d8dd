#!/bin/bash
# round 3, call BE: two columns in flight at 65 ... 128 units per column
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3be
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "batchnorm or coupling or rqs or permute or stacked or COL_UNALIGNED or covers" 2>&1 | tail -30 | tee gpurun_out/r3be/pytest.txt
BJX_BENCH_DIMS=77,101,300,333,500,1001 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "Planar\|Radial\|exp∘" | tee gpurun_out/r3be/tall_flows.md
exit 0
