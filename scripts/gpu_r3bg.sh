#!/bin/bash
# round 3, call BG: row slabs in the functor skeleton at 65 ... 128 packs per column
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bg
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "batchnorm or coupling or stacked or permute or rqs or COL_ or covers" 2>&1 | tail -30 | tee gpurun_out/r3bg/pytest.txt
BJX_BENCH_DIMS=257,300,333,500 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Coupling\|Batch\|Permute\|RQS" | tee gpurun_out/r3bg/tall_flows.md
BJX_COL_SLAB=0 BJX_BENCH_DIMS=300,500 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Coupling\|Batch\|Permute" | tee gpurun_out/r3bg/noslab.md
exit 0
