#!/bin/bash
# round 3, call BD: tail rows as one more unit of the same functor call (colgroup_tail_kernel)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "batchnorm or coupling or rqs or permute or stacked or COL_UNALIGNED or STACKED_SLAB or covers" 2>&1 | tail -30 | tee gpurun_out/r3bd/pytest.txt
BJX_BENCH_DIMS=101,201,252,333,1001 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "Planar\|Radial\|Permute\|exp∘" | tee gpurun_out/r3bd/tall_flows.md
BJX_COL_UNALIGNED_MIN=48 BJX_BENCH_DIMS=49,63,77 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Batch\|Coupling\|Stacked" | tee gpurun_out/r3bd/min48.md
BJX_BENCH_DIMS=49,77 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Batch\|Coupling\|Stacked" | tee gpurun_out/r3bd/min96.md
exit 0
