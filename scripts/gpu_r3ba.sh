#!/bin/bash
# round 3, call BA: Planar register kernels on element-aligned packs (odd heights) + the flows table of r02_tall_columns.md as it stands
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ba
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -x -q -k "planar or chain or env or switch" 2>&1 | tail -5 | tee gpurun_out/r3ba/pytest.txt
BJX_BENCH_DIMS=33,63,65,101,127,129,201,255 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Planar" | tee gpurun_out/r3ba/planar_new.md
BJX_PLANAR_REG_UNALIGNED=0 BJX_BENCH_DIMS=33,63,101,201 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Planar" | tee gpurun_out/r3ba/planar_old.md
BJX_BENCH_DIMS=101,201,252,500,1000 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | tee gpurun_out/r3ba/tall_flows.md
exit 0
