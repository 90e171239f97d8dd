#!/bin/bash
# round 3, call BB: odd / tall heights on the group kernels (element-aligned packs, Stacked row slabs, Planar 8 / 16-wave tiles)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bb
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_env_switches.py -m gpu -q --tb=line --maxfail=25 -k "planar or radial or batchnorm or coupling or rqs or permute or stacked or chain or env" 2>&1 | tail -40 | tee gpurun_out/r3bb/pytest.txt
BJX_BENCH_DIMS=101,201,252,500,1000 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" | tee gpurun_out/r3bb/tall_flows.md
BJX_BENCH_DIMS=63,333,1001 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | tee gpurun_out/r3bb/odd_flows.md
exit 0
