#!/bin/bash
# round 3, call BH: same-box A/B of C4 / C5a against the library of the commit before the odd-height work; multi-dword tail stores; spline slab sizes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bh
( bash scripts/gpu_ab_libs.sh c4 2 - scripts/ab_libbjx_prev.so; bash scripts/gpu_ab_libs.sh c5a 2 - scripts/ab_libbjx_prev.so; bash scripts/gpu_ab_libs.sh c3 2 - scripts/ab_libbjx_prev.so ) 2>&1 | tee gpurun_out/r3bh/ab_prev.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line --maxfail=25 -k "planar or radial or batchnorm or coupling or stacked" 2>&1 | tail -5 | tee gpurun_out/r3bh/pytest.txt
BJX_BENCH_DIMS=101,201 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "RQS\|Permute" | tee gpurun_out/r3bh/odd.md
for sl in 64 96 128; do echo "BJX_RQS_SLAB=$sl"; BJX_RQS_SLAB=$sl BJX_BENCH_DIMS=200,500,1000 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "RQS"; done | tee gpurun_out/r3bh/rqs_slab.md
exit 0
