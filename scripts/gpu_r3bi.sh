#!/bin/bash
# round 3, call BI: C4 against the library of the commit before the odd-height work (element-aligned accesses now behind a template flag); spline slab sizes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3bi
( bash scripts/gpu_ab_libs.sh c4 3 - scripts/ab_libbjx_prev.so ) 2>&1 | tee gpurun_out/r3bi/ab_prev.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line --maxfail=25 -k "planar or rqs" 2>&1 | tail -3 | tee gpurun_out/r3bi/pytest.txt
for sl in 128 160 192; do echo "BJX_RQS_SLAB=$sl"; BJX_RQS_SLAB=$sl BJX_BENCH_DIMS=101,200,500,1000 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "RQS"; done | tee gpurun_out/r3bi/rqs_slab.md
BJX_LIB_PATH=$PWD/scripts/ab_libbjx_prev.so BJX_BENCH_DIMS=128 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Planar" | tee gpurun_out/r3bi/planar128_prev.md
BJX_BENCH_DIMS=128 timeout 600 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "Planar" | tee gpurun_out/r3bi/planar128_new.md
exit 0
