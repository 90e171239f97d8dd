#!/bin/bash
# round 3, call AE: short columns without tiles (Radial, the column skeleton: BatchNorm / RQS / Permute / Coupling): parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ae; O=gpurun_out/r3ae
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=2,3,5,7
echo "--- new"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | tee $O/new.md
echo "--- BJX_COLDIRECT=0 BJX_PLANAR_WALK_DIRECT=0"; BJX_COLDIRECT=0 BJX_PLANAR_WALK_DIRECT=0 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | tee $O/old.md
exit 0
