#!/bin/bash
# round 3, call AO: 2x2 ... 4x4 matrix bijectors without the tile: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ao; O=gpurun_out/r3ao
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "matrix or corr or pd or shape or sweep" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
echo "--- new"; python scripts/bench_matrix_small.py 2>&1 | grep "^|" | grep "| 2 |\|| 3 |\|| 4 |" | tee $O/new.md
echo "--- BJX_MATRIX_LANE_DIRECT=0"; BJX_MATRIX_LANE_DIRECT=0 python scripts/bench_matrix_small.py 2>&1 | grep "^|" | grep "| 2 |\|| 3 |\|| 4 |" | tee $O/old.md
exit 0
