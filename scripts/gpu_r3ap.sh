#!/bin/bash
# round 3, call AP: dims 4 and 8 (whole packs) through the lane-per-column Stacked kernel: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ap; O=gpurun_out/r3ap
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "stacked or chain or sweep or shape or named" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=2,4,8
echo "--- new"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘\|Stacked" | tee $O/new.md
echo "--- BJX_STACKED_TINY=0"; BJX_STACKED_TINY=0 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘\|Stacked" | tee $O/old.md
exit 0
