#!/bin/bash
# round 3, call AF: RQS on short columns through the lane-per-column functor kernel: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3af; O=gpurun_out/r3af
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "rqs or spline or sweep or shape or coupling" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=1,2,3,5,6,7
echo "--- new"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "RQS" | tee $O/new.md
echo "--- BJX_RQS_TINY=0"; BJX_RQS_TINY=0 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "RQS" | tee $O/old.md
exit 0
