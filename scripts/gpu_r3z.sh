#!/bin/bash
# round 3, call Z: groups per block in the mixed walker (short columns): parity of the affected tests, A/B at dim = 2, 3, 4, 5, 10
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3z; O=gpurun_out/r3z
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "chain or stacked or sweep or shape or coupling or summation or order" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=2,3,4,5,10
echo "--- new"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | tee $O/new.md
echo "--- BJX_MIXED_GPB=1"; BJX_MIXED_GPB=1 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | grep -i "exp\|Stacked\|Coupling" | tee $O/old.md
exit 0
