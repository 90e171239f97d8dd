#!/bin/bash
# round 3, call AC: lane-per-column chains on short columns (chain_tiny_kernel): parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ac; O=gpurun_out/r3ac
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "chain or stacked or sweep or shape or coupling or elementwise or logit or truncated or leaky" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=3,9,10,11,13,14,15,17
echo "--- new"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘\|Stacked" | tee $O/new.md
echo "--- BJX_CHAIN_TINY=0"; BJX_CHAIN_TINY=0 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘" | tee $O/old.md
exit 0
