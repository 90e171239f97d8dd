#!/bin/bash
# round 3, call AZ: per-sample chains at odd column heights with element-aligned packs: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3az; O=gpurun_out/r3az
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "chain or sweep or shape or elementwise or logit or truncated" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=63,257,261,333,513,777,1001,2049
echo "--- new (MIN=17)"; python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘"
echo "--- BJX_CHAIN_UNALIGNED=0"; BJX_CHAIN_UNALIGNED=0 python scripts/bench_small_dims.py 2>&1 | grep "^|" | grep "exp∘"
exit 0
