#!/bin/bash
# round 3, call AG: Planar pullback on short columns without data tiles: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ag; O=gpurun_out/r3ag
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "planar or vjp or sweep or shape" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=2,3,4,8
echo "--- new"; python scripts/bench_small_vjp.py 2>&1 | grep "^|" | grep -v "^|--\|pullback" | tee $O/new.md
echo "--- BJX_PLANAR_WALK_DIRECT=0"; BJX_PLANAR_WALK_DIRECT=0 python scripts/bench_small_vjp.py 2>&1 | grep "^|" | grep "Planar" | tee $O/old.md
exit 0
