#!/bin/bash
# round 3, call AM: Ordered / Simplex pullbacks on short columns in registers: parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3am; O=gpurun_out/r3am
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "ordered or simplex or sweep or shape or vjp" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_DIMS=3,4,8
echo "--- new"; python scripts/bench_small_vjp.py 2>&1 | grep "^|" | grep "Simplex\|Ordered" | tee $O/new.md
echo "--- BJX_SEQ_TINY=0"; BJX_SEQ_TINY=0 python scripts/bench_small_vjp.py 2>&1 | grep "^|" | grep "Simplex\|Ordered" | tee $O/old.md
exit 0
