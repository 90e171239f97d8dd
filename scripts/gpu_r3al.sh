#!/bin/bash
# round 3, call AL: Ordered / Simplex on short columns in registers (bjx_tiny.hip): parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3al; O=gpurun_out/r3al
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "ordered or simplex or sweep or shape or nan or inf or edge" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
echo "--- new"; python scripts/bench_small_sizes.py 2>&1 | grep "^|" | grep "Simplex\|Ordered" | tee $O/new.md
echo "--- BJX_SEQ_TINY=0"; BJX_SEQ_TINY=0 python scripts/bench_small_sizes.py 2>&1 | grep "^|" | grep "Simplex\|Ordered" | tee $O/old.md
exit 0
