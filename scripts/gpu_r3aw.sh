#!/bin/bash
# round 3, call AW: two column sets per trip in the tall Simplex kernels (packed rounds): parity, A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3aw; O=gpurun_out/r3aw
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "simplex or sweep or shape or flags" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_PROBE_ROWS=fwd BJX_BENCH_KS=160,200,300,500,1000
echo "--- pair"; python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "Simplex"
echo "--- BJX_SEQ_TALL_PAIR=0"; BJX_SEQ_TALL_PAIR=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "Simplex"
exit 0
