#!/bin/bash
# round 3, call AR: dead rows' log-det terms in closed form: parity + tall table (forward maps)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3ar; O=gpurun_out/r3ar
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "simplex or sweep or shape or c5" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
BJX_PROBE_ROWS=fwd BJX_BENCH_KS=100,160,200,300,500,1000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
exit 0
