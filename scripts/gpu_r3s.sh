#!/bin/bash
# round 3, call S: is the forward tall kernel slower when the pullback rows run in the same process?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3s; O=gpurun_out/r3s
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "simplex_vjp_long" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_KS=200,500
echo "--- fwd only"; BJX_PROBE_ROWS=fwd python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
echo "--- all rows"; python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
echo "--- fwd only again"; BJX_PROBE_ROWS=fwd python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
exit 0
