#!/bin/bash
# round 3, call V: tall columns, Float32 after the trims and Float64 A/B (2^19 columns)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3v; O=gpurun_out/r3v
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "ordered or simplex" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
echo "--- f32"; BJX_PROBE_ROWS=fwd BJX_BENCH_KS=100,200,500,1000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector"
echo "--- f64 new"; BJX_PROBE_DTYPE=f64 BJX_BENCH_LOG2N=19 BJX_BENCH_KS=100,200,500,1000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | tee $O/f64_new.md
echo "--- f64 old"; BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0 BJX_PROBE_DTYPE=f64 BJX_BENCH_LOG2N=19 BJX_BENCH_KS=100,200,500,1000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep -v "^|--\|bijector" | tee $O/f64_old.md
exit 0
