#!/bin/bash
# round 3, call X: Simplex inverse on tall columns with the affine-scan carry: parity, A/B against the rounds, Float64
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3x; O=gpurun_out/r3x
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_env_switches.py -m gpu -q -p no:cacheprovider -k "ordered or simplex or sweep or shape or vjp or TALL or covers" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_BENCH_KS=100,160,200,300,500,1000,2000
echo "--- f32 scan";  python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse" | tee $O/f32_scan.md
echo "--- f32 rounds (INV_MAX 512)";  BJX_SEQ_TALL_INV_SCAN=0 BJX_SIMPLEX_VJP_TALL_SCAN=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse" | tee $O/f32_rounds.md
export BJX_BENCH_KS=100,200,500 BJX_PROBE_DTYPE=f64 BJX_BENCH_LOG2N=19
echo "--- f64 scan";  python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse" | tee $O/f64_scan.md
echo "--- f64 walkers";  BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse" | tee $O/f64_old.md
exit 0
