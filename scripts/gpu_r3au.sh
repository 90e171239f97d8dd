#!/bin/bash
# round 3, call AU: clamped rows: exact rounds for the transform's inverse, scan + fallback in its pullback
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3au; O=gpurun_out/r3au
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_env_switches.py -m gpu -q -p no:cacheprovider -k "simplex or sweep or shape or TALL or covers" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
BJX_BENCH_KS=200,500,1000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse"
BJX_PROBE_DTYPE=f64 BJX_BENCH_LOG2N=19 BJX_BENCH_KS=200,500 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse"
exit 0
