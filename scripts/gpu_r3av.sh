#!/bin/bash
# round 3, call AV: inverse-map pullback: scan carry with a clamped check trip, rounds where a clamp binds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3av; O=gpurun_out/r3av
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_env_switches.py -m gpu -q -p no:cacheprovider -k "simplex or sweep or shape or TALL or covers" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR|Mismatched|Max abs" $O/pytest.txt | head
export BJX_BENCH_KS=200,500,1000
echo "--- scan (default)"; python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "vjp(inverse"
echo "--- BJX_SIMPLEX_VJP_TALL_SCAN=0"; BJX_SIMPLEX_VJP_TALL_SCAN=0 BJX_BENCH_KS=200,500 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "vjp(inverse"
exit 0
