#!/bin/bash
# round 3, call AT: speculative unclamped rounds in the tall Simplex inverse: parity (incl. clamped rows), A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3at; O=gpurun_out/r3at
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "simplex or sweep or shape" ) > $O/pytest.txt 2>&1
echo "pytest: $(grep -E 'passed|failed|error' $O/pytest.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest.txt | head
export BJX_PROBE_ROWS=fwd BJX_BENCH_KS=160,200,300,500
echo "--- spec"; python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse"
echo "--- BJX_SEQ_TALL_INV_SPEC=0"; BJX_SEQ_TALL_INV_SPEC=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse"
echo "--- spec, K = 1000, 2000 with INV_MAX=2048"; BJX_SEQ_TALL_INV_MAX=2048 BJX_BENCH_KS=1000,2000 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep "inverse"
exit 0
