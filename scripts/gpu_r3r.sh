#!/bin/bash
# round 3, call R: Simplex pullbacks on tall columns (bjx_tall.hip): parity, then A/B against the chunked two-pass kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3r; O=gpurun_out/r3r
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "ordered or simplex or sweep or shape or vjp" ) > $O/pytest_seq.txt 2>&1
echo "pytest (ordered/simplex/vjp): $(grep -E 'passed|failed|error' $O/pytest_seq.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_seq.txt | head -30
export BJX_BENCH_KS=${KS:-160,200,256,500,1000}
echo "## bjx_tall.hip (G lanes per column)" > $O/tall_ab.md
python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall_ab.md
echo "## BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0 (walkers)" >> $O/tall_ab.md
BJX_SEQ_TALL=0 BJX_SIMPLEX_VJP_TALL=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | grep vjp | tee -a $O/tall_ab.md
exit 0
