#!/bin/bash
# round 3, call Q: G lanes per column for tall Ordered / Simplex (bjx_tall.hip): parity, then A/B against the walkers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q -p no:cacheprovider -k "ordered or simplex or sweep or shape" ) > $O/pytest_seq.txt 2>&1
echo "pytest (ordered/simplex): $(grep -E 'passed|failed|error' $O/pytest_seq.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_seq.txt | head -30
export BJX_PROBE_ROWS=fwd BJX_BENCH_KS=${KS:-100,200,256,300,500,1000,2000}
echo "## bjx_tall.hip (G lanes per column)" > $O/tall_ab.md
python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall_ab.md
echo "## BJX_SEQ_TALL=0 (walkers)" >> $O/tall_ab.md
BJX_SEQ_TALL=0 python scripts/probe_tall_simplex.py 2>&1 | grep "^|" | tee -a $O/tall_ab.md
exit 0
