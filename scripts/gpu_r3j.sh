#!/bin/bash
# round 3, call J: chunked walkers with all chunk loads in flight (raw buffer loads), f64 find_alpha with a Float32 pre-solve
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3j; O=gpurun_out/r3j
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
echo "== tall columns: NEW (chunked walkers)"; BJX_BENCH_LOG2N=20 BJX_BENCH_KS=80,100,200,256,500,1000 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^|" | tee $O/tall_new.md
echo "== tall columns: OLD (whole-column tiles)"; BJX_SEQ_CHUNK_MIN=100000000 BJX_SIMPLEX_VJP_CHUNK_MIN=100000000 BJX_BENCH_LOG2N=20 BJX_BENCH_KS=100,200,500 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^|" | tee $O/tall_old.md
echo "== f64 rows"; python scripts/bench_f64.py 2>/dev/null | grep "^|" | tee $O/f64_rows.md
exit 0
