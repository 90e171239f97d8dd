#!/bin/bash
# round 3, call I: chunked walkers (forward + Simplex vjp) and RQS row slabs — full suite, then same-box A/B against the round-2 paths
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3i; O=gpurun_out/r3i
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
echo "pytest -m gpu: $(grep -E 'passed|failed|error' $O/pytest_gpu.txt | tail -1)"; grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20; grep real $O/pytest_gpu.txt
echo "== tall columns: NEW (chunked walkers)"; BJX_BENCH_LOG2N=20 BJX_BENCH_KS=64,80,100,128,200,256,500,1000 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^|" | tee $O/tall_new.md
echo "== tall columns: OLD (whole-column tiles)"; BJX_SEQ_CHUNK_MIN=100000000 BJX_SIMPLEX_VJP_CHUNK_MIN=100000000 BJX_BENCH_LOG2N=20 BJX_BENCH_KS=80,100,200,256,500 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^|" | tee $O/tall_old.md
echo "== chunk threshold sweep (Simplex fwd K=64/80/100: tile 16.6/20.7/25.9 KB)"; for cm in 10000 18000 22000 30000; do echo "chunk_min=$cm"; BJX_SEQ_CHUNK_MIN=$cm BJX_BENCH_LOG2N=20 BJX_BENCH_KS=63,80,100 python scripts/probe_tall_simplex.py 2>/dev/null | grep "^| SimplexBijector\|^| OrderedBijector\|^| inverse(Simplex"; done | tee $O/chunk_sweep.md
echo "== small dims NEW"; python scripts/bench_small_dims.py 2>/dev/null | grep "RQS" | tee $O/rqs_dims_new.md
echo "== small dims OLD (no slabs)"; BJX_RQS_SLAB=0 python scripts/bench_small_dims.py 2>/dev/null | grep "RQS" | tee $O/rqs_dims_old.md
exit 0
