#!/bin/bash
# First GPU pass: smoke, parity tests, first bench lines, rocprof kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.txt
echo "== bench c2"; timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/bench_c2.json
echo "== bench c2 NT"; BJX_NT=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tee gpurun_out/bench_c2_nt.json
for w in c2v c3 c4 c5a c5b; do
  echo "== bench $w"; timeout 600 python bench.py --workload $w --steps 10 --warmup 3 2>&1 | tee gpurun_out/bench_$w.json
done
echo "== rocprof c2"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_c2" -o c2 -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof_c2.log" 2>&1 )
find gpurun_out/prof_c2 -name "*stats*" | head; 
f=$(find gpurun_out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
