#!/bin/bash
# round 3, last call: the whole GPU suite, the driver's bench line, the rows tables after the odd / tall height work
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3final
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2>$O/bench_default.err; cut -c1-400 $O/bench_default.json
BJX_BENCH_DIMS=101,201,252,500,1000 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" > $O/tall_flows.md; wc -l $O/tall_flows.md
BJX_BENCH_DIMS=63,77,257,300,333,1001 timeout 900 python scripts/bench_small_dims.py 2>&1 | grep "^|" > $O/odd_flows.md; wc -l $O/odd_flows.md
timeout 600 python scripts/bench_rows.py 2>/dev/null | grep "^|" > $O/rows.md; wc -l $O/rows.md
timeout 600 python scripts/bench_small_dims.py 2>/dev/null | grep "^|" > $O/small_dims.md; wc -l $O/small_dims.md
exit 0
