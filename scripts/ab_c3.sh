#!/bin/bash
# Same-box A/B of BASELINE configs[2] (the spline) between two builds of the library (BJX_LIB_PATH) and the two table policies.
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python bench.py --workload c3 --no-rows --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'kernel', round(d['roofline']['kernel_ms']*1e3,1))"; }
for i in 1 2; do
[ -f $R/ab_libs/libbjx_hip_r04.so ] && BJX_LIB_PATH=$R/ab_libs/libbjx_hip_r04.so one "r04 lib, tables kept     "
[ -f $R/ab_libs/libbjx_hip_r04.so ] && BJX_LIB_PATH=$R/ab_libs/libbjx_hip_r04.so one "r04 lib, helper per call " --no-cache-params
one "tree lib, tables kept    "
one "tree lib, helper per call" --no-cache-params
done
