#!/bin/bash
# Same-box A/B of one bench workload (default: BASELINE configs[2], the spline) between two builds of the library (BJX_LIB_PATH):
# ab/libbjx_base.so (kept by hand before a kernel change; git-ignored, shipped by gpurun) and the tree's, both table policies.
#   scripts/ab_c3.sh [workload] [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-c3}; N=${2:-2}
one() { python bench.py --workload $WL --no-rows --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'kernel', round(d['roofline']['kernel_ms']*1e3,1), 'frac', round(d['roofline']['frac'],3))"; }
for i in $(seq $N); do
[ -f $R/ab/libbjx_base.so ] && BJX_LIB_PATH=$R/ab/libbjx_base.so one "base lib, tables kept    "
[ -f $R/ab/libbjx_base.so ] && BJX_LIB_PATH=$R/ab/libbjx_base.so one "base lib, helper per call" --no-cache-params
one "tree lib, tables kept    "
one "tree lib, helper per call" --no-cache-params
done
