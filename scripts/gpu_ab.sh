#!/bin/bash
# usage: gpu_ab.sh "<bench_rows --only filter>" [rounds]   — A/B of tmp_ab/old.so vs tmp_ab/new.so on ONE box
# (runs on the box's scratch copy: an older library that lacks newer entry points is bound without them there)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
sed -i 's/^    if missing:  # an incomplete ABI.*/    if False:/; s/^    for name, (res, args) in SIGNATURES.items():/    for name, (res, args) in [kv for kv in SIGNATURES.items() if hasattr(lib, kv[0])]:/' bijectors.jl_amd/_lib.py
for r in $(seq 1 ${2:-2}); do
  for v in old new; do
    cp tmp_ab/$v.so bijectors.jl_amd/libbjx_hip.so
    echo "== $v (round $r)"
    python scripts/bench_rows.py --only "$1" --steps 10 2>/dev/null | grep "^| " | grep -v "^| row\|^|---" | cut -d'|' -f2,4,8
  done
done
cp tmp_ab/new.so bijectors.jl_amd/libbjx_hip.so
exit 0
