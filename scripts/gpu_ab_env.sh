#!/bin/bash
# usage: gpu_ab_env.sh "<bench_rows --only filter>" "ENV=V ..." ["ENV=V ..."]...  — old/new libs under each env setting, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
F=$1; shift
for e in "$@"; do
  for v in old new; do
    cp tmp_ab/$v.so bijectors.jl_amd/libbjx_hip.so
    echo "== $v  [$e]"
    env $e python scripts/bench_rows.py --only "$F" --steps 10 2>/dev/null | grep "^| " | grep -v "^| row\|^|---" | cut -d'|' -f2,4,8
  done
done
cp tmp_ab/old.so bijectors.jl_amd/libbjx_hip.so
exit 0
