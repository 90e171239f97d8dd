#!/bin/bash
# Same-box A/B of rows of scripts/bench_rows.py between the tree's library and every ab/*.so (BJX_LIB_PATH; ab/ is git-ignored and
# shipped by gpurun):   scripts/ab_rows.sh "<row substring>[,<row substring>...]" [log2 batch]
R=${GRAFT_REPO_ROOT:-/root/repo}
ONLY=${1:?row filter}; LB=${2:-22}
run() { echo "== $1"; python $R/scripts/bench_rows.py --only "$ONLY" --log2-batch $LB 2>/dev/null | grep -v "^|---\|^| row" | awk -F'|' '{printf "%-70s kernel %s ms  %s %%\n", $2, $4, $8}'; }
run "tree"
for f in $R/ab/*.so; do [ -f "$f" ] && BJX_LIB_PATH=$f run "$(basename $f)"; done
