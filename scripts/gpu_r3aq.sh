#!/bin/bash
# round 3, call AQ: the driver's sequence: smoke(), default bench line, two-rank bench on one device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3aq; O=gpurun_out/r3aq
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -4
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype')}); print(d['roofline']); print(d['cpu_baseline'])"
exit 0
