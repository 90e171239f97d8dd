#!/bin/bash
# round 2, second GPU call: tests of everything new, A/B of the MFMA planar kernel, new bench rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2b; O=gpurun_out/r2b
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt
grep -E "^FAILED" $O/pytest_gpu_full.txt | head -40
b() { python bench.py --no-cpu-baseline --no-rows --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f Msamp/s  %7.1f GB/s  frac %.3f  kernel_ms %.4f  region_ms %.4f  step_ms %.4f' % (d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['stream_region_ms_per_step'], d['ms_per_step']))"; }
for m in 0 1 2 3 4 5 6 0; do echo -n "c4 BJX_PLANAR_MFMA=$m : "; BJX_PLANAR_MFMA=$m b --workload c4; done
for wl in vcorr pdvec; do echo -n "$wl : "; b --workload $wl; done
exit 0
