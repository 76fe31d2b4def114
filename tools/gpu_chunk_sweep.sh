#!/bin/bash
# Occupancy / sub-chunk size sweep of the phase-1 kernel: rebuilds the library on the box per variant (the shipped build is untouched there).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-sweep}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for V in "$@"; do
  [ "$V" = "$TAG" ] && continue
  SWC_EXTRA_HIPCC_FLAGS="$V" python -m swcompression_amd.build --force > $O/build.log 2>&1 || { echo "build failed: $V"; tail -5 $O/build.log; continue; }
  timeout 300 python bench.py --no-cpu-baseline --no-per-codec --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), d['roofline'].get('per_kernel_ms'))"
done
