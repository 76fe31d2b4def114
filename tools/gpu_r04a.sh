#!/bin/bash
# Round-4 tuning loop: Deflate parity tests (+ named extra tests) and the headline bench line with per-kernel times.
# Usage: gpu_r04a.sh <tag> ["extra pytest args"]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r04a}; EXTRA=$2; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py $EXTRA -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-per-codec --steps 10 > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kernel_ms'))"
