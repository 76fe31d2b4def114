#!/bin/bash
# Quick pass: Deflate parity tests + the headline bench line with per-kernel times.  Usage: gpu_quick.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-quick}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-per-codec --steps 5 > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kernel_ms'))"
