#!/bin/bash
# Instruction-issue microbenchmark (tools/micro/issue_bench.hip) + a quick headline line.  Usage: gpu_micro.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-micro}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 300 tools/micro/issue_bench 2000 > $O/issue_bench.txt 2>&1; echo "issue_bench rc=$?"; cat $O/issue_bench.txt
timeout 600 python bench.py --no-cpu-baseline --no-per-codec --steps 5 > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kernel_ms'))"
