#!/bin/bash
# Bench lines of the LZ4 / BZip2 / LZMA2 workloads (HBM-resident, stated sizes).  Usage: gpu_other.sh <tag> [workloads]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-other}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for W in ${2:-lz4_4m bzip2_900k lzma2_256k}; do
  timeout 900 python bench.py --workload $W --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_$W.log 2>&1; echo "$W rc=$?"
  tail -1 $O/bench_$W.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'],2), 'GiB/s', round(d['ms_per_step'],2), 'ms', round(d['roofline']['frac']*100,3), '% of HBM peak')"
done
