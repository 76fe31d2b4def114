#!/bin/bash
# Kernel trace of the non-headline workloads (per-stage times of the BZip2 and LZMA kernels).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r01}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in ${2:-bzip2_900k lzma2_256k}; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o bench -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_$W.log 2>&1; echo "rocprof $W rc=$?"
  python $R/tools/rocpd_summary.py $O/prof_$W/bench_results.db | tee $O/kernel_stats_$W.txt
done
