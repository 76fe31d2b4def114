#!/bin/bash
# Kernel trace + PMC traffic of the headline command only (the tail of tools/gpu_round.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r01}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-per-codec > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
python $R/tools/rocpd_summary.py $O/prof_bench/bench_results.db | tee $O/kernel_stats.txt
tail -1 $O/prof_bench.log | cut -c1-200
bash $R/tools/pmc_bench.sh deflate64k $TAG > $O/pmc.log 2>&1; tail -12 $O/pmc.log
