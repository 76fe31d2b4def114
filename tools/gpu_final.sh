#!/bin/bash
# End-of-round evidence: the GPU test tier, the driver's bench command, kernel trace + PMC traffic + SQ counters of the
# headline command.  Usage: gpu_final.sh <tag>   (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
bash tools/gpu_bench_full.sh $TAG
bash tools/gpu_prof_headline.sh $TAG 2>&1 | tail -25
bash tools/pmc_sq.sh deflate64k $TAG > /dev/null 2>&1; cp gpurun_out/pmc_sq_${TAG}_deflate64k/summary.txt $O/sq_counters_deflate64k.txt; head -30 $O/sq_counters_deflate64k.txt
