#!/bin/bash
# Instruction-cache counters of the kernels of one bench launch (counters only).  Usage: pmc_icache.sh <workload> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-deflate64k}; TAG=${2:-r03}
O=$R/gpurun_out/pmc_icache_${TAG}_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/p1 -o p1 -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-per-codec > $O/p1.log 2>&1 || echo "pass failed"
python $R/tools/pmc_report_all.py $O swc_ > $O/summary.txt 2>&1
cat $O/summary.txt
