#!/bin/bash
# Shader-side counters (issue / wait / instruction mix / LDS) of the kernels of one bench launch.  Counters only.
# Usage: pmc_sq.sh <workload> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-deflate64k}; TAG=${2:-r02}
O=$R/gpurun_out/pmc_sq_${TAG}_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $line --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-per-codec > $O/p$i.log 2>&1 || echo "pass $i failed"
done <<'L'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU
L
for k in ${3:-swc_}; do python $R/tools/pmc_report_all.py $O $k; done > $O/summary.txt 2>&1
cat $O/summary.txt
