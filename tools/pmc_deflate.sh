#!/bin/bash
# PMC passes over the Deflate kernel at full occupancy (65,536 streams).  Counters only (no tracing domains).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_${1:-deflate}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/exp_small.py 2048 32 || exit 1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --output-format csv -d $O/p$i -o p$i -- python $R/tools/exp_small.py 2048 32 > $O/p$i.log 2>&1 || echo "pass $i failed"
done <<'L'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TOTAL_WAVEFRONTS_sum
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
L
python $R/tools/pmc_report.py $O swc_ > $O/summary.txt 2>&1
cat $O/summary.txt
