#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_${1:-p1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/exp_small.py 2048 64 || exit 1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --output-format csv -d $O/p$i -o p$i -- python $R/tools/exp_small.py 2048 64 > $O/p$i.log 2>&1 || echo "pass $i failed"
done <<'L'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA
GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
L
for k in swc_inflate swc_lz_resolve; do echo "== $k"; python $R/tools/pmc_report.py $O $k; done > $O/summary.txt 2>&1
cat $O/summary.txt
