#!/bin/bash
# Where the instructions of the record-granular copier go: variants of the library whose back() stops early (-DSWC_LZC_CUT=1..3,
# tools/build_variant.sh), each under rocprofv3 --pmc.  Usage: exp_copier_counts.sh <tag> <variants...>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-cuts}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    NOVERIFY=1 SWC_LIB=$R/swcompression_amd/variants/libswc_$V.so timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/$V/p$i -o p -- python $R/tools/exp_copier.py ${WL:-deflate64k} ${SCALE:-0.25} 1 > $O/$V.p$i.log 2>&1 || echo "$V pass failed"
  done
  grep lz_copier $O/$V.p*.log | tail -1
  python $R/tools/pmc_report_all.py $O/$V swc_lz_copy > $O/$V.txt 2>&1; cat $O/$V.txt
done
