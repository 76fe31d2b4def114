#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bzteam; mkdir -p $O; cd $R
for N in op8; do
( cd /tmp && export TMPDIR=/tmp && SWC_LIB=$R/swcompression_amd/variants/libswc_$N.so timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace$N -o bench -- python $R/tools/exp_bzteam.py > $O/trace$N.log 2>&1 ); echo "variant $N rc=$?"
python tools/rocpd_summary.py $O/trace$N/bench_results.db 2>&1 | grep "team_finish" | cut -c1-140
done
