#!/bin/bash
# BZip2.compress on the GPU box: its GPU tests, timing, a kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bzc; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_bzip2_compress.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
timeout 300 python tools/exp_bzip2_compress.py 32 9 2>&1 | tee $O/timing.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bzc -- python $R/tools/exp_bzip2_compress.py 32 9 > $O/trace.log 2>&1 ); echo "trace rc=$?"
python tools/rocpd_summary.py $O/trace/bzc_results.db > $O/kernel_stats.txt 2>&1; head -40 $O/kernel_stats.txt
