#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q > $O/pytest_deflate.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_deflate.log
timeout 300 python tools/exp_deflate.py > $O/exp2.log 2>&1; tail -20 $O/exp2.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_exp -o exp -- python $R/tools/exp_small.py 2048 32 > $O/prof_exp.log 2>&1
python $R/tools/rocpd_summary.py $O/prof_exp/exp_results.db
