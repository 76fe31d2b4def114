#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_bzip2.py tests/test_gpu_many.py -x -q -k "bzip2 or zip" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bz -o bench -- python $R/bench.py --workload bzip2_900k --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_bzip2.log 2>&1
tail -1 $O/bench_bzip2.log | cut -c1-160
python $R/tools/rocpd_summary.py $O/prof_bz/bench_results.db | cut -c1-150 | head -8
