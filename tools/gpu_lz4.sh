#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -15
timeout 900 python bench.py --workload lz4_4m --steps 3 --warmup 1 --cpu-seconds 3 > $O/bench_lz4.log 2>&1; tail -1 $O/bench_lz4.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lz4 -o lz4 -- python $R/bench.py --workload lz4_4m --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_lz4.log 2>&1
python $R/tools/rocpd_summary.py $O/prof_lz4/lz4_results.db
