#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_deflate.py -x -q 2>&1 | tail -15
timeout 900 python bench.py --workload bzip2_900k --steps 2 --warmup 1 --cpu-seconds 3 > $O/bench_bz.log 2>&1; tail -1 $O/bench_bz.log | cut -c1-250
