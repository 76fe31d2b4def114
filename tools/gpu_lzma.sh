#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_lzma.py -x -q 2>&1 | tail -5
timeout 900 python bench.py --workload lzma2_256k --steps 2 --warmup 1 --cpu-seconds 3 > $O/bench_lzma.log 2>&1; tail -1 $O/bench_lzma.log | cut -c1-200
