#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -3
timeout 300 python tools/exp_deflate.py > $O/exp3.log 2>&1; tail -5 $O/exp3.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_q.log 2>&1; tail -1 $O/bench_q.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GiB/s=%.1f ms=%.2f'%(d['value'], d['ms_per_step']), d['roofline'].get('per_kernel_ms'))"
