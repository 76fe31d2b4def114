#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1200 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --cpu-seconds 2 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GiB/s=%.1f ms=%.2f'%(d['value'], d['ms_per_step']), d['roofline'].get('per_kernel_ms'), d.get('config1_latency'))"
