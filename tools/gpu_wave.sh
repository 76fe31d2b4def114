#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -4
timeout 600 python tools/exp_wave.py 2>&1 | grep "wave\|n=  4096 lane"
