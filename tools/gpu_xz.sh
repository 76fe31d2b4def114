#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_lzma.py tests/test_gpu_many.py -x -q 2>&1 | tail -15
