#!/bin/bash
# Checksum kernels: parity tests + throughput on the headline batch.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_checksums.py -x -q 2>&1 | tail -5
timeout 600 python tools/exp_sums.py 2>&1 | tail -12
