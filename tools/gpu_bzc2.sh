#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bzc; mkdir -p $O; cd $R
SWC_BZ2C_TRACE=1 timeout 300 python tools/exp_bzip2_compress.py 32 9 2>&1 | tail -60 | tee $O/timing_trace.log
