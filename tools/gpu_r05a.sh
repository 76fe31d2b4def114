#!/bin/bash
# round 5, first measurement of the record-granular copier (lz_copy.h): GPU tests of the two codecs that use it, then the A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_lz4.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/exp_copier.py deflate64k 1.0 0,1,2 2>&1 | tee $O/ab_deflate64k.txt | tail -5
timeout 600 python tools/exp_copier.py lz4_4m 1.0 0,1,2 2>&1 | tee $O/ab_lz4_4m.txt | tail -5
