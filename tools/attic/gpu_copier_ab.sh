#!/bin/bash
# phase 2 of Deflate / LZ4 on the box: the GPU tests of the two codecs, the A/B of the kernels (MODES: lz_copier values, 0 = the
# workgroup resolver, 1 / 2 = the wave copier with an 8 / 16 KiB window), with SQ=1 the shader counters of the headline launch.
# Usage: gpu_copier_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_lz4.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/exp_copier.py deflate64k 1.0 ${MODES:-0,1,2} 2>&1 | tee $O/ab_deflate64k.txt | grep lz_copier
timeout 600 python tools/exp_copier.py lz4_4m 1.0 ${MODES:-0,1,2} 2>&1 | tee $O/ab_lz4_4m.txt | grep lz_copier
if [ -n "$SQ" ]; then bash tools/pmc_sq.sh deflate64k ${1:-r05b} swc_lz > $O/sq.txt 2>&1; cat $O/sq.txt | head -40; fi
