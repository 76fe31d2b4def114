#!/bin/bash
# Cycle-counter profile of the Deflate kernels: rebuilds the library with -DSWC_PROFILE on the box (the shipped build is untouched there).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-prof}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
SWC_EXTRA_HIPCC_FLAGS="-DSWC_PROFILE" python -m swcompression_amd.build --force > $O/build.log 2>&1; echo "build rc=$?"
python tools/exp_profile.py 2>&1 | tee $O/profile.txt; python tools/exp_profile_lz4.py 2>&1 | tee $O/profile_lz4.txt
