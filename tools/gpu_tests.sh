#!/bin/bash
# the GPU test tier (or the files given) on the box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tests; mkdir -p $O; cd $R
timeout 1500 python -m pytest ${@:-tests} -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
