#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; timeout 600 python tools/exp_bzteam_small.py 2>&1 | grep -v amdgpu
