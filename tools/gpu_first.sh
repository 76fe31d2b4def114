#!/bin/bash
# First on-GPU pass: parity tests, smoke, bench line, rocprofv3 kernel trace of the bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > $O/rocminfo.txt 2>&1
nproc > $O/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
ls -R $O/prof_bench | head -20
