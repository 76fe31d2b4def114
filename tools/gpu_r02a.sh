#!/bin/bash
# Round 2, first GPU pass of the rewritten Deflate path: parity tests, bench (new vs the previous phase 1), LZ4 through the new resolve, kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02a}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q > $O/pytest_deflate.log 2>&1; echo "pytest deflate rc=$?"; tail -5 $O/pytest_deflate.log
timeout 600 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_many.py -x -q > $O/pytest_lz4.log 2>&1; echo "pytest lz4/many rc=$?"; tail -5 $O/pytest_lz4.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | cut -c1-1200

timeout 900 python bench.py --workload lz4_4m --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_lz4_4m.log 2>&1; echo "bench lz4 rc=$?"; tail -1 $O/bench_lz4_4m.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
python $R/tools/rocpd_summary.py $O/prof_bench/bench_results.db | tee $O/kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lz4 -o bench -- python $R/bench.py --workload lz4_4m --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_lz4.log 2>&1; echo "rocprof lz4 rc=$?"
python $R/tools/rocpd_summary.py $O/prof_lz4/bench_results.db | tee $O/kernel_stats_lz4.txt
