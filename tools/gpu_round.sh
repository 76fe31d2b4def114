#!/bin/bash
# Per-round GPU pass: parity tests, bench lines for every workload, rocprofv3 kernel trace, PMC traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r01}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | cut -c1-400
for W in lz4_4m bzip2_900k lzma2_256k; do
  timeout 900 python bench.py --workload $W --steps 3 --warmup 1 > $O/bench_$W.log 2>&1; echo "bench $W rc=$?"; tail -1 $O/bench_$W.log | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
python $R/tools/rocpd_summary.py $O/prof_bench/bench_results.db | tee $O/kernel_stats.txt
bash $R/tools/pmc_bench.sh deflate64k $TAG > $O/pmc.log 2>&1; tail -12 $O/pmc.log
