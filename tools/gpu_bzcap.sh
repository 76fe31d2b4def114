#!/bin/bash
# BZip2 walk kernel vs segment-buffer capacity factor (rebuilds the library on the box for each value)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for F in 4 2; do
  SWC_EXTRA_HIPCC_FLAGS="-DSWC_BZ_SEGCAP=$F" python -c "from swcompression_amd import build; build.build(force=True)" > $O/build_$F.log 2>&1 || { tail -5 $O/build_$F.log; continue; }
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bz$F -o bench -- python $R/bench.py --workload bzip2_900k --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_bz$F.log 2>&1)
  echo "cap factor $F:"; python tools/rocpd_summary.py $O/prof_bz$F/bench_results.db | grep "walk\|stage" | cut -c1-140
done
