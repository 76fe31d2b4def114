#!/bin/bash
# LZ4 parse sub-chunk size sweep (library rebuilt on the box per variant).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-lz4sweep}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for V in "$@"; do
  [ "$V" = "$TAG" ] && continue
  SWC_EXTRA_HIPCC_FLAGS="$V" python -m swcompression_amd.build --force > $O/build.log 2>&1 || { echo "build failed: $V"; tail -5 $O/build.log; continue; }
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --workload lz4_4m --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1
  echo "== $V"; python $R/tools/rocpd_summary.py $O/prof/bench_results.db | grep "parse\|resolve" | cut -c1-110
  rm -rf $O/prof; cd $R
done
