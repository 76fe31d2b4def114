#!/bin/bash
# HBM traffic of one bench launch: rocprofv3 --pmc passes (counters only, no tracing domains) over
#   python bench.py --workload W --steps 1 --warmup 1 --no-cpu-baseline --no-per-codec
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots) -> separate passes.  Usage: pmc_bench.sh <workload> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-deflate64k}; TAG=${2:-r01}
O=$R/gpurun_out/pmc_bench_${TAG}_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-per-codec > $O/plain.log 2>&1 || { tail -5 $O/plain.log; exit 1; }
i=0
for ctr in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-per-codec > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python $R/tools/pmc_traffic.py $O $W $TAG | tee $O/traffic.json
