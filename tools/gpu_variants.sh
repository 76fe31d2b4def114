#!/bin/bash
# A/B of prebuilt library variants (tools/build_variant.sh): the headline bench line per variant.
# Usage: gpu_variants.sh <tag> <name> [<name> ...]     (name "base" = the shipped library)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for N in "$@"; do
  L=$R/swcompression_amd/variants/libswc_$N.so; [ "$N" = base ] && L=
  SWC_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-per-codec --steps ${STEPS:-10} --warmup 2 ${BENCH_ARGS} 2>$O/err_$N.log | tail -1 > $O/bench_$N.json
  python -c "
import sys, json
d = json.loads(open('$O/bench_$N.json').read()); print('$N', round(d['ms_per_step'],2), {k: round(v,2) for k,v in (d['roofline'].get('per_kernel_ms') or {}).items()})" 2>&1 | tail -1
done
