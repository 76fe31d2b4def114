#!/bin/bash
# LZ4 variants x payload classes: ms per step and per kernel.  Usage: gpu_lz4c.sh <tag> <variant>...
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-lz4c}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for N in "$@"; do for P in text:256 mix:256; do
  L=$R/swcompression_amd/variants/libswc_$N.so; [ "$N" = base ] && L=
  SWC_LIB=$L timeout 600 python bench.py --workload lz4_4m --parts $P --no-cpu-baseline --no-per-codec --steps 4 --warmup 1 2>$O/err_$N.log | tail -1 > $O/bench_${N}_$P.json
  python -c "
import json
d = json.loads(open('$O/bench_${N}_$P.json').read()); print('$N $P', round(d['ms_per_step'],2), {k: round(v,2) for k,v in (d['roofline'].get('per_kernel_ms') or {}).items()})" 2>&1 | tail -1
done; done
