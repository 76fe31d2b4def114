#!/bin/bash
# LZ4 tuning loop: parity tests + the lz4_4m bench line with per-kernel times (+ named library variants).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-lz4b}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for N in base "$@"; do
  L=$R/swcompression_amd/variants/libswc_$N.so; [ "$N" = base ] && L=
  SWC_LIB=$L timeout 600 python bench.py --workload lz4_4m --no-cpu-baseline --no-per-codec --steps 5 --warmup 1 2>$O/err_$N.log | tail -1 > $O/bench_$N.json
  python -c "
import json
d = json.loads(open('$O/bench_$N.json').read()); print('$N', round(d['ms_per_step'],2), {k: round(v,2) for k,v in (d['roofline'].get('per_kernel_ms') or {}).items()})" 2>&1 | tail -1
done
