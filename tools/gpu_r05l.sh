#!/bin/bash
# round 5: Deflate compression on the device: tests, the encode workload of bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_deflate_compress.py tests/test_lz4_compress.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --workload deflate_compress_64k --steps 5 --warmup 1 --no-per-codec > $O/bench_defc.json 2> $O/bench_defc.err; echo "rc=$?"; tail -3 $O/bench_defc.err
python - <<PY
import json
d = json.loads(open("$O/bench_defc.json").read().strip().splitlines()[-1])
print(round(d["value"], 2), "GiB/s", round(d["ms_per_step"], 2), "ms", d["verify"], d.get("cpu_baseline"), d.get("stats"))
PY
