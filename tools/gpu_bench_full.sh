#!/bin/bash
# The driver's command: `python bench.py` with no flags (headline + per_codec), timed.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-benchfull}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
S=$(date +%s); python bench.py > $O/bench_full.log 2> $O/bench_full.err; echo "rc=$? wall=$(( $(date +%s) - S ))s"; tail -c 800 $O/bench_full.err
python - <<PY
import json
d = json.loads(open("$O/bench_full.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"], d["verify"], d.get("cpu_baseline", {}).get("value"), d.get("config1_latency", {}).get("median_ms"))
for k, v in d.get("per_codec", {}).items():
    print(k, round(v["value"], 1), round(v["ms_per_step"], 1), round(v["roofline"]["frac"], 4), v["verify"]["units_verified"], round(v.get("cpu_baseline", {}).get("value", 0), 3))
PY
