#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bzteam; mkdir -p $O; cd $R
run() { timeout 600 python bench.py --workload bzip2_900k --steps 4 --warmup 1 --no-cpu-baseline --no-per-codec $@ > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
    print("$*:", round(d["ms_per_step"], 1), "ms", round(d["value"], 2), "GiB/s")
except Exception as e:
    print("$*: failed", e, open("$O/b.err").read()[-300:])
PY
}
run --tuning bzip2_pipeline=1
run --tuning bzip2_pipeline=2
run --tuning bzip2_pipeline=4
run --tuning bzip2_pipeline=8
run --tuning bzip2_pipeline=4 --tuning bzip2_team_threads=512
run --tuning bzip2_pipeline=4 --tuning bzip2_team_threads=512 --tuning bzip2_team_per_cu=2
run --tuning bzip2_pipeline=8 --tuning bzip2_team_threads=512
run --tuning bzip2_pipeline=4 --tuning bzip2_team_threads=256 --tuning bzip2_team_per_cu=2
