#!/bin/bash
# Round 2 quick pass: Deflate + LZ4 parity tests, bench lines (phase timing), no profiler.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r02b}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_lz4.py -x -q -k "not lane and not wave" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 5 > $O/bench_deflate64k.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_deflate64k.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('per_kernel_ms'))"
timeout 900 python bench.py --workload lz4_4m --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_lz4_4m.log 2>&1; echo "bench lz4 rc=$?"; tail -1 $O/bench_lz4_4m.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
if [ "$2" = "prof" ]; then bash tools/gpu_prof2.sh $TAG; fi
