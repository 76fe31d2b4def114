#!/bin/bash
# End-of-round evidence from the shipped build: the GPU test tier, smoke(), the driver's bench command, and for EVERY
# workload (deflate64k lz4_4m bzip2_900k lzma2_256k) a rocprofv3 kernel trace, the PMC traffic passes and the SQ counter
# passes.  Usage: gpu_final.sh <tag> [parts...]   parts: tests bench trace pmc sq bzc (default: all but bzc)
# Outputs under gpurun_out/<tag>/; tools/collect_profiles.sh <tag> copies what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-final}; shift; PARTS=${@:-tests bench trace pmc sq}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
WL=${WL:-"deflate64k lz4_4m bzip2_900k lzma2_256k"}; XWL=${XWL-"deflate64k_mix lzma2_256k_bin lz4_compress_4m deflate_compress_64k"}
PWL=${PWL-"deflate64k_mix lz4_compress_4m deflate_compress_64k"}   # PMC traffic for these too (the encode workloads: VERDICT r5)
for P in $PARTS; do case $P in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log ;;
bench)
  # (stdout's last line is the compact headline object the driver parses; the whole object is written to gpurun_out/bench_full.json by bench.py itself)
  S=$(date +%s); python bench.py > $O/bench_line.json 2> $O/bench_full.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s" | tee $O/bench_full.wall
  tail -1 $O/bench_line.json | wc -c; cp gpurun_out/bench_full.json $O/bench_full.json
  python - <<PY
import json
d = json.load(open("$O/bench_full.json"))
print("headline", round(d["value"], 1), "GiB/s", round(d["ms_per_step"], 2), "ms", d["roofline"]["per_kernel_ms"])
for k, v in d.get("per_codec", {}).items():
    print(k, round(v["value"], 1), "GiB/s", round(v["ms_per_step"], 1), "ms", {a: round(b, 1) for a, b in (v["roofline"].get("per_kernel_ms") or {}).items()})
PY
  ;;
trace)
  for W in $WL $XWL; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_$W -o bench -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-per-codec > $O/trace_$W.log 2>&1 ); echo "trace $W rc=$?"
    python tools/rocpd_summary.py $O/trace_$W/bench_results.db > $O/kernel_stats_$W.txt 2>&1; head -8 $O/kernel_stats_$W.txt
  done ;;
pmc)
  for W in $WL $PWL; do bash tools/pmc_bench.sh $W $TAG > $O/pmc_$W.log 2>&1; cp gpurun_out/pmc_bench_${TAG}_$W/traffic.json $O/${W}_traffic.json 2>/dev/null; python - <<PY
import json
try:
    d = json.load(open("$O/${W}_traffic.json")); print("$W traffic GB", round(d["hbm_bytes_per_launch"] / 1e9, 1), "fetch", round(d["fetch_bytes_corrected"] / 1e9, 1), "write", round(d["write_bytes"] / 1e9, 1))
except Exception as e:
    print("$W traffic: failed", e)
PY
  done ;;
bzc)   # BZip2.compress (host to host): timing with the stage trace, and a kernel trace of the same command
  SWC_BZ2C_TRACE=1 timeout 300 python tools/attic/exp_bzip2_compress.py 32 9 > $O/bzip2_compress_timing.txt 2>&1; grep -v "^\[bz2c\]\|amdgpu" $O/bzip2_compress_timing.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_bzc -o bzc -- python $R/tools/attic/exp_bzip2_compress.py 32 9 > $O/trace_bzc.log 2>&1 ); echo "trace bzc rc=$?"
  python tools/rocpd_summary.py $O/trace_bzc/bzc_results.db > $O/kernel_stats_bzip2_compress.txt 2>&1; head -8 $O/kernel_stats_bzip2_compress.txt ;;
sq)
  for W in $WL; do bash tools/pmc_sq.sh $W $TAG > /dev/null 2>&1; cp gpurun_out/pmc_sq_${TAG}_$W/summary.txt $O/sq_counters_$W.txt 2>/dev/null; head -12 $O/sq_counters_$W.txt; done ;;
esac; done
