#!/bin/bash
# Copies the summaries of a tools/gpu_final.sh run into profiles/ under the round's tag.  Usage: collect_profiles.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:?tag}; O=$R/gpurun_out/$TAG; P=$R/profiles
# (z files come from the SHIPPED build only: nothing uncommitted under csrc/, the library newer than every source)
if [[ "$TAG" == *z ]]; then
  if [ -n "$(cd $R && git status --porcelain swcompression_amd/csrc include)" ]; then echo "refusing: uncommitted changes under swcompression_amd/csrc or include/ -- commit first, then run gpu_final.sh again"; exit 1; fi
  for f in $R/swcompression_amd/csrc/* $R/include/*; do if [ "$f" -nt "$R/swcompression_amd/libswc_hip.so" ]; then echo "refusing: $f is newer than libswc_hip.so"; exit 1; fi; done
fi
cp $O/bench_full.json $P/${TAG}_bench_full.json 2>/dev/null
cp $O/bench_line.json $P/${TAG}_bench_line.json 2>/dev/null
cp $O/pytest_gpu.log $P/${TAG}_pytest_gpu.log 2>/dev/null
for W in deflate64k lz4_4m bzip2_900k lzma2_256k deflate64k_mix lzma2_256k_bin lz4_compress_4m deflate_compress_64k; do
  cp $O/kernel_stats_$W.txt $P/${TAG}_kernel_stats_$W.txt 2>/dev/null
  cp $O/${W}_traffic.json $P/${TAG}_${W}_traffic.json 2>/dev/null
  cp $O/sq_counters_$W.txt $P/${TAG}_sq_counters_$W.txt 2>/dev/null
done
cp $O/kernel_stats_bzip2_compress.txt $P/${TAG}_kernel_stats_bzip2_compress.txt 2>/dev/null
cp $O/bzip2_compress_timing.txt $P/${TAG}_bzip2_compress_timing.txt 2>/dev/null
ls $P | grep "^$TAG"
