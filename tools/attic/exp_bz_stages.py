"""Time of the fused bzip2 kernel cut short after stage 1 / stage 2 (libraries built with -DSWC_BZ_STOP_AFTER=1 | 2:
tools/build_variant.sh; results are wrong by construction, only the clock is read).  Usage: SWC_LIB=... exp_bz_stages.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
w = bench.WORKLOADS["bzip2_900k"]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else w["n_units"] // 256
b, raw, plains, _ = bench.make_batch("bzip2_900k", w, list(w["parts"]), 1, "cuda:0", (0, 256 * tile))
for _ in range(2):
    b.launch(sync=True)
t0 = time.perf_counter()
for _ in range(3):
    b.launch(sync=True)
print("%s: %.1f ms per launch of %d blocks" % (os.environ.get("SWC_LIB", "shipped"), (time.perf_counter() - t0) / 3 * 1e3, b.n))
