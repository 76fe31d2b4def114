"""Time of the bzip2 block kernel cut short after stage 1 / stage 2 (libraries built with -DSWC_BZ_STOP_AFTER=1 | 2:
tools/build_variant.sh; results are wrong by construction, only the clock is read: the first phase of swc_last_phase_ms is the
block kernel).  Usage: SWC_LIB=... exp_bz_stages.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from swcompression_amd import _lib
lib = _lib.load()
lib.swc_set_tuning(b"phase_timing", 1)
w = bench.WORKLOADS["bzip2_900k"]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else w["n_units"] // 256
b, raw, plains, _ = bench.make_batch("bzip2_900k", w, list(w["parts"]), 1, "cuda:0", (0, 256 * tile))
best = None
for _ in range(4):
    b.launch(sync=True)
    ms = (C.c_float * 8)()
    m = lib.swc_last_phase_ms(ms, 8)
    v = [round(ms[i], 1) for i in range(m)]
    if best is None or v[0] < best[0]:
        best = v
print("%s: phases (ms) %s, %d blocks" % (os.environ.get("SWC_LIB", "shipped"), best, b.n))
