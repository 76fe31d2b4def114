"""Quick on-GPU experiments for the Deflate kernel (not part of the test suite)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

def timeit(b, reps=3):
    b.launch(sync=True)
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); b.launch(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts)

from swcompression_amd import _lib
lib = _lib.load()
units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
for G in (1, 2, 4):
    assert lib.swc_set_tuning(b"inflate_lanes_per_stream", G) == 0
    for mode in (0, 1, 2):
        if G > 1 and mode: continue
        assert lib.swc_set_tuning(b"inflate_debug_mode", mode) == 0
        for tile, label in ((32, "64k"), (16, "32k"), (8, "16k"), (2, "4k")):
            b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
            ms = timeit(b)
            print("G=%d mode=%d jobs=%-6s %8.2f ms  %.1f GB/s out" % (G, mode, label, ms, b.n * 65536 / ms / 1e6), flush=True)
            del b
