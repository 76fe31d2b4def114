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
units, plains = corpus.build_units("gzip", 4000, 65536)
raw = [u[10:-8] for u in units]
for G in (1, 2, 4):
    assert lib.swc_set_tuning(b"inflate_lanes_per_stream", G) == 0
    for tile, label in ((25, "100k"), (4, "16k")):
        b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
        ms = timeit(b)
        r = b.results()
        ok = (r["status"] == 0).all() and b.output(5) == plains[5] and b.output(b.n - 1) == plains[-1]
        print("G=%d full   jobs=%-6s %8.2f ms  %.1f GB/s out ok=%s" % (G, label, ms, b.n * 65536 / ms / 1e6, ok))
        del b
b = DeviceBatch("deflate", raw, [0] * len(raw), tile=25)
ms = timeit(b)
print("count-only (cap=0) 100k %8.2f ms  %.1f GB/s out-equivalent" % (ms, b.n * 65536 / ms / 1e6))
