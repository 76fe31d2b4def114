"""Quick on-GPU experiments for the Deflate path (not part of the test suite)."""
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
for T in (512,):
    assert lib.swc_set_tuning(b"resolve_threads", T) == 0
    for tile, label in ((48, "96k"),):
        b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
        ms = timeit(b)
        r = b.results()
        ok = bool((r["status"] == 0).all()) and b.output(5, 65536) == plains[5] and b.output(len(raw) + 7, 65536) == plains[7]
        print("T=%d jobs=%-6s %8.2f ms  %.1f GB/s out  ok=%s" % (T, label, ms, b.n * 65536 / ms / 1e6, ok), flush=True)
        del b
