"""Statistics of the sub-chunk-parallel Deflate decoder from the host emulation (no GPU): rounds, passes, code iterations
per kind of pass, and wave-steps (a pass takes as long as its busiest lane) against the perfectly balanced figure."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _emu as E
from swcompression_amd import corpus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
units, plains = corpus.build_units("deflate", n, 65536, seed=2)
lib = E.load() if hasattr(E, "load") else E.lib
st, wv = (C.c_uint64 * 8)(), (C.c_uint64 * 4)()
lib.emu_sync_stats(st, 1)
lib.emu_sync_wave(wv)
assert all(r[0] == 0 for r in E.inflate(units, [65536] * n))
lib.emu_sync_stats(st, 1)
lib.emu_sync_wave(wv)
names = ["rounds", "bails", "passes", "lane decodes", "walk+count code iterations", "emit code iterations", "check iterations", "long codes"]
for k, v in zip(names, st):
    print("%-28s %10.1f per stream" % (k, v / n))
print("wave-steps per stream: walk %.0f, count %.0f, emit %.0f; balanced: %.0f per full decode"
      % (wv[0] / n, wv[1] / n, wv[2] / n, st[5] / n / 64))
