import bz2, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import swcompression_amd as swc
from swcompression_amd import _lib, corpus
lib = _lib.load()
for name, x in (("500 B", b"hello" * 100), ("100 kB", corpus.p_text(100000, 1)), ("900 kB", corpus.p_text(899000, 2)), ("9 x 900 kB", corpus.p_text(9 * 899000, 3))):
    z = bz2.compress(x, 9)
    row = []
    for mode in (0, 2):
        lib.swc_set_tuning(b"bzip2_team_walk", mode)
        assert swc.BZip2.decompress(z) == x
        ts = []
        for _ in range(7):
            t = time.perf_counter(); swc.BZip2.decompress(z); ts.append((time.perf_counter() - t) * 1e3)
        row.append(min(ts))
    print("%-12s fused %.3f ms, team %.3f ms" % (name, row[0], row[1]))
