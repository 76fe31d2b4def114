"""BZip2.compress on the device: wall time per call and size against libbz2 (GPU box).
    python tools/exp_bzip2_compress.py [MiB] [level]"""
import bz2
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import swcompression_amd as swc
from swcompression_amd import corpus

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
level = int(sys.argv[2]) if len(sys.argv) > 2 else 9
for name, gen in (("text", corpus.p_text), ("mix", corpus.p_mix)):
    x = b"".join(gen(1 << 20, 500 + i) for i in range(mib))
    swc.BZip2.compress(x[:1 << 20], level)          # warm-up: code objects, pool
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        z = swc.BZip2.compress(x, level)
        ts.append(time.perf_counter() - t)
    t = time.perf_counter()
    ref = bz2.compress(x[:8 << 20], level)
    tref = (time.perf_counter() - t) * (len(x) / (8 << 20))
    ok = bz2.decompress(z) == x
    print("%s %d MiB level %d: %.1f ms (best of 3: %s) = %.3f GiB/s; ratio %.4f, libbz2 %.4f (x%.3f), libbz2 one core %.1f s; round trip %s" % (
        name, mib, level, min(ts) * 1e3, ["%.0f" % (a * 1e3) for a in ts], len(x) / min(ts) / 2**30, len(z) / len(x),
        len(ref) / (8 << 20), (len(z) / len(x)) / (len(ref) / (8 << 20)), tref, ok))
