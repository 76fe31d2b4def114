"""Single-shot latency of LZ4.decompress (host buffers both ways) by frame shape: what ONE block costs when the launch has nothing else."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swcompression_amd as swc
from swcompression_amd import corpus
for size, bs in ((65536, 65536), (1 << 20, 65536), (4 << 20, 4 << 20), (16 << 20, 4 << 20), (16 << 20, 65536)):
    p = corpus.p_text(size, 3)
    f = swc.LZ4.compress(p, block_size=bs)
    ts = []
    for k in range(8):
        t0 = time.perf_counter()
        out = swc.LZ4.decompress(f)
        ts.append(time.perf_counter() - t0)
    assert out == p
    ts.sort()
    print("%8d bytes in blocks of %7d: median %.2f ms (%.2f GiB/s)" % (size, bs, ts[len(ts) // 2] * 1e3, size / ts[len(ts) // 2] / 2**30))
