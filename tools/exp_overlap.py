"""Does the CRC-32 pass (HBM-bound) hide under the decode kernels (VALU-bound) when it runs on a second HIP stream?
The headline workload cut into K slices; decode of slice i on stream A, its CRC on stream B behind an event."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
TILE = 48
for K in (1, 2, 4, 8):
    bs = [DeviceBatch("deflate", raw, [65536] * len(raw), tile=TILE // K) for _ in range(K)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(K)]

    def serial():
        for b in bs:
            b.launch()
            b.crc32_async()

    def overlapped():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        for b, e in zip(bs, evs):
            with torch.cuda.stream(sa):
                b.launch()
                e.record(sa)
            with torch.cuda.stream(sb):
                sb.wait_event(e)
                b.crc32_async()
        cur.wait_stream(sa); cur.wait_stream(sb)

    for name, fn in (("serial", serial), ("two streams", overlapped)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print("K=%d %-12s %.2f ms/step" % (K, name, ms), flush=True)
    ref = bs[0].crc32()
    bs[0].crc32_async(); torch.cuda.synchronize()
    assert (bs[0]._crc_buf.cpu().numpy().view(np.uint32) == ref).all()
    del bs
    torch.cuda.empty_cache()
