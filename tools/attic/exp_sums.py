"""Throughput of the device checksum kernels on decoded batches (100,000 x 64 KiB members; 2,048 x 4 MiB blocks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

for name, scale in (("deflate64k", 1.0), ("lz4_4m", 0.25)):
    w = bench.WORKLOADS[name]
    nd = max(8, int(w["n_distinct"] * scale))
    b, raw, plains = bench.make_batch(name, w, nd, 2, "cuda:0")
    b.launch(sync=True)
    assert (b.results()["status"] == 0).all()
    total = sum(len(p) for p in plains) * w["tile"]
    for kind in ("crc32", "adler32", "crc64", "bzip2crc32", "xxh32"):
        b.checksum(kind)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.checksum(kind)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%s %-11s %.2f ms  %.0f GB/s" % (name, kind, dt * 1e3, total / dt / 1e9))
