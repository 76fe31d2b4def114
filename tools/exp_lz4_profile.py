"""Cycle split of the LZ4 parse kernel (library built with -DSWC_LZ4_PROFILE; not a production build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
units, plains = corpus.build_units("lz4_block", 8, 4 << 20)
for tile in (32, 256):
    b = DeviceBatch("lz4_block", units, [4 << 20] * len(units), tile=tile)
    b.launch(sync=True)
    stride = b.ws_bytes // b.n
    ws = b.d_ws.cpu().numpy()
    tail = ((4 << 20) + 16 + 15) // 16 * 16
    rows = []
    for j in range(0, b.n, max(1, b.n // 16)):
        base = j * stride + stride - 32
        rows.append(np.frombuffer(ws[base:base + 32].tobytes(), dtype=np.uint64))
    rows = np.array(rows).astype(np.float64)
    print("blocks=%d  stripes/blk %.0f  cycles/stripe %.0f  careful/blk %.0f  cycles/careful %.0f  -> stripe Mcyc %.1f careful Mcyc %.1f" %
          (b.n, rows[:, 1].mean(), (rows[:, 0] / rows[:, 1]).mean(), rows[:, 3].mean(), (rows[:, 2] / rows[:, 3]).mean(), rows[:, 0].mean() / 1e6, rows[:, 2].mean() / 1e6))
    del b
