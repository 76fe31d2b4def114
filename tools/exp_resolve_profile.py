"""Wall-cycle split of the LZ resolve kernel per batch stage (library built with -DSWC_RESOLVE_PROFILE)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
names = ["top/fetch", "scan+A", "publish+lits", "barrier B", "search", "barrier C", "prefetch+jump", "copies", "barrier D", "stage+flush"]
for tile in (32,):
    b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
    b.launch(sync=True)
    stride = b.ws_bytes // b.n
    ws = b.d_ws.cpu().numpy()
    for wave in (0, 1):
        rows = []
        for j in range(0, b.n, max(1, b.n // 64)):
            end = j * stride + stride - 16 - 80 * wave
            rows.append(np.frombuffer(ws[end - 80:end].tobytes(), dtype=np.uint64))
        rows = np.array(rows).astype(np.float64)
        tot = rows.sum(axis=1).mean()
        print("jobs=%d wave %d: total %.0f kcycles per stream; per stage %%: " % (b.n, wave, tot / 1e3) + ", ".join("%s %.1f" % (n, 100 * rows[:, k].mean() / tot) for k, n in enumerate(names)))
