#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q -k "crc32 or config2" 2>&1 | tail -3
python - <<'P'
import sys; sys.path.insert(0,'.')
import torch, numpy as np
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch
units, plains = corpus.build_units("gzip", 2048, 65536)
raw=[u[10:-8] for u in units]
b=DeviceBatch("deflate", raw, [65536]*len(raw), tile=32); b.launch(sync=True)
b.crc32()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record(); b.crc32(); e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1); print("crc32 of %d x 64 KiB: %.2f ms  %.0f GB/s" % (b.n, ms, b.n*65536/ms/1e6))
P
