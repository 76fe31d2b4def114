#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_lz4.py tests/test_gpu_many.py -x -q 2>&1 | tail -3
timeout 300 python - <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, ".")
import bench
from swcompression_amd import _lib, corpus
lib = _lib.load()
units, plains = corpus.build_units("gzip", 8, 65536, seed=2)
raw = [u[10:-8] for u in units]
print(bench.config1_latency(lib, raw, plains, reps=50))
PY
