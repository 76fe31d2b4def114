"""One small Deflate launch (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
units, plains = corpus.build_units("gzip", 4000, 65536)
raw = [u[10:-8] for u in units][:n]
b = DeviceBatch("deflate", raw, [cap] * len(raw), tile=1)
b.launch(sync=True)
b.launch(sync=True)
