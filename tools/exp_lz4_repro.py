"""Bisect helper: which units of the 64 KiB 'mix' LZ4 corpus make the batch launch fail (each try in its own process)."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    sys.path.insert(0, ROOT)
    from swcompression_amd import corpus
    from swcompression_amd.batch import DeviceBatch
    units, plains = corpus.build_units("lz4_block", 2048, 65536, payload="mix")
    b = DeviceBatch("lz4_block", units[lo:hi], [65536] * (hi - lo))
    b.launch(sync=True)
    r = b.results()
    bad = [lo + i for i in range(hi - lo) if r["status"][i] != 0 or b.output(i, 65536) != plains[lo + i]]
    print("range", lo, hi, "bad", bad[:8])
    sys.exit(0)
def ok(lo, hi):
    p = subprocess.run([sys.executable, __file__, str(lo), str(hi)], capture_output=True, text=True)
    tail = (p.stdout + p.stderr).strip().splitlines()[-3:]
    print(lo, hi, "rc", p.returncode, tail, flush=True)
    return p.returncode == 0
lo, hi = 0, 2048
if ok(lo, hi):
    sys.exit(0)
while hi - lo > 1:
    mid = (lo + hi) // 2
    if not ok(lo, mid): hi = mid
    elif not ok(mid, hi): lo = mid
    else:
        print("only the combination fails", lo, hi); break
print("culprit range", lo, hi)
