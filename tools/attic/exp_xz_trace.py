"""The xz line of bench.py's archive_paths alone; with SWC_TRACE=1 the library prints the host-side stages of every call."""
import os, shutil, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import swcompression_amd as swc
from swcompression_amd import corpus
x = b"".join(corpus.p_text(262144, 0x5C0DE + 5 + i) for i in range(64)) * 8
a = subprocess.run(["xz", "-z", "-c", "-T4", "--block-size=262144", "--check=crc64"], input=x, stdout=subprocess.PIPE, check=True).stdout
swc.XZArchive.unarchive(a)
for _ in range(3):
    t = time.perf_counter(); y = swc.XZArchive.unarchive(a); dt = time.perf_counter() - t
    print("xz unarchive %.1f ms" % (dt * 1e3), file=sys.stderr)
assert y == x
