"""CPU tier: the C-ABI library builds for gfx950, loads, exports every symbol include/swc_hip.h declares,
its host-only helpers (checksums) work, and -- with no GPU in this container -- every decode entry point
fails loudly with SWC_E_DEVICE instead of falling back to a CPU decoder."""
import ctypes as C
import hashlib
import os
import re
import zlib

import pytest

import swcompression_amd as swc
from swcompression_amd import _lib, corpus
from swcompression_amd.zipcontainer import ZipContainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "swc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(swc_[a-z0-9_]+)\s*\(", hdr)))


def test_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libswc_hip.so does not export %s" % n


def test_memory_knobs_without_a_device():
    """swc_trim() and the three limits of what the library keeps between calls work -- and do nothing harmful -- on a
    box without a GPU: nothing is parked, nothing pinned, no pool to trim."""
    lib = _lib.load()
    assert lib.swc_trim() == 0
    for key in (b"pool_keep_mib", b"pinned_keep_mib", b"result_cache_mib"):
        assert lib.swc_set_tuning(key, 256) == 0 and lib.swc_set_tuning(key, -1) != 0
    assert lib.swc_set_tuning(b"pool_keep_mib", 2048) == 0 and lib.swc_set_tuning(b"pinned_keep_mib", 512) == 0
    assert lib.swc_set_tuning(b"result_cache_mib", 512) == 0
    assert swc.trim() == 0


def test_status_table_matches_header():
    hdr = open(os.path.join(ROOT, "include", "swc_status.h")).read()
    codes = {int(v) for v in re.findall(r"=\s*(\d+)", hdr)} - {0}
    assert codes == set(swc.STATUS)


def test_job_layout():
    assert C.sizeof(_lib.SwcJob) == 72 and _lib.SwcJob.status.offset == 48 and _lib.SwcJob.dict_len.offset == 64


def test_host_checksums():
    lib = _lib.load()
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 5551, 5552, 5553, 100000):
        x = corpus.p_mix(n, 5)
        assert lib.swc_crc32(x, n, 0) == zlib.crc32(x) & 0xFFFFFFFF
        assert lib.swc_adler32(x, n) == zlib.adler32(x) & 0xFFFFFFFF
        d = C.create_string_buffer(32)
        lib.swc_sha256(x, n, d)
        assert d.raw == hashlib.sha256(x).digest()
    assert lib.swc_crc32(b"world", 5, zlib.crc32(b"hello ")) == zlib.crc32(b"hello world")
    assert lib.swc_crc64(b"123456789", 9) == 0x995DC9BBDF1939FA
    assert lib.swc_bzip2_crc32(b"123456789", 9) == 0xFC891918
    assert lib.swc_xxh32(b"abc", 3, 0) == 0x32D153FF
    assert lib.swc_xxh32(b"1234567890" * 8, 80, 0) == 0x9C05F475


def test_code_object_is_gfx950_only():
    """Every code object bundled into the library targets gfx950 and nothing else.  (Looked up in the offload-bundle entries and
    the code objects' target strings, not as bare text: the host side of the rocPRIM sort/scan dispatch that the BZip2 encoder
    uses carries a table of architecture NAMES for its tuning configs -- names, not code.)"""
    import re
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--([A-Za-z0-9_]+)", blob))
    assert targets == {b"gfx950"}, targets
    for other in (b"sm_90", b"nvptx", b"__CUDA"):
        assert other not in blob


def _one_entry_zip(x):
    import io
    import zipfile
    b = io.BytesIO()
    with zipfile.ZipFile(b, "w", compression=zipfile.ZIP_DEFLATED) as z:
        z.writestr("x.txt", x)
    return b.getvalue()


@pytest.mark.skipif(swc.device_available(), reason="GPU present: covered by the gpu tier")
def test_no_cpu_fallback_without_gpu():
    x = corpus.p_text(1000, 1)
    calls = [
        lambda: swc.Deflate.decompress(corpus.deflate_raw(x)),
        lambda: swc.GzipArchive.unarchive(corpus.gzip_member(x)),
        lambda: swc.ZlibArchive.unarchive(zlib.compress(x)),
        lambda: swc.BZip2.decompress(corpus.bzip2_stream(x)),
        lambda: swc.LZMA.decompress(corpus.lzma_alone(x)),
        lambda: swc.XZArchive.unarchive(corpus.xz_stream(x)),
        lambda: swc.LZ4.decompress(corpus.lz4_frame(x)),
        lambda: swc.unarchive_many("gzip", [corpus.gzip_member(x)]),
        lambda: swc.unarchive_many("bzip2", [corpus.bzip2_stream(x)]),
        lambda: swc.unarchive_many("gzip", [corpus.gzip_member(x)] * 3, devices=[0, 1]),
        lambda: ZipContainer.open(_one_entry_zip(x)),
    ]
    for c in calls:
        with pytest.raises(swc.DeviceError):
            c()


def test_every_exported_function_is_documented():
    """INTEGRATION.md / DESIGN.md mention every entry point of include/swc_hip.h (the reference-side binding is part of the
    boundary, not an afterthought)."""
    docs = open(os.path.join(ROOT, "INTEGRATION.md")).read() + open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [n for n in _declared() if n not in docs]
    assert not missing, missing
