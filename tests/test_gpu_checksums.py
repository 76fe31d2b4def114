"""swc_batch_checksum on the GPU: CRC-32, Adler-32, CRC-64, bzip2 CRC-32 and XXH32 of decoded outputs against the oracle's
restatements of CheckSums.swift:12-57 and XxHash32.swift:24-83."""
import numpy as np
import pytest

import _oracle as O
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu

ORACLE = {"crc32": O.crc32, "adler32": O.adler32, "crc64": O.crc64, "bzip2crc32": O.bzip2crc32, "xxh32": O.xxh32}


def _decoded_batch(plains):
    streams = [corpus.deflate_raw(p) for p in plains]
    b = DeviceBatch("deflate", streams, [max(len(p), 1) for p in plains])
    b.launch(sync=True)
    assert (b.results()["status"] == 0).all()
    return b


def test_every_checksum_on_awkward_lengths():
    sizes = [0, 1, 2, 3, 4, 5, 7, 15, 16, 17, 31, 32, 33, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4099, 65535, 65536,
             65537, 300001, 1 << 20, (1 << 21) + 11,
             # swc_batch_crc32: around the rows of 2 KB of the wave kernel and around its hand-over to the group kernel at 1 MB
             2044, 2045, 2047, 2048, 2049, 2051, 6144, 6145, (1 << 20) - 1, (1 << 20) + 1, 3 << 20]
    plains = [corpus.p_mix(n, 140 + i) for i, n in enumerate(sizes)]
    plains.append(b"\xff" * 700001)    # Adler-32: largest per-byte increments
    plains.append(bytes(500000))
    b = _decoded_batch(plains)
    for kind, fn in ORACLE.items():
        got = b.checksum(kind)
        for i, p in enumerate(plains):
            assert int(got[i]) == fn(p), "%s, length %d" % (kind, len(p))
    got = b.crc32()   # swc_batch_crc32: its own pair of kernels
    for i, p in enumerate(plains):
        assert int(got[i]) == ORACLE["crc32"](p), "swc_batch_crc32, length %d" % len(p)


def test_checksums_of_many_members():
    """4,101 members of about 64 KiB: more groups than the chip holds at once; 16 streams per wave in the XXH32 kernel
    with a partially filled last wave."""
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 91))
    base = [corpus.p_text(65536 - int(rng.integers(0, 40)), 3000 + k) for k in range(97)]
    n = 4096 + 5
    b = _decoded_batch([base[i % 97] for i in range(n)])
    for kind, fn in ORACLE.items():
        got = b.checksum(kind)
        want = np.array([fn(p) for p in base], dtype=np.uint64)
        assert (got == want[np.arange(n) % 97]).all(), kind


def test_invalid_kind_is_rejected():
    b = _decoded_batch([b"abc"])
    with pytest.raises(KeyError):
        b.checksum("md5")
    assert b.lib.swc_batch_checksum(99, b.d_jobs.data_ptr(), 1, b.d_jobs.data_ptr(), None) == 903   # SWC_E_INVALID_ARGUMENT


def test_delta_filter_codec():
    """SWC_CODEC_DELTA (DeltaFilter.swift:11-33) through the batch API: every distance, awkward lengths, and in place."""
    import ctypes as C
    import _sevenzip as Z7
    O.lib.refcpu_delta_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
    O.lib.refcpu_delta_decode.restype = None
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 93))
    units, dists = [], []
    for dist in list(range(0, 256, 7)) + [1, 2, 3, 4, 255]:
        n = int(rng.choice([0, 1, 255, 256, 257, 5000, 65536, 300001]))
        units.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        dists.append(dist)
    x = corpus.p_text(200000, 5)
    units.append(Z7.delta_encode(x, 4)); dists.append(4)          # the encoder's inverse: decode gives the text back
    b = DeviceBatch("delta", units, [max(len(u), 1) for u in units], aux=dists)
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == [len(u) for u in units]).all()
    for i, (u, d) in enumerate(zip(units, dists)):
        want = C.create_string_buffer(max(len(u), 1))
        O.lib.refcpu_delta_decode(u, len(u), d, C.cast(want, C.c_void_p))
        assert b.output(i, len(u)) == want.raw[:len(u)], (len(u), d)
    assert b.output(len(units) - 1, len(x)) == x
    # in place: out == in (the encoded bytes are put where the outputs were, then decoded over themselves)
    import torch
    jobs = b.results().copy()
    host = b.d_out.cpu().numpy().copy()
    base = b.d_out.data_ptr()
    for i, u in enumerate(units):
        off = int(jobs["out"][i]) - base
        host[off:off + len(u)] = np.frombuffer(u, dtype=np.uint8)
    b.d_out.copy_(torch.from_numpy(host))
    jobs["in"] = jobs["out"]
    jobs["status"] = 902
    b.d_jobs.copy_(torch.from_numpy(jobs.view(np.uint8).copy()))
    b.launch(sync=True)
    assert (b.results()["status"] == 0).all()
    for i, (u, d) in enumerate(zip(units, dists)):
        want = C.create_string_buffer(max(len(u), 1))
        O.lib.refcpu_delta_decode(u, len(u), d, C.cast(want, C.c_void_p))
        assert b.output(i, len(u)) == want.raw[:len(u)], ("in place", len(u), d)
