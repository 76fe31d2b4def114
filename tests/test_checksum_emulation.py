"""Group-per-stream checksum kernels (crc32_group.h, checksum_group.h) compiled for the host and run with 64 real threads
per group, against the oracle's byte-at-a-time restatements of CheckSums.swift / XxHash32.swift."""
import numpy as np
import pytest

import _emu as E
import _oracle as O

ORACLE = {1: O.crc32, 2: O.adler32, 3: O.crc64, 4: O.bzip2crc32, 5: O.xxh32}
NAMES = {1: "crc32", 2: "adler32", 3: "crc64", 4: "bzip2crc32", 5: "xxh32"}


def payloads():
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 77))
    out = [b"", b"a", b"123456789", bytes(15), bytes(16), bytes(17)]
    for n in (63, 64, 65, 127, 128, 1023, 1024, 1025, 4096 + 3, 64 * 16, 64 * 16 + 1, 64 * 2048 + 5, 65536, 200001, 1 << 20):
        out.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    out.append(b"\xff" * 300000)   # Adler-32 worst case for the deferred modulo
    out.append(bytes(100000))
    return out


@pytest.mark.parametrize("kind", sorted(ORACLE))
def test_group_checksum_matches_oracle(kind):
    for p in payloads():
        assert E.checksum(kind, p) == ORACLE[kind](p), "%s, length %d" % (NAMES[kind], len(p))


def test_wave_crc32_matches_oracle():
    """crc32_wave.h (one stream per wave, the kernel behind swc_batch_crc32 for members below 1 MB): every length around the
    front padding to 2 KB rows and around the four bytes that carry the initial value, every start alignment."""
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 80))
    base = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    assert E.crc32_wave(b"123456789") == 0xCBF43926
    lengths = list(range(0, 80)) + [2044, 2045, 2046, 2047, 2048, 2049, 2050, 2051, 2052, 2053, 4095, 4096, 4097, 6143, 6144, 6150, 65535, 65536, 65537, 69999]
    for n in lengths:
        assert E.crc32_wave(base[:n]) == O.crc32(base[:n]), n
    for off in range(1, 17):
        p = base[off:off + 5000 + 3 * off]
        assert E.crc32_wave(p, misalign=off) == O.crc32(p), off
    for p in payloads():
        assert E.crc32_wave(p) == O.crc32(p), len(p)
    for order in (1, 2):
        E.set_order(order)
        try:
            assert E.crc32_wave(base[:9001]) == O.crc32(base[:9001])
        finally:
            E.set_order(0)


def test_known_answers():
    # the check values of the CRC catalogue / xxHash specification for "123456789"
    assert E.checksum(1, b"123456789") == 0xCBF43926
    assert E.checksum(3, b"123456789") == 0x995DC9BBDF1939FA
    assert E.checksum(4, b"123456789") == 0xFC891918
    assert E.checksum(2, b"Wikipedia") == 0x11E60398
    assert E.checksum(5, b"") == 0x02CC5D05


def test_unaligned_starts():
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 78))
    base = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    for off in (1, 3, 7, 13):
        for kind in sorted(ORACLE):
            p = base[off:off + 66000 + off]
            assert E.checksum(kind, p, misalign=off) == ORACLE[kind](p)


def test_delta_filter_group_matches_oracle():
    """delta_group.h against the oracle's restatement of DeltaFilter.swift:11-33: every distance class the
    scan treats differently (1 chunk per class .. 256 chunks per class), lengths around the chunking, in place."""
    import ctypes as C
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 79))
    O.lib.refcpu_delta_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
    O.lib.refcpu_delta_decode.restype = None
    for n in (0, 1, 2, 255, 256, 257, 1000, 65536, 100003):
        x = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for dist in (0, 1, 2, 3, 4, 7, 64, 100, 128, 129, 255):
            want = C.create_string_buffer(max(n, 1))
            O.lib.refcpu_delta_decode(x, n, dist, C.cast(want, C.c_void_p))
            assert E.delta(x, dist) == want.raw[:n], (n, dist)
            assert E.delta(x, dist, in_place=True) == want.raw[:n], (n, dist)
