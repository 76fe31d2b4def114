"""Deterministic synthetic corpus for the five BASELINE.json configs.

Payload classes mirror the reference's fixture recipes (Tests/Constants.swift:10-20):
  P-text  Zipf-distributed pseudo-words (ratio ~3x under deflate level 6)
  P-rep   short phrase repeated
  P-zero  zeros
  P-rand  uniform random bytes (forces stored / incompressible paths)
  P-mix   50 % text / 25 % rep / 25 % rand spliced at 4 KiB granularity

The reference's own compressors cannot produce dynamic-Huffman / LZMA2 / XZ inputs,
so zlib / bz2 / lzma / liblz4 act as ENCODERS here.  Everything is seeded: numpy PCG64, base 0x5C0DE.
"""
import bz2
import ctypes
import ctypes.util
import lzma
import struct
import zlib

import numpy as np

SEED_BASE = 0x5C0DE

_VOCAB = None


def _vocab():
    global _VOCAB
    if _VOCAB is None:
        rng = np.random.Generator(np.random.PCG64(SEED_BASE))
        v = 3000
        lens = rng.integers(2, 11, size=v)
        table = np.zeros((v, 12), dtype=np.uint8)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        lp = 1.0 / np.arange(1, 27) ** 0.8
        lp /= lp.sum()
        for i in range(v):
            w = rng.choice(letters, size=lens[i], p=lp)
            table[i, : lens[i]] = w
            table[i, lens[i]] = 0x20
        # a few "punctuation" words
        for i, tok in enumerate([b". ", b", ", b";\n", b"\n", b"(", b") ", b"= ", b"0x", b"1", b"42 "]):
            table[i * 37 + 5, :] = 0
            table[i * 37 + 5, : len(tok)] = np.frombuffer(tok, dtype=np.uint8)
            lens[i * 37 + 5] = len(tok) - 1
        p = 1.0 / np.arange(1, v + 1) ** 1.05
        p /= p.sum()
        _VOCAB = (table, lens + 1, p)
    return _VOCAB


def p_text(n, seed):
    table, wl, p = _vocab()
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    nwords = n // 4 + 16
    idx = rng.choice(len(wl), size=nwords, p=p)
    lens = wl[idx]
    starts = np.cumsum(lens) - lens
    total = int(starts[-1] + lens[-1])
    ci = np.arange(total) - np.repeat(starts, lens)
    out = table[np.repeat(idx, lens), ci]
    while out.size < n:  # pragma: no cover (nwords is generous)
        out = np.concatenate([out, out])
    return out[:n].tobytes()


def p_rep(n, seed):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    phrase = bytes(rng.integers(32, 127, size=int(rng.integers(8, 48)), dtype=np.uint8))
    return (phrase * (n // len(phrase) + 1))[:n]


def p_zero(n, seed=0):
    return bytes(n)


def p_rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


def p_mix(n, seed):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    text, rep, rand = p_text(n, seed), p_rep(n, seed + 1), p_rand(n, seed + 2)
    parts = []
    for off in range(0, n, 4096):
        k = rng.integers(0, 4)
        src = text if k < 2 else rep if k == 2 else rand
        parts.append(src[off:off + 4096])
    return b"".join(parts)[:n]


def p_bin(n, seed):
    """Binary records: 16-byte rows of a little-endian counter, two 16-bit samples of a random walk, a float32 of a slowly
    varying value and four bytes of a small alphabet with the top bit set -- compressible, and its bytes fall into all eight
    classes of "top three bits of the previous byte" (the worst case of the LZMA literal-coder cache)."""
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + seed))
    rows = n // 16 + 1
    rec = np.zeros((rows, 16), dtype=np.uint8)
    rec[:, 0:4] = (np.arange(rows, dtype=np.uint32) * 3 + int(rng.integers(0, 1 << 20))).view(np.uint8).reshape(rows, 4)
    walk = np.cumsum(rng.integers(-40, 41, size=(rows, 2)), axis=0).astype(np.int16)
    rec[:, 4:8] = walk.view(np.uint8).reshape(rows, 4)
    val = (1000.0 + np.cumsum(rng.normal(0, 0.01, rows))).astype(np.float32)
    rec[:, 8:12] = val.view(np.uint8).reshape(rows, 4)
    rec[:, 12:16] = rng.choice(np.array([0x80, 0x9C, 0xA5, 0xC3, 0xE2, 0xFF, 0x41, 0x0A], dtype=np.uint8), size=(rows, 4))
    return rec.reshape(-1)[:n].tobytes()


PAYLOADS = {"text": p_text, "rep": p_rep, "zero": p_zero, "rand": p_rand, "mix": p_mix, "bin": p_bin}


# ------------------------------------------------------------------------------- Deflate / gzip / zlib
def deflate_raw(payload, level=6):
    """Raw RFC-1951 stream.  memLevel 9 keeps 64 KiB of text in ONE dynamic-Huffman block (config 1)."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    return c.compress(payload) + c.flush()


def gzip_member(payload, level=6, bgzf=False):
    raw = deflate_raw(payload, level)
    trailer = struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF)
    if bgzf:
        # BGZF: FEXTRA with subfield 'B','C', len 2, BSIZE = total member size - 1
        total = 18 + len(raw) + 8
        if total > 65536:
            raise ValueError("BGZF members are limited to 64 KiB (BSIZE is a u16)")
        hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1)
    else:
        hdr = b"\x1f\x8b\x08\x00" + b"\0\0\0\0" + b"\x00\xff"
    return hdr + raw + trailer


def zlib_stream(payload, level=6):
    return zlib.compress(payload, level)


# ------------------------------------------------------------------------------- LZ4
_lz4 = None


def _liblz4():
    global _lz4
    if _lz4 is None:
        name = ctypes.util.find_library("lz4")
        if name is None:
            raise RuntimeError("liblz4 not found (needed only as an ENCODER for the synthetic corpus)")
        _lz4 = ctypes.CDLL(name)
        _lz4.LZ4_compressBound.restype = ctypes.c_int
        _lz4.LZ4_compressBound.argtypes = [ctypes.c_int]
        _lz4.LZ4_compress_default.restype = ctypes.c_int
        _lz4.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        _lz4.LZ4_compress_fast_continue.restype = ctypes.c_int
    return _lz4


def lz4_block(payload):
    lib = _liblz4()
    bound = lib.LZ4_compressBound(len(payload))
    dst = ctypes.create_string_buffer(bound)
    n = lib.LZ4_compress_default(payload, dst, len(payload), bound)
    if n <= 0:
        raise RuntimeError("LZ4_compress_default failed")
    return dst.raw[:n]


def _xxh32(data, seed=0):
    # small pure-python XXH32 for frame headers only (a few bytes)
    P1, P2, P3, P4, P5 = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F, 0x165667B1
    M = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
    n = len(data)
    i = 0
    if n >= 16:
        v = [(seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M]
        while n - i >= 16:
            for j in range(4):
                lane = int.from_bytes(data[i + 4 * j:i + 4 * j + 4], "little")
                v[j] = (rotl((v[j] + lane * P2) & M, 13) * P1) & M
            i += 16
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
    else:
        h = (seed + P5) & M
    h = (h + n) & M
    while n - i >= 4:
        h = (rotl((h + int.from_bytes(data[i:i + 4], "little") * P3) & M, 17) * P4) & M
        i += 4
    while i < n:
        h = (rotl((h + data[i] * P5) & M, 11) * P1) & M
        i += 1
    h ^= h >> 15
    h = (h * P2) & M
    h ^= h >> 13
    h = (h * P3) & M
    h ^= h >> 16
    return h


def lz4_frame(payload, block_size_code=7, independent=True, content_checksum=False, block_checksum=False,
              content_size=False, xxh32=None):
    """LZ4 frame v1.x written by hand (FLG/BD per LZ4.swift:196-228), blocks from LZ4_compress_default.

    Only independent blocks are produced here (dependent blocks need the streaming encoder)."""
    assert independent
    xxh = xxh32 or _xxh32
    max_block = {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}[block_size_code]
    flg = 0x40 | (0x20 if independent else 0) | (0x10 if block_checksum else 0) | (0x08 if content_size else 0) | (0x04 if content_checksum else 0)
    desc = bytes([flg, block_size_code << 4])
    if content_size:
        desc += struct.pack("<Q", len(payload))
    out = [struct.pack("<I", 0x184D2204), desc, bytes([(xxh(desc) >> 8) & 0xFF])]
    for off in range(0, len(payload), max_block):
        chunk = payload[off:off + max_block]
        comp = lz4_block(chunk)
        if len(comp) >= len(chunk):
            out.append(struct.pack("<I", len(chunk) | 0x80000000))
            body = chunk
        else:
            out.append(struct.pack("<I", len(comp)))
            body = comp
        out.append(body)
        if block_checksum:
            out.append(struct.pack("<I", xxh(body)))
    out.append(struct.pack("<I", 0))
    if content_checksum:
        out.append(struct.pack("<I", xxh(payload)))
    return b"".join(out)


# ------------------------------------------------------------------------------- BZip2 / LZMA / XZ
def bzip2_stream(payload, level=9):
    return bz2.compress(payload, level)


def lzma2_raw(payload, preset=6, dict_size=1 << 20):
    """Raw LZMA2 stream (XZ block body) -- config 5 unit."""
    return lzma.compress(payload, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": preset, "dict_size": dict_size}])


def lzma2_dict_byte(dict_size):
    for bits in range(40):
        if ((2 | (bits & 1)) << (bits // 2 + 11)) >= dict_size:
            return bits
    return 39


def lzma_alone(payload, preset=6):
    return lzma.compress(payload, format=lzma.FORMAT_ALONE, preset=preset)


def xz_stream(payload, preset=6, check=lzma.CHECK_CRC64, filters=None):
    if filters is not None:
        return lzma.compress(payload, format=lzma.FORMAT_XZ, check=check, filters=filters)
    return lzma.compress(payload, format=lzma.FORMAT_XZ, check=check, preset=preset)


# ------------------------------------------------------------------------------- bench / test corpora
def _cache_path(tag):
    import os
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "swc_corpus_cache")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, tag + ".npz")


def build_units(kind, n_distinct, unit_size, payload="text", seed=2, cache=True):
    """`n_distinct` independent compressed units of `unit_size` uncompressed bytes each.

    kind: 'deflate' (raw RFC 1951, one dynamic block at 64 KiB), 'gzip', 'lz4_block', 'bzip2', 'lzma2'.
    Returns (units: list[bytes], plains: list[bytes]).  Cached under $TMPDIR because zlib/bz2/lzma
    as encoders dominate the set-up time of tests and bench."""
    import os
    tag = "%s_%s_%d_%d_%d" % (kind, payload, n_distinct, unit_size, seed)
    path = _cache_path(tag)
    if cache and os.path.exists(path):
        try:
            z = np.load(path)
            ub, uo, pb, po = z["ub"], z["uo"], z["pb"], z["po"]
            units = [ub[uo[i]:uo[i + 1]].tobytes() for i in range(n_distinct)]
            plains = [pb[po[i]:po[i + 1]].tobytes() for i in range(n_distinct)]
            return units, plains
        except Exception:
            pass  # a damaged cache file (interrupted writer): rebuild below
    gen = PAYLOADS[payload]
    enc = {"deflate": deflate_raw, "gzip": gzip_member, "lz4_block": lz4_block, "bzip2": bzip2_stream,
           "lzma2": lzma2_raw}[kind]
    # the encoders (zlib / bz2 / lzma / liblz4 through ctypes) release the GIL: a thread pool uses the host cores
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(32, (os.cpu_count() or 1)))
    with ThreadPoolExecutor(workers) as ex:
        plains = list(ex.map(lambda i: gen(unit_size, seed + i), range(n_distinct)))
        units = list(ex.map(enc, plains))
    if cache:
        uo = np.cumsum([0] + [len(u) for u in units])
        po = np.cumsum([0] + [len(p) for p in plains])
        tmp = "%s.%d.tmp.npz" % (path[:-4], os.getpid())   # written aside and renamed: concurrent ranks never see half a file
        np.savez(tmp, ub=np.frombuffer(b"".join(units), dtype=np.uint8), uo=uo,
                 pb=np.frombuffer(b"".join(plains), dtype=np.uint8), po=po)
        os.replace(tmp, path)
    return units, plains


def build_units_mixed(kind, parts, unit_size, seed=2, cache=True):
    """Units of several payload classes in one list: parts = [("text", 192), ("mix", 64), ...] (class, count).  The
    classes are interleaved so that any contiguous range of the list (a rank's shard) holds all of them."""
    lists = []
    for k, (payload, n) in enumerate(parts):
        u, p = build_units(kind, n, unit_size, payload=payload, seed=seed + 100003 * k, cache=cache)
        lists.append(list(zip(u, p)))
    total = sum(len(x) for x in lists)
    out, taken = [], [0] * len(lists)
    for i in range(total):   # proportional interleave
        k = max(range(len(lists)), key=lambda j: (len(lists[j]) * (i + 1) / total - taken[j]) if taken[j] < len(lists[j]) else -1e9)
        out.append(lists[k][taken[k]])
        taken[k] += 1
    return [u for u, _ in out], [p for _, p in out]


class _LZ4F_frameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int), ("contentChecksumFlag", ctypes.c_int),
                ("frameType", ctypes.c_int), ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_int)]


class _LZ4F_preferences(ctypes.Structure):
    _fields_ = [("frameInfo", _LZ4F_frameInfo), ("compressionLevel", ctypes.c_int), ("autoFlush", ctypes.c_uint),
                ("favorDecSpeed", ctypes.c_uint), ("reserved", ctypes.c_uint * 3)]


def lz4f_frame(payload, block_size_code=4, linked=True, content_checksum=False, block_checksum=False, content_size=False):
    """A frame produced by liblz4's own frame API (LZ4F_compressFrame) -- the only way to get
    DEPENDENT (linked) blocks, which the reference decodes with a 64 KiB sliding prefix (LZ4.swift:306-313)."""
    lib = _liblz4()
    lib.LZ4F_compressFrameBound.restype = ctypes.c_size_t
    lib.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    lib.LZ4F_compressFrame.restype = ctypes.c_size_t
    lib.LZ4F_compressFrame.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.LZ4F_isError.restype = ctypes.c_uint
    lib.LZ4F_isError.argtypes = [ctypes.c_size_t]
    prefs = _LZ4F_preferences()
    prefs.frameInfo.blockSizeID = block_size_code
    prefs.frameInfo.blockMode = 0 if linked else 1
    prefs.frameInfo.contentChecksumFlag = 1 if content_checksum else 0
    prefs.frameInfo.blockChecksumFlag = 1 if block_checksum else 0
    prefs.frameInfo.contentSize = len(payload) if content_size else 0
    bound = lib.LZ4F_compressFrameBound(len(payload), ctypes.byref(prefs))
    dst = ctypes.create_string_buffer(bound)
    n = lib.LZ4F_compressFrame(dst, bound, payload, len(payload), ctypes.byref(prefs))
    if lib.LZ4F_isError(n):
        raise RuntimeError("LZ4F_compressFrame failed")
    return dst.raw[:n]
