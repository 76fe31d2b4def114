"""Streams BUILT for the soak tests (test_gpu_soak.py on the device, test_lane_emulation_soak.py on the host emulation): LZ4 blocks
assembled sequence by sequence around every boundary of the record formats, plain text spliced for the Deflate encoder, damage."""
import random
import struct
import zlib

from swcompression_amd import corpus


def ext(v):
    out = bytearray()
    while v >= 255:
        out.append(255)
        v -= 255
    out.append(v)
    return bytes(out)


def sequence(lits, offset=None, mlen=None):
    ll = len(lits)
    ml = 0 if mlen is None else mlen - 4
    s = bytes([(min(ll, 15) << 4) | min(ml, 15)]) + (ext(ll - 15) if ll >= 15 else b"") + lits
    if mlen is not None:
        s += struct.pack("<H", offset) + (ext(ml - 15) if ml >= 15 else b"")
    return s


LIT_LENS = [0, 0, 0, 0, 1, 1, 2, 3, 5, 7, 13, 14, 15, 16, 17, 30, 63, 64, 65, 254, 255, 256, 268, 269, 270, 271, 272, 300, 524, 525, 1000, 4095, 4096, 9000]
MATCH_LENS = [4, 4, 4, 5, 5, 6, 7, 8, 9, 12, 15, 16, 17, 18, 19, 20, 21, 33, 34, 35, 64, 100, 255, 272, 273, 274, 275, 528, 529, 1000, 1023, 1024, 1025, 5000, 70000]
OFFSETS = [1, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 1023, 1024, 3327, 3328, 3329, 4095, 4096, 6143, 6144, 7168, 8192,
           16383, 32767, 32768, 32769, 65534, 65535]


def random_lz4_block(rnd, target):
    plain, blk = bytearray(), bytearray()
    style = rnd.randrange(4)   # 0: anything; 1: short everything (many records per byte); 2: long literal runs; 3: long matches
    while len(plain) < target:
        if style == 1:
            ll, ml = rnd.choice([0, 0, 1, 2, 3]), rnd.choice([4, 4, 5, 6, 8])
        elif style == 2:
            ll, ml = rnd.choice([14, 15, 16, 269, 270, 300, 4096, 9000, 20000]), rnd.choice(MATCH_LENS[:20])
        elif style == 3:
            ll, ml = rnd.choice(LIT_LENS[:12]), rnd.choice([273, 274, 1000, 5000, 70000, 200000])
        else:
            ll, ml = rnd.choice(LIT_LENS), rnd.choice(MATCH_LENS)
        if rnd.randrange(8) == 0:
            ll, ml = rnd.randrange(0, 600), rnd.randrange(4, 600)
        if not plain and ll == 0:
            ll = 1
        lits = rnd.randbytes(ll)
        plain += lits
        have = len(plain)
        pick = rnd.randrange(6)
        if pick == 0:
            off = min(have, 65535)                       # reaches exactly to the start of the output / the full window
        elif pick == 1:
            off = rnd.randrange(1, min(have, 65535) + 1)
        else:
            off = min(rnd.choice(OFFSETS), have)
        blk += sequence(lits, off, ml)
        start = have - off
        if off >= ml:
            plain += plain[start:start + ml]
        else:
            pat = bytes(plain[start:])
            plain += (pat * (ml // off + 1))[:ml]
    # (the reference wants five literals in the last sequence and twelve bytes behind the start of the last match, LZ4.swift:370-372:
    # one block in eight ends in a way it refuses, and the engine must refuse it with the same status)
    tail = rnd.randbytes(rnd.choice([0, 1, 4, 5, 11]) if rnd.randrange(8) == 0 else rnd.choice([12, 13, 15, 16, 300]))
    blk += sequence(tail)
    plain += tail
    return bytes(blk), bytes(plain)


def damage(rnd, z):
    b = bytearray(z)
    m = rnd.randrange(4)
    if m == 0 and len(b) > 1:
        b = b[:rnd.randrange(1, len(b))]
    elif m == 1 and b:
        for _ in range(rnd.randrange(1, 4)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif m == 2 and b:
        b[rnd.randrange(min(len(b), 64))] ^= 1 << rnd.randrange(8)
    else:
        b += rnd.randbytes(rnd.randrange(1, 9))
    return bytes(b)


def spliced_plain(rnd, n, seed):
    """Text for the Deflate encoder: stretches of the payload classes, random bytes, and copies of earlier stretches from every
    distance up to (and beyond) the 32 KiB window."""
    gens = [corpus.p_text, corpus.p_rep, corpus.p_mix, corpus.p_rand, lambda k, s: corpus.p_zero(k)]
    out = bytearray()
    while len(out) < n:
        k = rnd.choice([1, 3, 8, 40, 258, 259, 1000, 5000, 40000])
        if out and rnd.randrange(3) == 0:
            d = min(len(out), rnd.choice([1, 2, 3, 4, 8, 64, 3328, 3329, 6144, 8192, 32767, 32768, 32769, 50000]))
            start = len(out) - d
            out += (bytes(out[start:]) * (k // d + 1))[:k]
        else:
            out += gens[rnd.randrange(len(gens))](k, seed + len(out))
    return bytes(out[:n])



STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def random_deflate_stream(rnd, n, seed):
    """zlib with random level / window / memory level / strategy over spliced text, flushed somewhere in the middle."""
    p = spliced_plain(rnd, n, seed)
    co = zlib.compressobj(rnd.choice([0, 1, 2, 3, 4, 6, 9]), zlib.DEFLATED, -rnd.choice([9, 10, 12, 15]), rnd.choice([1, 4, 8, 9]), rnd.choice(STRATEGIES))
    cut = rnd.randrange(len(p) + 1)
    return co.compress(p[:cut]) + co.flush(rnd.choice([zlib.Z_NO_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])) + co.compress(p[cut:]) + co.flush()


def random_lzma2_unit(rnd, n, seed):
    """(raw LZMA2 stream, dictionary byte): liblzma with random lc / lp / pb, dictionary size, mode, match finder and nice length over
    spliced text -- small dictionaries make the encoder restart its state and dictionary between chunks."""
    import lzma
    p = spliced_plain(rnd, n, seed)
    lc = rnd.randrange(0, 5)
    lp = rnd.randrange(0, 5 - lc)
    dict_size = rnd.choice([4096, 4096, 65536, 1 << 20, 1 << 23])
    f = {"id": lzma.FILTER_LZMA2, "lc": lc, "lp": lp, "pb": rnd.randrange(0, 5), "dict_size": dict_size,
         "mode": rnd.choice([lzma.MODE_FAST, lzma.MODE_NORMAL]), "nice_len": rnd.choice([5, 8, 32, 64, 273]),
         "mf": rnd.choice([lzma.MF_HC3, lzma.MF_HC4, lzma.MF_BT2, lzma.MF_BT4]), "depth": rnd.choice([0, 1, 4])}
    if f["mode"] == lzma.MODE_FAST and f["mf"] in (lzma.MF_BT2, lzma.MF_BT4) and f["nice_len"] < 8:
        f["nice_len"] = 8
    return lzma.compress(p, format=lzma.FORMAT_RAW, filters=[f]), corpus.lzma2_dict_byte(dict_size)
