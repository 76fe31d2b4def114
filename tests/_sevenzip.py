"""Folder descriptions for the SevenZipFolder.unpack tests: packed streams made with the system encoders, the coder chains a
7-Zip header would describe (the header itself stays on the caller's side of the boundary)."""
import bz2
import lzma
import struct
import zlib

from swcompression_amd import corpus


def delta_encode(data, distance):
    out = bytearray(len(data))
    for i, b in enumerate(data):
        out[i] = (b - (data[i - distance] if i >= distance else 0)) & 0xFF
    return bytes(out)


def lzma1_raw(payload, dict_size=1 << 20):
    """(properties[5], raw LZMA1 stream without end marker handling differences: liblzma writes one only when asked)."""
    raw = lzma.compress(payload, format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA1, "preset": 6, "dict_size": dict_size,
                                                                    "lc": 3, "lp": 0, "pb": 2}])
    props = bytes([(2 * 5 + 0) * 9 + 3]) + struct.pack("<I", dict_size)
    return props, raw


def folders(seed=0):
    """[(name, packed, chain, expected plain)]"""
    x = corpus.p_text(300000, seed + 1)
    y = corpus.p_mix(70000, seed + 2)
    z = corpus.p_rep(5000, seed + 3)
    out = []
    db = bytes([corpus.lzma2_dict_byte(1 << 20)])
    out.append(("copy", x[:1000], [("copy", None, 1000)], x[:1000]))
    out.append(("deflate", corpus.deflate_raw(x), [("deflate", None, len(x))], x))
    out.append(("bzip2", bz2.compress(y, 9), [("bzip2", None, len(y))], y))
    out.append(("bzip2-multiblock", bz2.compress(x, 1), [("bzip2", None, len(x))], x))
    out.append(("lzma2", corpus.lzma2_raw(x), [("lzma2", db, len(x))], x))
    props, raw = lzma1_raw(y)
    out.append(("lzma", raw, [("lzma", props, len(y))], y))
    out.append(("lz4", corpus.lz4_frame(y), [("lz4", None, len(y))], y))
    for dist in (1, 4, 255, 256):
        d = delta_encode(x[:50000], dist)
        out.append(("delta%d+lzma2" % dist, corpus.lzma2_raw(d), [("lzma2", db, len(d)), ("delta", bytes([(dist - 1) & 0xFF]), len(d))], x[:50000]))
    out.append(("copy+deflate", corpus.deflate_raw(z), [("copy", None, 7), ("deflate", None, len(z))], z))
    out.append(("empty-deflate", corpus.deflate_raw(b""), [("deflate", None, 0)], b""))
    return out


def damaged(seed=0):
    """[(name, packed, chain)] whose outcome is an error; compared with the oracle only."""
    x = corpus.p_text(40000, seed + 5)
    db = bytes([corpus.lzma2_dict_byte(1 << 20)])
    props, raw = lzma1_raw(x)
    d = corpus.deflate_raw(x)
    flip = bytearray(d)
    flip[len(flip) // 2] ^= 0x40
    return [
        ("wrong-size", d, [("deflate", None, len(x) + 1)]),
        ("wrong-size-after-delta", corpus.lzma2_raw(x), [("lzma2", db, len(x)), ("delta", b"\x00", len(x) - 1)]),
        ("multi-stream", d, [("deflate", None, len(x), True)]),
        ("encryption", d, [("encryption", None, len(x))]),
        ("unsupported", d, [("other", None, len(x))]),
        ("lzma2-no-props", corpus.lzma2_raw(x), [("lzma2", None, len(x))]),
        ("lzma2-two-props", corpus.lzma2_raw(x), [("lzma2", db + db, len(x))]),
        ("lzma-short-props", raw, [("lzma", props[:4], len(x))]),
        ("lzma-bad-props-byte", raw, [("lzma", bytes([230]) + props[1:], len(x))]),
        ("delta-no-props", x, [("delta", None, len(x))]),
        ("deflate-flipped-bit", bytes(flip), [("deflate", None, len(x))]),
        ("bzip2-truncated", bz2.compress(x)[:-9], [("bzip2", None, len(x))]),
        ("lz4-not-a-frame", x[:100], [("lz4", None, 100)]),
        ("lzma-wrong-declared-size", raw, [("lzma", props, len(x) - 10)]),
    ]
