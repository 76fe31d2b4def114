"""CPU tier: phase 2 of the Deflate / LZ4 paths on its own -- swcompression_amd/csrc/lz_copy.h, the record-granular copier the
gfx950 kernel swc_lz_copy_kernel is compiled from -- on record lists made HERE, so that every class of record the copier
treats differently is hit on purpose and in bulk: matches that end in front of their group, matches that reach into it,
matches in front of the LDS window (read back from the output buffer), overlapping matches (distance < length) down to
distance 1, matches of 1 / 2 bytes (tails of split matches) and of up to 511, literal runs of every class (0, 1-3, 4-8,
9-16, 17-32, 33-127 in front of a match -- the classes of both shipped configurations; literal-only records up to 2,048), capacity cuts, every output alignment, every order of
the emulated lanes.  The expected bytes come from replaying the records one by one in Python (what the reference's
`out.append` loops do, Deflate.swift:216-232, LZ4.swift:398-410); the byte-cell resolver of rounds 2-4 (lz_resolve.h) runs on the
same lists as a second witness."""
import ctypes as C
import random

import pytest

import _emu as E

lib = E.lib
lib.emu_copy_records.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
lib.emu_copy_records.restype = None


def make_match(lit, length, dist):
    return lit | (length << 7) | ((dist - 1) << 16)


def make_lits(n):
    return (n & 127) | ((n >> 7) << 16)


def build(rnd, nbytes, profile):
    """A random record list that covers about nbytes of output; returns (records, literal bytes, expected output)."""
    recs, lits, out = [], bytearray(), bytearray()
    while len(out) < nbytes:
        kind = rnd.choices(profile["kinds"], profile["weights"])[0]
        if kind == "lits" or len(out) == 0:
            n = rnd.choice(profile["litonly"])
            b = bytes(rnd.randrange(256) for _ in range(n))
            recs.append(make_lits(n)); lits += b; out += b
            continue
        lit = rnd.choice(profile["lit"])
        length = rnd.choice(profile["len"])
        dmax = min(len(out) + lit, 65536)
        dist = min(dmax, max(1, rnd.choice(profile["dist"])))
        if kind == "near":
            dist = min(dmax, rnd.randint(1, 40))
        elif kind == "far":
            dist = min(dmax, rnd.randint(9000, 65536))
        b = bytes(rnd.randrange(256) for _ in range(lit))
        lits += b; out += b
        for _ in range(length):
            out.append(out[-dist])
        recs.append(make_match(lit, length, dist))
    return recs, bytes(lits), bytes(out)


TEXT = dict(kinds=["match", "near", "far", "lits"], weights=[70, 10, 15, 5], lit=[0] * 14 + [1, 1, 2, 3, 4, 5, 8, 9, 20], len=[3, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16, 17, 24, 25, 32, 33, 40],
            dist=[50, 300, 700, 2000, 5000, 7000, 12000, 30000], litonly=[1, 2, 3, 5, 17, 100, 255, 256, 300])
RUNS = dict(kinds=["near", "match", "lits"], weights=[60, 30, 10], lit=[0, 0, 0, 1, 4, 64, 65, 127], len=[1, 2, 3, 4, 30, 64, 65, 100, 258, 300, 511],
            dist=[1, 2, 3, 4, 7, 255, 256, 257, 1000], litonly=[1, 255, 256, 1023, 1024, 2048])
DENSE = dict(kinds=["near"], weights=[1], lit=[0, 0, 1], len=[3, 4, 5, 8], dist=[1], litonly=[4])   # (every match reaches into its group)
# literal runs around every boundary of the eight-byte pieces a lane copies on its own (Deflate: two pieces, LZ4: four), and beyond
LITRUNS = dict(kinds=["match", "near", "far", "lits"], weights=[60, 15, 15, 10], lit=[0, 1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 24, 25, 31, 32, 33, 40, 63, 64, 65, 100, 127],
               len=[3, 4, 5, 8, 9, 16, 31, 32, 33], dist=[1, 5, 50, 700, 5000, 7000, 12000, 30000], litonly=[1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 100, 255, 256])
STORED = dict(kinds=["lits", "far", "match"], weights=[80, 10, 10], lit=[0, 3], len=[4, 40, 500], dist=[20000, 65536, 100], litonly=[2048, 2048, 2047, 1500, 256, 255])


def run_copy(recs, lits, cap, out_len, misalign, copier):
    buf = C.create_string_buffer(cap + 64 + 32)
    C.memset(buf, 0xA5, len(buf))
    o0 = (-C.addressof(buf)) % 16 + 16 + misalign
    arr = (C.c_uint32 * max(len(recs), 1))(*recs)
    lib.emu_copy_records(arr, len(recs), lits, len(lits), C.addressof(buf) + o0, cap, out_len, copier)
    raw = buf.raw
    assert raw[:o0] == b"\xA5" * o0 and raw[o0 + cap:] == b"\xA5" * (len(raw) - o0 - cap), "bytes outside the output were written"
    return raw[o0:o0 + min(cap, out_len)]


@pytest.mark.parametrize("profile", [TEXT, RUNS, DENSE, STORED, LITRUNS], ids=["text", "runs", "dense", "stored", "litruns"])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_record_classes(profile, order):
    rnd = random.Random(hash((profile["len"][0], order)) & 0xFFFF)
    E.set_order(order)
    try:
        for size in (0, 1, 63, 64, 200, 5000, 9000, 40000, 140000):
            recs, lits, exp = build(rnd, size, profile) if size else ([], b"", b"")
            for misalign in (0, 5, 15):
                for cfg in (1, 3):       # the two configurations the library ships (lz_copy.h: CfgLz4, CfgDeflate)
                    got = run_copy(recs, lits, max(len(exp), 1), len(exp), misalign, cfg)
                    assert got == exp, "copier differs (size %d, misalign %d, configuration %d)" % (size, misalign, cfg)
            if order == 0 and size and profile is not STORED:
                # (not on STORED: runs of literal-only records of unequal sizes followed by matches 64 KiB back are a stream no
                # phase 1 writes, and the old resolver's long-literal path -- not shipped any more -- does not take it)
                assert run_copy(recs, lits, len(exp), len(exp), 3, 0) == exp, "the byte-cell resolver differs"
    finally:
        E.set_order(0)


lib.emu_copy_records8.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
lib.emu_copy_records8.restype = None


def run_copy8(recs, offs, block, cap, out_len, misalign, copier):
    buf = C.create_string_buffer(cap + 64 + 32)
    C.memset(buf, 0xA5, len(buf))
    o0 = (-C.addressof(buf)) % 16 + 16 + misalign
    flat = [x for r, o in zip(recs, offs) for x in (r, o)]
    arr = (C.c_uint32 * max(len(flat), 1))(*flat)
    lib.emu_copy_records8(arr, len(recs), block, len(block), C.addressof(buf) + o0, cap, out_len, copier)
    raw = buf.raw
    assert raw[:o0] == b"\xA5" * o0 and raw[o0 + cap:] == b"\xA5" * (len(raw) - o0 - cap), "bytes outside the output were written"
    return raw[o0:o0 + min(cap, out_len)]


@pytest.mark.parametrize("profile", [TEXT, LITRUNS, STORED], ids=["text", "litruns", "stored"])
@pytest.mark.parametrize("order", [0, 2])
def test_records_that_point_at_their_literals(profile, order):
    """The LZ4 form of the records (lz4_wave.h R8, LZ4.swift:364-366: literals are byte-aligned in the block): eight bytes, the
    upper dword the offset of the literal run in the BLOCK, no dense literal stream.  The block here is the literal runs with one
    to four other bytes between them (tokens, offsets, length bytes), and the last run ends with the block: nothing may be read
    past it (the emulation copies the block into a buffer of exactly its size; tests/test_emulation_asan.py runs the LZ4 path
    of it under AddressSanitizer)."""
    rnd = random.Random(hash((profile["len"][-1], order)) & 0xFFFF)
    E.set_order(order)
    try:
        for size in (1, 63, 64, 200, 5000, 40000, 140000):
            recs, lits, exp = build(rnd, size, profile)
            block, offs, lp = bytearray(), [], 0
            for r in recs:
                ln = (r >> 7) & 511
                li = (r & 127) + (((r >> 16) << 7) if ln == 0 else 0)
                block += bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 4)))
                offs.append(len(block))
                block += lits[lp:lp + li]
                lp += li
            # (the last record's literals end the block when it has any; cut what was appended behind a record without literals)
            block = bytes(block[:max(o + ((r & 127) + (((r >> 16) << 7) if (r >> 7) & 511 == 0 else 0)) for r, o in zip(recs, offs))])
            for misalign in (0, 7):
                for cfg in (1, 3):
                    got = run_copy8(recs, offs, block, len(exp), len(exp), misalign, cfg)
                    assert got == exp, "copier differs (size %d, misalign %d, configuration %d)" % (size, misalign, cfg)
        # a block of fewer than eight bytes that still claims output (a stream that ends in an error): byte by byte
        recs = [make_match(3, 100, 2)]
        assert run_copy8(recs, [1], b"\x00abc", 103, 103, 0, 1) == b"abc" + b"bc" * 50
    finally:
        E.set_order(0)


lib.emu_copy_records4.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
lib.emu_copy_records4.restype = None


def _seq_bytes(li, le):
    return 3 + li + (1 if li >= 15 else 0) + (1 if le >= 19 else 0) if le else li


def _lit_skip(li, le):
    return 1 + (1 if li >= 15 else 0) if le else 0


@pytest.mark.parametrize("profile", [TEXT, LITRUNS, STORED, RUNS], ids=["text", "litruns", "stored", "runs"])
@pytest.mark.parametrize("order", [0, 2])
def test_records_whose_literal_offsets_are_derived(profile, order):
    """The shipped LZ4 form (lz4_wave.h / lz_copy.h record mode 2): four-byte records, the place of a record's literals in the
    block DERIVED -- the start S of its sequence is a running sum of seq_bytes(literals, length) over the records, the literals
    lie lit_skip() behind S -- and an ANCHOR (record index, S) wherever the rule would miss.  The block here is laid out by the
    rule for most records and OFF the rule (extra bytes in front) for a random tenth of them, which get anchors -- computed by
    the simulation the parse runs -- so groups are cut at anchors in the middle, at their first record, back to back; the last run
    ends with the block (nothing may be read past it: the emulation copies the block into a buffer of exactly its size)."""
    rnd = random.Random(hash((profile["len"][0], order, 4)) & 0xFFFF)
    E.set_order(order)
    try:
        for size in (1, 63, 64, 200, 5000, 40000, 140000):
            recs, lits, exp = build(rnd, size, profile)
            block, anchors, lp, S = bytearray(), [], 0, 0
            for i, r in enumerate(recs):
                le = (r >> 7) & 511
                li = (r & 127) + (((r >> 16) << 7) if le == 0 else 0)
                start = len(block)                                   # where this record's sequence begins in the block
                if rnd.random() < 0.1 or start > S:                  # off the rule (forced when the block has run ahead of the rule's sum)
                    block += bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 5) + max(0, S - start)))
                    start = len(block)
                else:
                    block += bytes(rnd.randrange(256) for _ in range(S - start))
                    start = S
                block += bytes(rnd.randrange(256) for _ in range(_lit_skip(li, le)))
                if li and S != start:
                    anchors += [i, start]
                    S = start
                block += lits[lp:lp + li]
                lp += li
                if le:                                               # the rest of the sequence: offset field, a length byte
                    block += bytes(rnd.randrange(256) for _ in range(2 + (1 if le >= 19 else 0)))
                S += _seq_bytes(li, le)
            # (a stream whose last record is literal-only ends with its literals, as an LZ4 block does)
            arr = (C.c_uint32 * len(recs))(*recs)
            anc = (C.c_uint32 * max(len(anchors), 1))(*anchors)
            for misalign in (0, 7):
                for cfg in (1, 3):
                    buf = C.create_string_buffer(len(exp) + 64 + 32)
                    C.memset(buf, 0xA5, len(buf))
                    o0 = (-C.addressof(buf)) % 16 + 16 + misalign
                    lib.emu_copy_records4(arr, len(recs), anc, len(anchors) // 2, bytes(block), len(block), C.addressof(buf) + o0, len(exp), len(exp), cfg)
                    raw = buf.raw
                    assert raw[:o0] == b"\xA5" * o0 and raw[o0 + len(exp):] == b"\xA5" * (len(raw) - o0 - len(exp))
                    assert raw[o0:o0 + len(exp)] == exp, "copier differs (size %d, misalign %d, configuration %d, %d anchors)" % (size, misalign, cfg, len(anchors) // 2)
    finally:
        E.set_order(0)


def test_capacity_cuts_the_last_records():
    """Records exist only for output below the capacity: the last one may reach past it (phase 2 clamps at the limit)."""
    rnd = random.Random(77)
    for profile in (TEXT, RUNS):
        recs, lits, exp = build(rnd, 30000, profile)
        for cut in (1, 2, 17, 300):
            cap = len(exp) - cut
            # drop the records that start at or behind the capacity
            keep, pos = [], 0
            for r in recs:
                if pos >= cap:
                    break
                keep.append(r)
                ln = (r >> 7) & 511
                lit = (r & 127) + (((r >> 16) << 7) if ln == 0 else 0)
                pos += lit + ln
            for cfg in (1, 3):
                assert run_copy(keep, lits, cap, len(exp), 7, cfg) == exp[:cap]
