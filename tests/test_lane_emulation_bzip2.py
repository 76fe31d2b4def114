"""CPU tier: the three BZip2 device stages (swcompression_amd/csrc/bzip2_block.h) built for the host vs the
oracle (reference Sources/BZip2/BZip2.swift, BurrowsWheeler.swift).  The stages decode ONE block; stream
walking lives in the C++ host framing and is covered by the GPU tier through the C ABI, so here single-block
streams are used and only block-level outcomes are compared."""
import random

import pytest

import _emu as E
import _oracle as O
import _streams as S

BLOCK_LEVEL = {205, 206, 207, 208, 209, 900}


@pytest.fixture(autouse=True, params=["fused", "team", "team-stealing"])
def stage3(request):
    """Every test with stage 3a inside the block's wavefront (bzip2_block.h), as kernels of its own (bzip2_team.h: every team's
    tickets drawn by that team's thread), and with ONE thread that starts at team 5 and goes through all teams' tickets."""
    E.set_bzip2_team({"fused": None, "team": (0, 0), "team-stealing": (5, 1)}[request.param])
    yield
    E.set_bzip2_team(None)


def _single_block(stream):
    return stream[4:10] == bytes.fromhex("314159265359")


def test_valid_single_block_streams():
    cases = [(z, x) for z, x in S.bzip2_valid() if _single_block(z) and z.count(bytes.fromhex("314159265359")) == 1 and len(x) <= 100000 or len(x) == 300000 and z[3:4] == b"9"]
    assert len(cases) >= 15
    res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                        [max(len(x), 1) for _, x in cases])
    for r, (z, x) in zip(res, cases):
        assert r[:2] == (0, x)
        assert O.bzip2(z)[:2] == (0, x)


def test_the_team_path_finishes_valid_blocks_itself():
    """(Otherwise the serial fallback of stage 3b would hide a team path that never completes.)"""
    import ctypes as C
    if E._bzip2_team is None:
        return
    E.lib.emu_bzip2_team_finished.restype = C.c_uint64
    E.lib.emu_bzip2_team_finished(1)
    cases = [(z, x) for z, x in S.bzip2_valid() if _single_block(z) and z.count(bytes.fromhex("314159265359")) == 1 and 0 < len(x) <= 100000][:40]
    res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                        [max(len(x), 1) for _, x in cases])
    assert all(r[:2] == (0, x) for r, (_, x) in zip(res, cases))
    assert E.lib.emu_bzip2_team_finished(1) >= len(cases) - 2     # (a periodic payload's permutation has several cycles: the serial walk's, in both forms)


def test_body_fuzz_block_level_outcomes():
    O.lib.refcpu_set_max_output(1 << 24)
    rnd = random.Random(23)
    import bz2
    from swcompression_amd import corpus
    base = [bz2.compress(corpus.PAYLOADS[k](n, 3), 9) for k in ("text", "rep", "mix", "rand", "zero") for n in (40, 700, 6000, 40000)]
    ins = []
    for z in base:
        for _ in range(30):
            b = bytearray(z)
            for _ in range(rnd.randrange(1, 3)):
                b[rnd.randrange(14, len(b) - 10)] ^= 1 << rnd.randrange(8)
            ins.append(bytes(b))
        for _ in range(10):
            ins.append(z[:rnd.randrange(15, len(z))])
    exp = [O.bzip2(z) for z in ins]
    res = E.bzip2_block(ins, [112] * len(ins), [int.from_bytes(z[10:14], "big") for z in ins], [1 << 20] * len(ins), lcap=1 << 20)
    checked = 0
    for r, e, z in zip(res, exp, ins):
        if e[0] in BLOCK_LEVEL:
            assert r[0] == e[0], z[:24].hex()
            checked += 1
        elif e[0] == 0:
            assert r[:2] == (0, e[1])
            checked += 1
        elif e[0] == 210 and r[0] == 210:
            assert r[1] == e[1]
            checked += 1
    assert checked > len(ins) // 2
    O.lib.refcpu_set_max_output(1 << 30)


def test_capacity_reports_required_size():
    cases = [(z, x) for z, x in S.bzip2_valid() if _single_block(z) and 5000 <= len(x) <= 100000][:6]
    res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                        [len(x) // 2 for _, x in cases])
    for r, (z, x) in zip(res, cases):
        assert r[0] == 901 and r[3] == len(x) and r[1] == x[:len(x) // 2]


def test_block_sizes_around_the_segment_geometry():
    """Stage 3a cuts the BWT cycle at multiples of M = 2^m (n / M <= 512) and keeps 3 M bytes per segment: sizes around
    every change of m, tiny blocks (M = 1) and payloads whose RLE1 form differs a lot from the output."""
    import bz2
    from swcompression_amd import corpus
    sizes = list(range(1, 40)) + [255, 256, 257, 511, 512, 513, 514, 600, 1023, 1024, 1025, 1030, 2047, 2048, 2049, 2055, 4095, 4096,
                                  4097, 8191, 8192, 8193, 8200, 20000, 65535, 65536, 65537]
    cases = []
    for i, n in enumerate(sizes):
        for kind in ("text", "rand", "rep"):
            x = corpus.PAYLOADS[kind](n, 100 + i)
            cases.append((bz2.compress(x, 9), x))
    res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                        [max(len(x), 1) for _, x in cases], lcap=70000)
    for r, (z, x) in zip(res, cases):
        assert r[:2] == (0, x), len(x)


def test_unchecked_final_code_length():
    """BZip2.swift:185 checks a code length BEFORE the deltas of a symbol, so the last symbol's (EOB's) length can leave
    0...20.  The reference builds its tree for whatever maxBits results (DecodingTree.swift:19); oracle and engine follow
    it up to 26 bits and classify longer ones as trap-class."""
    d = b"banana bandana cabana " * 3
    cases = []
    for eob_len in (20, 21, 22, 25, 26, 27, 40, 0, -3):
        cases.append((S.bzip2_crafted(d, [1, 2, 3, 4, 5, 6, 7, eob_len]), eob_len))
    cases.append((S.bzip2_crafted(d, [2, 2, 3, 3, 3, 4, 5, 5]), 5))
    cases.append((S.bzip2_crafted(d, [2, 2, 3, 3, 3, 4, 4, 23]), 23))     # 4-bit codes 1110 and 1111, then EOB = 23 bits past 1111: over-subscribed
    ins = [z for z, _ in cases]
    exp = [O.bzip2(z) for z in ins]
    res = E.bzip2_block(ins, [112] * len(ins), [int.from_bytes(z[10:14], "big") for z in ins], [len(d)] * len(ins))
    for (z, eob_len), e, r in zip(cases, exp, res):
        if 1 <= eob_len <= 26 and eob_len != 23:
            assert e[:2] == (0, d), eob_len
        if eob_len > 26:
            assert e[0] == 900
        assert r[0] == e[0], (eob_len, r[0], e[0])
        if e[0] == 0:
            assert r[1] == e[1]


def test_lane_order_does_not_matter():
    """The stage-1 symbol loop and the stage-2 counting sort are written as SIMT regions over 64 lanes (csrc/simt.h): the host
    build runs a region's lanes in forward, reverse or shuffled order, and a region that depends on the order of its lanes
    would show up as a difference here."""
    import bz2
    from swcompression_amd import corpus
    cases = [(bz2.compress(corpus.PAYLOADS[k](n, 7), 9), corpus.PAYLOADS[k](n, 7)) for k in ("text", "mix", "zero", "rand") for n in (1, 300, 70000)]
    try:
        for order in (0, 1, 2):
            E.set_order(order)
            res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                                [max(len(x), 1) for _, x in cases])
            for r, (z, x) in zip(res, cases):
                assert r[:2] == (0, x), order
    finally:
        E.set_order(0)


def test_symbol_loop_boundaries_emulated():
    """The CPU twin of tests/test_gpu_bzip2.py::test_symbol_loop_boundaries: the same kinds of input through the host build
    (hot_symbols_cxx, the C++ form of the assembly loop), at exact capacities."""
    import bz2, random
    rnd = random.Random(77)
    cases = []
    for trial in range(60):
        alpha = rnd.choice([1, 2, 3, 17, 63, 64, 65, 130, 256])
        syms = bytes(rnd.sample(range(256), alpha))
        parts = []
        total = rnd.choice([1, 49, 50, 51, 99, 500, 5000])
        while sum(len(p) for p in parts) < total:
            k = rnd.random()
            if k < 0.4:
                parts.append(bytes([rnd.choice(syms)]) * rnd.choice([1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 49, 50, 51, 63, 64, 65, 200, 3000]))
            elif k < 0.7:
                w = bytes(rnd.choice(syms) for _ in range(rnd.randrange(1, 9)))
                parts.append(w * rnd.randrange(1, 60))
            else:
                parts.append(bytes(rnd.choice(syms) for _ in range(rnd.randrange(1, 300))))
        x = b"".join(parts)
        z = bz2.compress(x, 9)
        if z.count(bytes.fromhex("314159265359")) == 1:
            cases.append((z, x))
    res = E.bzip2_block([z for z, _ in cases], [112] * len(cases), [int.from_bytes(z[10:14], "big") for z, _ in cases],
                        [len(x) for _, x in cases])
    for r, (z, x) in zip(res, cases):
        assert r[:2] == (0, x)


def test_workspace_geometry_holds_for_every_block_size():
    """A workspace is cut for a capacity; the blocks decoded into it are of any size up to that.  The segment arrays and buffers
    of both forms of stage 3 must fit whatever the block's own cut is (the number of segments is not monotonic in the size)."""
    import ctypes as C
    E.lib.emu_bzip2_layout.argtypes = [C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64)]
    rnd = random.Random(3)
    caps = [16, 17, 31, 32, 33, 100, 511, 512, 513, 4096, 16384, 16385, 65535, 65536, 524288, 524289, 900000, 1 << 20, (1 << 20) + 1, 16000000]
    caps += [rnd.randrange(16, 16000000) for _ in range(200)]
    prev = 0
    for lcap in sorted(caps):
        out = (C.c_uint64 * 8)()
        for n in {k for k in (1, 2, 31, 32, 33) if k <= lcap} | {max(lcap // 2, 1), max(lcap - 1, 1), lcap} | {rnd.randrange(1, lcap + 1) for _ in range(40)}:
            E.lib.emu_bzip2_layout(lcap, n, out)
            slots, segs, bufs, need_team, need_fused, info, info_need, ws = list(out)
            assert segs + 1 <= slots, (lcap, n, segs, slots)
            assert need_team + 64 <= bufs and need_fused + 64 <= bufs, (lcap, n, need_team, need_fused, bufs)
            assert info_need <= info
        assert ws >= prev                      # launch_bzip2 searches the capacity a workspace holds by bisection
        prev = ws
