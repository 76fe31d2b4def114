"""CPU tier: the per-lane LZ4 block decoder (swcompression_amd/csrc/lz4_lane.h) built for the host vs
the oracle's restatement of LZ4.process(block:_:) (reference Sources/LZ4/LZ4.swift:332-413)."""
import random

import pytest

import _emu as E
import _oracle as O
import _streams as S


@pytest.fixture(autouse=True, params=[2, 1], ids=["derived-offsets", "eight-byte-records"])
def record_mode(request):
    """Every test runs with both forms in which the parse tells the copier where a record's literals lie in the block
    (kernels.hip SWC_LZ4_RECORD_MODE: 2 ships -- four-byte records, offsets derived by a running sum, anchors where the rule
    breaks; 1 -- eight-byte records with the offset in the upper dword)."""
    E.lib.emu_set_lz4_record_mode(request.param)
    yield request.param
    E.lib.emu_set_lz4_record_mode(2)


def test_valid_blocks_and_dictionaries():
    cases = S.lz4_blocks_valid()
    exp = [O.lz4_block(z, d) for z, d in cases]
    assert all(e[0] == 0 for e in exp)
    res = E.lz4_block([z for z, _ in cases], [len(e[1]) for e in exp], [d for _, d in cases])
    for r, e, (z, _) in zip(res, exp, cases):
        assert r[:2] == e and r[2] == len(z)


def test_fuzz_status_and_bytes():
    O.lib.refcpu_set_max_output(1 << 22)
    cases = S.lz4_blocks_fuzz()
    exp = [O.lz4_block(z, d) for z, d in cases]
    rnd = random.Random(1)
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    res = E.lz4_block([cases[i][0] for i in keep], [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 3, 64]) for i in keep],
                      [cases[i][1] for i in keep])
    for r, i in zip(res, keep):
        assert r[0] == exp[i][0], cases[i][0].hex()
        if exp[i][0] == 0:
            assert r[1] == exp[i][1]
    O.lib.refcpu_set_max_output(1 << 30)


def test_capacity_reports_required_size():
    cases = [c for c in S.lz4_blocks_valid() if c[1] is None]
    exp = [O.lz4_block(z) for z, _ in cases]
    res = E.lz4_block([z for z, _ in cases], [len(e[1]) // 2 for e in exp])
    for r, e in zip(res, exp):
        assert r[0] == 901 and r[3] == len(e[1]) and r[1] == e[1][:len(e[1]) // 2]


import pytest


@pytest.mark.parametrize("order", [0, 1, 2])
def test_large_blocks_sub_chunk_rounds(order):
    """Blocks large enough for the sub-chunk-parallel rounds of lz4_wave.h (64 lanes x 256 bytes per round): every payload
    class, then damaged copies (byte flips -- offset 0, offsets beyond the output, broken length extensions -- and
    truncations) whose status and output must still be the oracle's.  The thread order of the emulated SIMT regions must
    not matter."""
    from swcompression_amd import corpus
    O.lib.refcpu_set_max_output(1 << 23)
    rnd = random.Random(5 + order)
    cases = []
    for kind, size in (("text", 300000), ("mix", 400000), ("rep", 200000), ("zero", 100000), ("rand", 70000), ("text", 1 << 20)):
        z = corpus.lz4_block(corpus.PAYLOADS[kind](size, 31))
        cases.append(z)
        if kind in ("text", "mix") and size <= 400000:
            for _ in range(6):
                b = bytearray(z)
                for _ in range(rnd.choice([1, 1, 3])):
                    b[rnd.randrange(len(b))] = rnd.choice([0, 0, 255, rnd.randrange(256)])
                cases.append(bytes(b))
            cases.append(z[:rnd.randrange(len(z) // 2, len(z))])
    exp = [O.lz4_block(z) for z in cases]
    E.set_order(order)
    try:
        res = E.lz4_block(cases, [max(len(e[1]), 1) + 64 for e in exp], misalign=order)
    finally:
        E.set_order(0)
    for i, (r, e) in enumerate(zip(res, exp)):
        if e[0] == 901:
            continue
        assert r[0] == e[0], "case %d" % i
        if e[0] == 0:
            assert r[1] == e[1] and r[2] == len(cases[i]), "case %d" % i
    O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("order", [0, 2])
def test_matches_older_than_the_ring_history(order):
    """The resolve kernel keeps 32 KiB of history in LDS; LZ4 offsets reach 65,535 bytes back.  Match bytes whose source is
    older than the ring's history take the far path of lz_resolve.h (the output buffer): some of them among near ones
    (the far list), whole batches of nothing else (the list overflows: read in place), sources that straddle the
    32 KiB line, and far matches that overlap a span's own output."""
    from swcompression_amd import corpus
    rnd = random.Random(77)
    a = corpus.p_rand(40000, 1)
    b = corpus.p_rand(21000, 2)
    t = corpus.p_text(50000, 3)
    payloads = [
        a + b + a,                                   # one giant match 61,000 back: 40,000 far bytes in a row
        a + t + a[:500] + t[:30000] + a[10000:10400],    # far matches among near ones
        a[:33000] + a[:33000] + a[:33000],           # distance 33,000: just past the history, back to back
        t + a[:32700] + t[:5000] + a[100:9000],      # sources on both sides of the 32 KiB line
    ]
    for k in range(6):                               # random mixtures of near and far copies
        parts = [corpus.p_rand(rnd.randrange(1000, 50000), 10 + k)]
        for _ in range(12):
            src = b"".join(parts)
            if rnd.random() < 0.5 and len(src) > 200:
                back = rnd.randrange(100, min(len(src), 65000))
                ln = rnd.randrange(4, min(back, 3000) + 1)
                parts.append(src[len(src) - back:len(src) - back + ln])
            else:
                parts.append(corpus.p_text(rnd.randrange(10, 4000), 100 + k))
        payloads.append(b"".join(parts))
    blocks = [corpus.lz4_block(p) for p in payloads]
    exp = [O.lz4_block(z) for z in blocks]
    assert all(e[:2] == (0, p) for e, p in zip(exp, payloads))
    E.set_order(order)
    try:
        res = E.lz4_block(blocks, [len(p) for p in payloads], misalign=order)
    finally:
        E.set_order(0)
    for i, (r, p) in enumerate(zip(res, payloads)):
        assert r[0] == 0 and r[1] == p and r[2] == len(blocks[i]), "payload %d" % i


@pytest.mark.parametrize("misalign", range(16))
def test_long_literal_runs_at_every_output_alignment(misalign):
    """Runs of plain literals (incompressible data: literal-only records of 2,048 bytes, stored Deflate blocks) are copied
    by the resolve kernel, not expanded: a run longer than the 64 KiB ring wraps around it, runs begin and end at any byte
    of the output, and what follows them picks up the ring and the bytes of an incomplete dword."""
    from swcompression_amd import corpus
    pays = [corpus.p_rand(70000, 31), corpus.p_text(3001, 1) + corpus.p_rand(66000, 2) + corpus.p_text(5003, 3) + corpus.p_rand(1500, 4) + corpus.p_text(777, 5),
            corpus.p_mix(150000, 6)]
    blocks = [corpus.lz4_block(p) for p in pays]
    for r, p, z in zip(E.lz4_block(blocks, [len(p) for p in pays], misalign=misalign), pays, blocks):
        assert r[0] == 0 and r[1] == p and r[2] == len(z)
    streams = [corpus.deflate_raw(p) for p in pays]
    for r, p in zip(E.inflate(streams, [len(p) for p in pays], misalign=misalign), pays):
        assert r[0] == 0 and r[1] == p
