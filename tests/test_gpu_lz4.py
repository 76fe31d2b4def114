"""GPU tier: LZ4 through the C ABI on the MI355X vs the oracle (reference Sources/LZ4/LZ4.swift)."""
import random
import struct

import numpy as np
import pytest

import _oracle as O
import _streams as S
import swcompression_amd as swc
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "wave"])
def lz_copy_kernel(request):
    """Phase 2 has two kernels that must write the same bytes: launches of fewer than 2,560 streams take the workgroup kernel
    (lz_resolve.h, latency), larger ones the wave kernel (lz_copy.h, throughput).  Most tests here launch few streams, so every
    test runs twice: with the library's own choice, and with the wave kernel forced (swc_set_tuning("lz_copier", -1))."""
    from swcompression_amd import _lib
    lib = _lib.load()
    assert lib.swc_set_tuning(b"lz_copier", -1 if request.param == "wave" else 1) == 0
    yield request.param
    lib.swc_set_tuning(b"lz_copier", 1)


def test_batch_blocks_vs_oracle():
    O.lib.refcpu_set_max_output(1 << 22)
    cases = [c for c in S.lz4_blocks_valid() + S.lz4_blocks_fuzz() if c[1] is None]
    exp = [O.lz4_block(z) for z, _ in cases]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    rnd = random.Random(1)
    caps = [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 3, 64]) for i in keep]
    b = DeviceBatch("lz4_block", [cases[i][0] for i in keep], caps)
    b.launch(sync=True)
    r = b.results()
    for k, i in enumerate(keep):
        assert int(r["status"][k]) == exp[i][0], cases[i][0].hex()
        if exp[i][0] == 0:
            assert int(r["out_len"][k]) == len(exp[i][1]) and b.output(k, len(exp[i][1])) == exp[i][1]
    O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("case", S.lz4_frames(), ids=lambda c: c[0])
def test_frames(case):
    name, frame, d, did = case
    st, out, cons = O.lz4(frame, d, -1 if did is None else did)
    assert st == 0
    assert swc.LZ4.decompress(frame, d, did) == out


def test_multi_frame_and_error_taxonomy():
    x = corpus.p_text(5000, 4)
    f1, f2 = corpus.lz4f_frame(x[:700], 4, False, True), corpus.lz4f_frame(x[700:], 4, True, True)
    skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
    assert swc.LZ4.multi_decompress(f1 + skip + f2) == [x[:700], x[700:]]
    assert swc.LZ4.decompress(f1 + f2) == x[:700]                       # only the first frame (LZ4.swift:41)
    bad = bytearray(f1); bad[-1] ^= 1
    with pytest.raises(swc.DataError) as ei:
        swc.LZ4.decompress(bytes(bad))
    assert ei.value.case == "checksumMismatch" and ei.value.data == x[:700]   # LZ4Tests.swift:186-203
    for data, case in ((b"", "truncated"), (b"\x00", "truncated"), (bytes(1 << 16), "corrupted")):
        with pytest.raises(swc.DataError) as ei:
            swc.LZ4.decompress(data)
        assert ei.value.case == case                                     # LZ4Tests.swift:87-108
    rnd = random.Random(9)
    for _ in range(16):                                                  # truncation fuzz, LZ4Tests.swift:205-214
        cut = f2[:rnd.randrange(1, len(f2))]
        st = O.lz4(cut)[0]
        assert st != 0
        with pytest.raises(swc.SWCError) as ei:
            swc.LZ4.decompress(cut)
        assert ei.value.status == st
    for _ in range(16):                                                  # bit-flip fuzz: same status, same bytes
        b2 = bytearray(f2); b2[rnd.randrange(7, len(b2))] ^= 1 << rnd.randrange(8)
        st, out, _ = O.lz4(bytes(b2))
        if st == 0:
            assert swc.LZ4.decompress(bytes(b2)) == out
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.LZ4.decompress(bytes(b2))
            assert ei.value.status == st


def test_dictionary_frames():
    d = corpus.p_text(70000, 8)
    blk = bytes([0x0F, 0x03, 0x00, 0x02, 0x80]) + b"stuvwxyz"
    desc = bytes([0x60, 0x40])
    frame = struct.pack("<I", 0x184D2204) + desc + bytes([(O.xxh32(desc) >> 8) & 0xFF]) + struct.pack("<I", len(blk)) + blk + struct.pack("<I", 0)
    for dictionary in (b"abc", d, None):
        st, out, _ = O.lz4(frame, dictionary)
        if st == 0:
            assert swc.LZ4.decompress(frame, dictionary) == out
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.LZ4.decompress(frame, dictionary)
            assert ei.value.status == st


def test_many_64k_blocks_bit_exact():
    units, plains = corpus.build_units("lz4_block", 2048, 65536, payload="mix")
    b = DeviceBatch("lz4_block", units, [65536] * len(units))
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == 65536).all()
    blob = b.d_out.cpu().numpy()
    for i in range(len(units)):
        o = int(b._out_off[i])
        assert blob[o:o + 65536].tobytes() == plains[i]


def test_large_blocks_two_phase_path():
    """Blocks far larger than the parse stripes and the resolve window: 4 MiB text (BASELINE config 3 shape), long
    matches (zeros / repeated phrase: records are split at 258 bytes and chain), incompressible data (one literal run of
    hundreds of KiB: skip records), and a capacity that is too small."""
    plains = [corpus.p_text(4 << 20, 11), corpus.p_zero(1 << 20), corpus.p_rand(300000, 5), corpus.p_rep(1 << 20, 3),
              corpus.p_mix(2 << 20, 9), corpus.p_text(70000, 1), b"a" * 20, corpus.p_text(5 << 20, 2)[:3000000]]
    blocks = [corpus.lz4_block(p) for p in plains]
    caps = [len(p) for p in plains]
    caps[5] = 50000   # too small: SWC_E_CAPACITY with the required size
    b = DeviceBatch("lz4_block", blocks, caps)
    b.launch(sync=True)
    r = b.results()
    O.lib.refcpu_set_max_output(1 << 23)
    for i, p in enumerate(plains):
        st, out = O.lz4_block(blocks[i])[:2]
        assert st == 0 and out == p
        if i == 5:
            assert int(r["status"][i]) == 901 and int(r["out_len"][i]) == len(p)
            assert b.output(i, 50000) == p[:50000]
            continue
        assert int(r["status"][i]) == 0 and int(r["out_len"][i]) == len(p) and int(r["in_consumed"][i]) == len(blocks[i]), i
        assert b.output(i, len(p)) == p, "bytes differ on block %d" % i
    O.lib.refcpu_set_max_output(1 << 30)


def test_randomised_blocks_two_phase_batch():
    """Random sizes x payload classes in one batch through the stripe parser and the 64 KiB-history resolve kernel."""
    rnd = random.Random(4242)
    gens = [corpus.p_text, corpus.p_rep, corpus.p_mix, corpus.p_rand, lambda n, s: corpus.p_zero(n)]
    plains = []
    for i in range(160):
        n = rnd.choice([1, 5, 12, 13, 64, 65, 300, 4096, 65535, 65536, 70001, 200000, 1 << 20])
        plains.append(gens[i % len(gens)](n, 500 + i))
    blocks = [corpus.lz4_block(p) for p in plains]
    b = DeviceBatch("lz4_block", blocks, [len(p) for p in plains])
    b.launch(sync=True)
    r = b.results()
    O.lib.refcpu_set_max_output(1 << 23)
    for i, p in enumerate(plains):
        assert int(r["status"][i]) == 0 and int(r["out_len"][i]) == len(p) and int(r["in_consumed"][i]) == len(blocks[i]), i
        assert b.output(i, len(p)) == p, "bytes differ on block %d (len %d)" % (i, len(p))
        if i % 13 == 0:
            assert O.lz4_block(blocks[i])[:2] == (0, p)
    O.lib.refcpu_set_max_output(1 << 30)


def _lz4_sequence(lits, offset=None, mlen=None):
    """One LZ4 sequence (LZ4.swift:341-412): token, literal-length extension, literals, offset, match-length extension."""
    def ext(v):
        out = bytearray()
        while v >= 255:
            out.append(255); v -= 255
        out.append(v)
        return bytes(out)
    ll = len(lits)
    ml = 0 if mlen is None else mlen - 4
    tok = (min(ll, 15) << 4) | min(ml, 15)
    s = bytes([tok]) + (ext(ll - 15) if ll >= 15 else b"") + lits
    if mlen is not None:
        s += struct.pack("<H", offset) + (ext(ml - 15) if ml >= 15 else b"")
    return s


def test_far_matches_right_behind_a_long_literal_run():
    """the resolve kernel keeps 32 KiB of history in LDS and reads older match sources back from the output
    buffer.  A literal run of more than 32 KiB is stored by the copy path in one step; matches with offsets 32769..65535
    that follow at once read bytes of that run from HBM -- they must see them (fence + workgroup barrier after the run)."""
    rnd = random.Random(77)
    blocks, plains = [], []
    for L in (32769, 40000, 65535, 70000, 200000, 1 << 20):
        for rep in range(4):
            lits = corpus.p_rand(L, 900 + rep)
            plain = bytearray(lits)
            blk = bytearray()
            first = True
            for _ in range(rnd.choice([1, 3, 40, 400])):
                off = rnd.randrange(32769, min(65535, len(plain)) + 1)
                ml = rnd.choice([4, 5, 8, 19, 64, 300, 2000])
                blk += _lz4_sequence(bytes(lits) if first else b"", off, ml)
                first = False
                for _k in range(ml):
                    plain.append(plain[len(plain) - off])
            tail = corpus.p_rand(12, 5)
            blk += _lz4_sequence(tail)
            plain += tail
            blocks.append(bytes(blk)); plains.append(bytes(plain))
    O.lib.refcpu_set_max_output(1 << 23)
    for z, p in zip(blocks[::5], plains[::5]):
        assert O.lz4_block(z)[:2] == (0, p)
    O.lib.refcpu_set_max_output(1 << 30)
    # many copies at once so that both workgroups of a CU and several rounds of the grid are in flight
    reps = 8
    b = DeviceBatch("lz4_block", blocks * reps, [len(p) for p in plains] * reps)
    b.launch(sync=True)
    r = b.results()
    for i in range(len(blocks) * reps):
        p = plains[i % len(blocks)]
        assert int(r["status"][i]) == 0 and int(r["out_len"][i]) == len(p), i
        assert b.output(i, len(p)) == p, "bytes differ on block %d" % i
