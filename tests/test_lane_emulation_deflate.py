"""CPU tier: the Deflate decoder the gfx950 kernels are compiled from (swcompression_amd/csrc/inflate_sync.h,
inflate_lane.h, lz_resolve.h), built for the host with its SIMT regions run thread by thread, must agree with the
oracle on status, output bytes and bytes consumed.
The GPU tier (test_gpu_deflate.py) repeats the same cases through the C ABI on the device."""
import random

import pytest

import _emu as E
import _oracle as O
import _streams as S

@pytest.fixture(autouse=True, params=["one-wavefront", "team"])
def phase1_form(request):
    """Phase 1 has two forms that must write the same records: one wavefront per stream (launches of many streams) and a TEAM of
    wavefronts per stream (inflate_sync.h: the master on the job, five helpers on the rounds behind the master's, whose results the
    master adopts when their first sub-chunk began where the round in front ended; kernels.hip: launches of up to 256 streams).  The
    emulation runs the helpers' rounds one after the other where the device has its barrier.  Every test here runs with both."""
    E.lib.emu_set_deflate_team(1 if request.param == "team" else 0)
    yield request.param
    E.lib.emu_set_deflate_team(0)


@pytest.fixture
def inflate():
    return E.inflate


def _check(inflate, inputs, caps=None):
    exp = [O.deflate(z) for z in inputs]
    if caps is None:
        caps = [max(len(e[1]), 1) for e in exp]
    res = inflate(inputs, caps)
    for i, (r, e) in enumerate(zip(res, exp)):
        st, out, cons, _ = r
        assert st == e[0], "status mismatch on input %d (%s...)" % (i, inputs[i][:12].hex())
        if e[0] == 0:
            assert out == e[1] and cons == e[2], "output/consumed mismatch on input %d" % i


def test_valid_corpus(inflate):
    pairs = S.valid_deflate_corpus()
    _check(inflate, [z for z, _ in pairs])
    for (z, x), r in zip(pairs, inflate([z for z, _ in pairs], [len(x) for _, x in pairs])):
        assert r[:2] == (0, x)


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("misalign", [0, 1, 7, 15])
def test_resolve_thread_order_and_output_alignment(order, misalign):
    """Phase 2 (lz_resolve.h) is written as SIMT regions (csrc/simt.h): its result must not depend on the order in which
    the emulated threads of a region run (a region that reads what another thread of the same region writes would), nor
    on the alignment of the output buffer (slots are 16-byte aligned HBM chunks)."""
    pairs = S.valid_deflate_corpus(sizes=(0, 1, 15, 16, 17, 100, 5000, 65536, 70000, 200000))
    E.set_order(order)
    try:
        res = E.inflate([z for z, _ in pairs], [len(x) for _, x in pairs], misalign=misalign)
    finally:
        E.set_order(0)
    assert [r[:2] for r in res] == [(0, x) for _, x in pairs]


def test_crafted_reference_semantics(inflate):
    cr = S.crafted_deflate()
    ins = [z for _, z in cr]
    exp = [O.deflate(z) for z in ins]
    res = inflate(ins, [600] * len(ins))
    for (name, _), r, e in zip(cr, res, exp):
        assert r[0] == e[0], name
        # partial output up to the error is not part of the contract, but for these vectors it pins
        # the heap-overwrite / shadowing semantics of over-subscribed sets:
        assert r[1] == e[1][:600], name
    by_name = {n: e for (n, _), e in zip(cr, exp)}
    assert by_name["oversub-ok"][:2] == (0, b"BB")
    assert by_name["oversub-ok-shadowed-A"][:2] == (0, b"B")      # 'A' was overwritten by EOB
    assert by_name["dist-too-far"][0] == 900 and by_name["dist-sym-30"][0] == 103


def test_fuzz_status_parity(inflate):
    O.lib.refcpu_set_max_output(1 << 24)
    ins = S.fuzz_deflate()
    exp = [O.deflate(z) for z in ins]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    rnd = random.Random(7)
    _check(inflate, [ins[i] for i in keep], [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 5]) for i in keep])
    O.lib.refcpu_set_max_output(1 << 30)


def test_capacity_reports_required_size(inflate):
    pairs = S.valid_deflate_corpus(sizes=(5000, 70000))
    ins = [z for z, _ in pairs]
    res = inflate(ins, [len(x) // 2 for _, x in pairs])
    for (z, x), r in zip(pairs, res):
        assert r[0] == 901 and r[3] == len(x)                      # SWC_E_CAPACITY, out_len = bytes required
        assert r[1] == x[:len(x) // 2]


@pytest.mark.parametrize("lanes", [1, 65, 130])
def test_table_reuse_across_jobs(lanes):
    """Job after job through the same LDS: tables of a previous job must not leak."""
    pairs = S.valid_deflate_corpus(sizes=(100, 5000))
    pairs = (pairs * (lanes // len(pairs) + 1))[:lanes]
    res = E.inflate([z for z, _ in pairs], [len(x) for _, x in pairs])
    assert [r[:2] for r in res] == [(0, x) for _, x in pairs]


def test_randomised_encoder_settings(inflate):
    """Every zlib strategy / level / window size on every payload class (static blocks, literal-only blocks, distance-1
    runs, tiny windows, full flushes): the two-phase path (entropy decode + LZ77 resolve, host build) against zlib."""
    import zlib
    from swcompression_amd import corpus
    rnd = random.Random(99)
    gens = [corpus.p_text, corpus.p_rep, corpus.p_mix, corpus.p_rand, lambda n, s: corpus.p_zero(n)]
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    plains, streams = [], []
    for i in range(75):
        n = rnd.choice([0, 1, 2, 17, 255, 256, 257, 1000, 4095, 20000, 65535, 65536, 65537, 90000])
        p = gens[i % len(gens)](n, 2000 + i)
        co = zlib.compressobj(rnd.choice([1, 2, 4, 6, 9]), zlib.DEFLATED, -rnd.choice([9, 10, 12, 15]), rnd.choice([1, 4, 8, 9]),
                              strategies[(i // len(gens)) % len(strategies)])
        z = co.compress(p[:len(p) // 2]) + co.flush(zlib.Z_FULL_FLUSH if i % 7 == 0 else zlib.Z_NO_FLUSH) + co.compress(p[len(p) // 2:]) + co.flush()
        plains.append(p)
        streams.append(z)
    res = inflate(streams, [max(len(p), 1) for p in plains])
    for i, (r, p) in enumerate(zip(res, plains)):
        assert r[0] == 0 and r[1] == p and r[2] == len(streams[i]), "stream %d (len %d)" % (i, len(p))


def test_randomised_encoder_settings_gpu_seed(inflate):
    """The 240 streams of the GPU tier's batch (test_gpu_deflate.py, seed 20260926) on the host build: stream 191 of it
    (Z_RLE on a repeated phrase: sub-chunks of > 100 five-bit literals, an end-of-block symbol early in a round) once sent
    the lanes behind the end of the block through several sub-chunks of the round's scratch."""
    import zlib
    from swcompression_amd import corpus
    rnd = random.Random(20260926)
    gens = [corpus.p_text, corpus.p_rep, corpus.p_mix, corpus.p_rand, lambda n, s: corpus.p_zero(n)]
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    plains, streams = [], []
    for i in range(240):
        n = rnd.choice([0, 1, 2, 17, 255, 256, 257, 1000, 4095, 20000, 65535, 65536, 65537, 100000, 250000, 600000])
        p = gens[i % len(gens)](n, 1000 + i)
        co = zlib.compressobj(rnd.choice([1, 2, 4, 6, 9]), zlib.DEFLATED, -rnd.choice([9, 10, 12, 15]), rnd.choice([1, 4, 8, 9]),
                              strategies[(i // len(gens)) % len(strategies)])
        z = co.compress(p[:len(p) // 2]) + co.flush(zlib.Z_FULL_FLUSH if i % 7 == 0 else zlib.Z_NO_FLUSH) + co.compress(p[len(p) // 2:]) + co.flush()
        if n <= 100000:
            plains.append(p)
            streams.append(z)
    res = inflate(streams, [max(len(p), 1) for p in plains])
    for i, (r, p) in enumerate(zip(res, plains)):
        assert r[0] == 0 and r[1] == p and r[2] == len(streams[i]), "stream %d (len %d)" % (i, len(p))


@pytest.mark.parametrize("flush", ["sync", "partial", "full"])
@pytest.mark.parametrize("n", [1, 2, 64, 65, 500, 2000, 20000])
def test_a_block_per_byte_at_exact_capacity(inflate, flush, n):
    """Valid streams made of MANY tiny blocks (a flush after every byte: one literal per block, or empty stored blocks in
    between) at out_cap == len: the record list is sized from the capacity, so a decoder that closes the literal run with a
    record at every block end runs out of records (round-2 advisor: status 904 with wrong output).  The run is carried
    across block boundaries instead."""
    import zlib
    from swcompression_amd import corpus
    p = corpus.p_text(n, 4242 + n)
    mode = {"sync": zlib.Z_SYNC_FLUSH, "partial": zlib.Z_PARTIAL_FLUSH, "full": zlib.Z_FULL_FLUSH}[flush]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = b"".join(co.compress(p[i:i + 1]) + co.flush(mode) for i in range(n)) + co.flush()
    assert zlib.decompress(z, -15) == p
    e = O.deflate(z)
    assert e[:2] == (0, p)
    r = inflate([z], [n])[0]
    assert r[0] == 0 and r[1] == p and r[2] == e[2]


def test_the_team_adopts_its_helpers_rounds():
    """Text of 256 KiB is 19 rounds of the fast path: with five helpers the master decodes four of them and adopts fifteen (a helper's
    lane 0 walks 544 bits before its end is used: in step 99.3 % of the time).  Incompressible data has no rounds to adopt."""
    import ctypes as C
    from swcompression_amd import corpus
    E.lib.emu_team_adopted.restype = C.c_uint64
    E.lib.emu_set_deflate_team(1)
    for kind, least in (("text", 12), ("mix", 12), ("bin", 12), ("rand", 0)):
        p = corpus.PAYLOADS[kind](262144, 3)
        z = corpus.deflate_raw(p, 6)
        E.lib.emu_team_adopted(1)
        st, out, cons, n = E.inflate([z], [len(p)])[0]
        assert (st, out, cons) == (0, p, len(z))
        got = E.lib.emu_team_adopted(1)
        assert got >= least and (least or got == 0), (kind, got)
