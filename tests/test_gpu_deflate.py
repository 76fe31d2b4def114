"""GPU tier (MI355X): the HIP Deflate path, reached through the C ABI, against the CPU oracle.

 - single-shot calls (host buffers) on the reference's payload classes, inline golden vectors, crafted
   App.-A streams and framing error cases -- the same cases the CPU tier runs on the host emulation;
 - the batched many-buffer launch on >= 4096 DISTINCT 64 KiB dynamic-Huffman blocks, bit-exact;
 - BASELINE.json config 2 at full size (100,000 x 64 KiB members) through size-independent
   properties: every job OK with the declared length, every replica tile identical on device,
   CRC-32 of sampled outputs equal to the gzip trailers.
"""
import json
import os
import random
import zlib

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

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_inline_vectors.json")))


def test_device_present_and_native_library_loaded():
    assert swc.device_available(), "the HIP engine must run on the GPU box (no CPU fallback exists)"


def _batch_vs_oracle(inputs, caps=None, check_partial=False):
    exp = [O.deflate(z) for z in inputs]
    if caps is None:
        caps = [max(len(e[1]), 1) for e in exp]
    b = DeviceBatch("deflate", inputs, caps)
    b.launch(sync=True)
    r = b.results()
    for i, e in enumerate(exp):
        assert int(r["status"][i]) == e[0], "status mismatch on input %d (%s...)" % (i, inputs[i][:12].hex())
        if e[0] == 0:
            assert int(r["out_len"][i]) == len(e[1]) and int(r["in_consumed"][i]) == e[2]
            assert b.output(i, len(e[1])) == e[1], "bytes differ on input %d" % i
        elif check_partial:
            n = min(len(e[1]), int(caps[i]))
            assert b.output(i, n) == e[1][:n]
    return r


@pytest.mark.parametrize("vec", GOLD["deflate"], ids=lambda v: v["name"])
def test_inline_golden_vectors_single_shot(vec):
    data = bytes.fromhex(vec["input"])
    if vec["expect"] == "throws":
        with pytest.raises(swc.DeflateError):
            swc.Deflate.decompress(data)
    else:
        assert swc.Deflate.decompress(data) == bytes.fromhex(vec["expect"])


def test_single_shot_payload_classes():
    for z, x in S.valid_deflate_corpus(sizes=(0, 1, 9, 5000, 65536, 300000)):
        out, consumed = swc.Deflate.decompress_consumed(z)
        assert out == x and consumed == len(z)


def test_single_shot_error_taxonomy():
    for name, z in S.crafted_deflate():
        st, out, cons = O.deflate(z)
        if st == 0:
            assert swc.Deflate.decompress_consumed(z) == (out, cons), name
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.Deflate.decompress(z)
            assert ei.value.status == st, name


def test_batch_valid_corpus_and_crafted():
    pairs = S.valid_deflate_corpus()
    _batch_vs_oracle([z for z, _ in pairs])
    cr = S.crafted_deflate()
    _batch_vs_oracle([z for _, z in cr], caps=[600] * len(cr), check_partial=True)


def test_batch_fuzz_status_parity():
    O.lib.refcpu_set_max_output(1 << 24)
    ins = S.fuzz_deflate()
    exp = [O.deflate(z) for z in ins]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    rnd = random.Random(7)
    _batch_vs_oracle([ins[i] for i in keep], [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 5]) for i in keep])
    O.lib.refcpu_set_max_output(1 << 30)


def test_batch_capacity_reports_required_size():
    pairs = S.valid_deflate_corpus(sizes=(5000, 70000))
    b = DeviceBatch("deflate", [z for z, _ in pairs], [len(x) // 2 for _, x in pairs])
    b.launch(sync=True)
    r = b.results()
    for i, (z, x) in enumerate(pairs):
        assert int(r["status"][i]) == 901 and int(r["out_len"][i]) == len(x)
        assert b.output(i, len(x) // 2) == x[:len(x) // 2]


def test_batch_4096_distinct_64k_blocks_bit_exact():
    units, plains = corpus.build_units("deflate", 4096, 65536)
    assert all(((u[0] >> 1) & 3) == 2 and (u[0] & 1) == 1 for u in units[:64])  # single dynamic-Huffman block
    b = DeviceBatch("deflate", units, [65536] * len(units))
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == 65536).all()
    assert (r["in_consumed"] == np.array([len(u) for u in units])).all()
    blob = b.d_out.cpu().numpy()
    for i in range(len(units)):
        o = int(b._out_off[i])
        got = blob[o:o + 65536].tobytes()
        assert got == plains[i], "block %d differs" % i
    # oracle == zlib == engine on a sample (three-way)
    for i in range(0, len(units), 97):
        assert O.deflate(units[i])[:2] == (0, plains[i])


def test_gzip_zlib_framing():
    x = corpus.p_text(70000, 2)
    g = corpus.gzip_member(x)
    assert swc.GzipArchive.unarchive(g) == x
    assert swc.GzipArchive.unarchive(corpus.gzip_member(x[:30000], bgzf=True)) == x[:30000]
    assert swc.GzipArchive.multi_unarchive(corpus.gzip_member(x[:100]) + corpus.gzip_member(x[100:])) == [x[:100], x[100:]]
    assert swc.ZlibArchive.unarchive(zlib.compress(x)) == x
    bad = bytearray(g); bad[-8] ^= 1
    with pytest.raises(swc.GzipError) as ei:
        swc.GzipArchive.unarchive(bytes(bad))
    assert ei.value.case == "wrongCRC" and ei.value.data == x          # GzipTests.swift:190-206
    bad = bytearray(g); bad[-1] ^= 1
    with pytest.raises(swc.GzipError) as ei:
        swc.GzipArchive.unarchive(bytes(bad))
    assert ei.value.case == "wrongISize"
    zl = bytearray(zlib.compress(x)); zl[-1] ^= 1
    with pytest.raises(swc.ZlibError) as ei:
        swc.ZlibArchive.unarchive(bytes(zl))
    assert ei.value.case == "wrongAdler32" and ei.value.data == x      # ZlibTests.swift:59-75
    for name, data in (("empty", b""), ("zero", b"\x00"), ("zeros", bytes(1 << 16))):
        assert O.gzip_unarchive(data)[0] != 0
        with pytest.raises(swc.SWCError) as ei:
            swc.GzipArchive.unarchive(data)
        assert ei.value.status == O.gzip_unarchive(data)[0], name
        with pytest.raises(swc.SWCError) as ei:
            swc.ZlibArchive.unarchive(data)
        assert ei.value.status == O.zlib_unarchive(data)[0], name
    # truncation fuzz (GzipTests.swift:249-291): must raise the same error the reference raises
    rnd = random.Random(3)
    for _ in range(12):
        cut = g[:rnd.randrange(1, len(g))]
        st = O.gzip_unarchive(cut)[0]
        assert st != 0
        with pytest.raises(swc.SWCError) as ei:
            swc.GzipArchive.unarchive(cut)
        assert ei.value.status == st


def test_config2_full_size_properties():
    """BASELINE.json config 2: 100,000 independent 64 KiB members on one GPU."""
    n_distinct, tile = 4000, 25
    units, plains = corpus.build_units("gzip", n_distinct, 65536)
    raw = [u[10:-8] for u in units]                                    # host-side framing: 10-byte header, 8-byte trailer
    b = DeviceBatch("deflate", raw, [65536] * n_distinct, tile=tile)
    assert b.n == 100000
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == 65536).all()
    assert (r["in_consumed"] == np.tile(np.array([len(x) for x in raw]), tile)).all()
    span = n_distinct * 65536
    first = b.d_out[:span]
    for t in range(1, tile):                                           # every replica identical, compared in HBM
        assert bool((b.d_out[t * span:(t + 1) * span] == first).all()), "tile %d differs" % t
    host = first.cpu().numpy()
    for i in range(0, n_distinct, 7):                                  # gzip trailers: CRC-32 + ISIZE
        crc, isize = np.frombuffer(units[i][-8:], dtype="<u4")
        assert isize == 65536 and zlib.crc32(host[i * 65536:(i + 1) * 65536].tobytes()) & 0xFFFFFFFF == crc
    # every one of the 100,000 members against its gzip trailer, CRC-32 computed on the device (swc_batch_crc32)
    want = np.tile(np.array([np.frombuffer(u[-8:-4], dtype="<u4")[0] for u in units], dtype=np.uint32), tile)
    assert (b.crc32() == want).all()


def test_device_crc32_matches_zlib():
    """swc_batch_crc32 (CheckSums.crc32, reference CheckSums.swift:12-28) on outputs of awkward lengths."""
    sizes = [0, 1, 2, 3, 4, 5, 7, 255, 256, 257, 1023, 1024, 1025, 4099, 65535, 65536, 65537, 300001, 1 << 20]
    plains = [corpus.p_mix(n, 40 + i) for i, n in enumerate(sizes)]
    streams = [corpus.deflate_raw(p) for p in plains]
    b = DeviceBatch("deflate", streams, [max(len(p), 1) for p in plains])
    b.launch(sync=True)
    assert (b.results()["status"] == 0).all()
    got = b.crc32()
    for i, p in enumerate(plains):
        assert int(got[i]) == zlib.crc32(p) & 0xFFFFFFFF, "length %d" % len(p)


def test_large_multi_block_streams_two_phase_path():
    """Streams far larger than the 64 KiB resolve window: multi-block zlib output of 3 MiB text (window slides, matches
    reach back across block boundaries), 2 MiB of zeros (258-byte matches at distance 1 chain through every batch),
    incompressible data (stored blocks: literal runs of 64 KiB -> skip records) and a mixed payload."""
    plains = [corpus.p_text(3 << 20, 21), corpus.p_zero(2 << 20), corpus.p_rand(400000, 8), corpus.p_mix(1 << 20, 4),
              corpus.p_rep(700000, 6)]
    streams = [corpus.deflate_raw(p) for p in plains]
    b = DeviceBatch("deflate", streams, [len(p) for p in plains])
    b.launch(sync=True)
    r = b.results()
    for i, p in enumerate(plains):
        st, out, cons = O.deflate(streams[i])
        assert st == 0 and out == p and cons == len(streams[i])
        assert int(r["status"][i]) == 0 and int(r["out_len"][i]) == len(p) and int(r["in_consumed"][i]) == cons, i
        assert b.output(i, len(p)) == p, "bytes differ on stream %d" % i


def test_randomised_encoder_settings_batch():
    """Streams from every zlib strategy / level / window size on every payload class, random sizes, ONE batch: static
    blocks (Z_FIXED), literal-only blocks (Z_HUFFMAN_ONLY), distance-1 runs (Z_RLE), tiny windows (many short-distance
    matches), level 1 (long lazy-free matches) ... all through the two-phase path, bit-exact against zlib's own inflate
    and the oracle's status."""
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
        assert zlib.decompress(z, -15) == p
        plains.append(p)
        streams.append(z)
    b = DeviceBatch("deflate", streams, [max(len(p), 1) for p in plains])
    b.launch(sync=True)
    r = b.results()
    bad = [i for i in range(len(plains)) if int(r["status"][i]) != 0 or int(r["out_len"][i]) != len(plains[i])]
    assert not bad, "status/length wrong on streams %s" % bad[:10]
    for i, p in enumerate(plains):
        assert b.output(i, len(p)) == p, "bytes differ on stream %d (len %d)" % (i, len(p))
    want = np.array([zlib.crc32(p) & 0xFFFFFFFF for p in plains], dtype=np.uint32)
    assert (b.crc32() == want).all()
    for i in range(0, len(plains), 17):
        st, out, cons = O.deflate(streams[i])
        assert (st, out, cons) == (0, plains[i], int(r["in_consumed"][i]))


def test_phase1_team_and_single_wavefront_write_the_same():
    """Phase 1 has two forms (kernels.hip: launch_inflate): a team of six wavefronts per stream for launches of up to 256 streams
    (inflate_team.hip), one wavefront per stream above.  The same 1,200 streams -- encoder settings and sizes of every kind, damaged
    copies among them -- as launches of 200 (teams), as one launch with the teams forced, and with the teams off: the oracle's
    status, bytes, consumed input and length every time."""
    from swcompression_amd import _lib
    lib = _lib.load()
    rnd = random.Random(606)
    gens = [corpus.p_text, corpus.p_rep, corpus.p_mix, corpus.p_rand, lambda n, s: corpus.p_zero(n), corpus.PAYLOADS["bin"]]
    O.lib.refcpu_set_max_output(1 << 24)
    streams = []
    for i in range(1000):
        p = gens[i % len(gens)](rnd.choice([0, 1, 100, 4000, 20000, 65536, 65536, 65536, 150000, 400000]), 3000 + i)
        co = zlib.compressobj(rnd.choice([1, 6, 9]), zlib.DEFLATED, -rnd.choice([9, 12, 15]), rnd.choice([1, 8, 9]), rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE]))
        cut = rnd.randrange(len(p) + 1)
        streams.append(co.compress(p[:cut]) + co.flush(rnd.choice([zlib.Z_NO_FLUSH, zlib.Z_FULL_FLUSH])) + co.compress(p[cut:]) + co.flush())
    for i in range(200):
        b = bytearray(streams[rnd.randrange(1000)])
        if rnd.random() < 0.5 and len(b) > 1:
            b = b[:rnd.randrange(1, len(b))]
        elif b:
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        streams.append(bytes(b))
    exp = [O.deflate(z) for z in streams]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    ins, want = [streams[i] for i in keep], [exp[i] for i in keep]
    caps = [max(len(e[1]), 1) for e in want]

    def check(lo, hi, label):
        b = DeviceBatch("deflate", ins[lo:hi], caps[lo:hi])
        b.launch(sync=True)
        r = b.results()
        for k in range(lo, hi):
            e = want[k]
            assert int(r["status"][k - lo]) == e[0], "%s: status %d, oracle %d on stream %d" % (label, int(r["status"][k - lo]), e[0], k)
            if e[0] == 0:
                assert int(r["out_len"][k - lo]) == len(e[1]) and int(r["in_consumed"][k - lo]) == e[2], (label, k)
                assert b.output(k - lo, len(e[1])) == e[1], "%s: bytes differ on stream %d" % (label, k)
    try:
        assert lib.swc_set_tuning(b"deflate_team", 1) == 0
        for lo in range(0, len(ins), 200):
            check(lo, min(lo + 200, len(ins)), "teams, launches of 200")
        assert lib.swc_set_tuning(b"deflate_team", -1) == 0
        check(0, len(ins), "teams forced, one launch")
        assert lib.swc_set_tuning(b"deflate_team", 0) == 0
        check(0, len(ins), "one wavefront per stream")
        check(0, 150, "one wavefront per stream, a small launch")
    finally:
        lib.swc_set_tuning(b"deflate_team", 1)
        O.lib.refcpu_set_max_output(1 << 30)
