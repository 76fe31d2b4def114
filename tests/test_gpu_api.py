"""GPU tier: properties of the C ABI beyond single results -- concurrent callers, workspace sizing per unit, staging that
does not grow with (units x container size)."""
import io
import threading
import time
import zipfile

import pytest

import _oracle as O
import swcompression_amd as swc
from swcompression_amd import _lib, corpus
from swcompression_amd.zipcontainer import ZipContainer

pytestmark = pytest.mark.gpu


def test_sixteen_threads_share_the_library():
    """include/swc_hip.h promises thread safety: sixteen threads call the single-shot entry points at once (each on its own
    HIP stream, hipStreamPerThread) and every result is the oracle's."""
    payloads = [corpus.p_text(3000 + 7919 * i, 300 + i) for i in range(16)]
    raws = [corpus.deflate_raw(p) for p in payloads]
    gz = [corpus.gzip_member(p) for p in payloads]
    for z, p in zip(raws, payloads):
        assert O.deflate(z)[:2] == (0, p)
    errors = []

    def worker(i):
        try:
            for rep in range(12):
                j = (i + rep) % 16
                out, used = swc.Deflate.decompress_consumed(raws[j])
                assert out == payloads[j] and used == len(raws[j])
                assert swc.GzipArchive.unarchive(gz[j]) == payloads[j]
                with pytest.raises(swc.SWCError):
                    swc.Deflate.decompress(raws[j][:len(raws[j]) // 2])
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(16)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_sixteen_threads_every_codec():
    """The same promise for the other entry points (BZip2.swift:22-26, LZMA.swift:25-34, LZ4.swift:73-91, XZArchive.swift:27-51
    are as re-entrant as Deflate.decompress): sixteen threads, each walking through bzip2 / LZMA / LZMA2 / LZ4 / xz / zlib
    single-shot calls on different inputs at the same time -- valid ones, and a damaged one per codec whose error must be the
    right class."""
    import lzma as pylzma
    payloads = [corpus.p_mix(2500 + 6007 * i, 800 + i) for i in range(16)]
    bz = [corpus.bzip2_stream(p) for p in payloads]
    la = [corpus.lzma_alone(p) for p in payloads]
    l2 = [corpus.lzma2_raw(p) for p in payloads]
    db = corpus.lzma2_dict_byte(1 << 20)
    l4 = [corpus.lz4_frame(p, block_size_code=4, content_checksum=True, xxh32=O.xxh32) for p in payloads]
    xz = [corpus.xz_stream(p, check=pylzma.CHECK_SHA256 if i % 2 else pylzma.CHECK_CRC64) for i, p in enumerate(payloads)]
    zl = [corpus.zlib_stream(p) for p in payloads]
    for i, p in enumerate(payloads):   # the oracle first, single-threaded
        assert O.bzip2(bz[i])[:2] == (0, p) and O.lzma2(l2[i], db)[:2] == (0, p) and O.lz4(l4[i])[:2] == (0, p)
    errors = []

    def worker(i):
        try:
            for rep in range(6):
                j = (i + 3 * rep) % 16
                assert swc.BZip2.decompress(bz[j]) == payloads[j]
                assert swc.LZMA.decompress(la[j]) == payloads[j]
                assert swc.LZMA2.decompress_raw(l2[j], db) == (payloads[j], len(l2[j]))
                assert swc.LZ4.decompress(l4[j]) == payloads[j]
                assert swc.XZArchive.unarchive(xz[j]) == payloads[j]
                assert swc.ZlibArchive.unarchive(zl[j]) == payloads[j]
                with pytest.raises(swc.BZip2Error):
                    swc.BZip2.decompress(bz[j][:-6] + bytes(6))          # stream CRC
                with pytest.raises(swc.SWCError):
                    swc.LZ4.decompress(l4[j][:len(l4[j]) // 2])
                with pytest.raises(swc.SWCError):
                    swc.XZArchive.unarchive(xz[j][:-9] + b"\0" + xz[j][-8:])
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(16)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_eight_threads_compress():
    """The encoders are entry points like the others: eight threads compress different inputs at once with all three of them
    (each call has its own stream, staging buffers and -- BZip2 -- its own device memory from the pool) and every result decodes."""
    import bz2
    import zlib
    payloads = [corpus.p_mix(30000 + 50021 * i, 900 + i) + corpus.p_text(100000, 950 + i) for i in range(8)]
    errors = []

    def worker(i):
        try:
            for rep in range(4):
                x = payloads[(i + rep) % 8]
                assert bz2.decompress(swc.BZip2.compress(x, 1 + (i + rep) % 3)) == x
                assert zlib.decompress(swc.Deflate.compress(x), -15) == x
                assert swc.LZ4.decompress(swc.LZ4.compress(x, block_size=65536)) == x
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]


def test_one_large_unit_among_many_small_ones():
    """Workspace areas are sized per unit (prefix sums), not n x the largest capacity: 4,000 small streams and one of
    64 MiB decode in one call (n x 2.35 x 64 MiB would be more than half a terabyte)."""
    small = [corpus.p_text(2000 + (i % 50) * 40, 700 + i) for i in range(4000)]
    big = corpus.p_text(64 << 20, 9)
    archives = [corpus.deflate_raw(p) for p in small[:2000]] + [corpus.deflate_raw(big)] + [corpus.deflate_raw(p) for p in small[2000:]]
    got = swc.unarchive_many("deflate", archives)
    plains = small[:2000] + [big] + small[2000:]
    assert [st for st, _ in got] == [0] * len(archives)
    assert all(d == p for (_, d), p in zip(got, plains))


def test_one_large_bzip2_block_among_many_small_streams():
    """bzip2's workspace areas are one size per launch (9 x the capacity): the launch is split by capacity so that ONE block
    that expands to 40 MB (a run of zeros: RLE1 packs 40 MB into a single block) does not size the areas of 3,000 small
    streams (3,000 x 9 x 40 MB would be a terabyte and the whole call would fail)."""
    import bz2
    small = [corpus.p_text(1500 + (i % 40) * 50, 900 + i) for i in range(3000)]
    big = bytes(40 << 20)
    plains = small[:1000] + [big] + small[1000:]
    archives = [bz2.compress(p, 9) for p in plains]
    assert archives[1000].count(bytes.fromhex("314159265359")) == 1
    got = swc.unarchive_many("bzip2", archives)
    assert [st for st, _ in got] == [0] * len(archives)
    assert all(d == p for (_, d), p in zip(got, plains))


def test_zip_with_many_entries_stages_the_container_once():
    """Every Deflate entry reads on from its offset to the end of the container (ZipContainer.swift:74), but the container is
    staged once: 3,000 entries in a ~25 MB archive take seconds, not (entries x archive size) of copying."""
    buf = io.BytesIO()
    names, plains = [], []
    with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED) as z:
        for i in range(3000):
            p = corpus.p_text(20000 + (i % 7) * 1000, 50 + i)
            names.append("f%05d.txt" % i)
            plains.append(p)
            z.writestr(names[-1], p)
    data = buf.getvalue()
    t0 = time.time()
    entries = ZipContainer.open(data)
    dt = time.time() - t0
    assert [n for n, _ in entries] == names
    assert all(d == p for (_, d), p in zip(entries, plains))
    assert dt < 60, "took %.1f s" % dt


@pytest.mark.gpu
def test_trim_gives_back_and_the_next_call_still_works():
    """swc_trim(): after a call that left a parked result, pinned staging buffers and pooled device memory behind,
    the trim returns SWC_OK, and the same call afterwards produces the same bytes (everything is simply acquired again)."""
    import swcompression_amd as swc
    from swcompression_amd import corpus
    parts = [corpus.p_text(65536, 4242 + i) for i in range(96)]
    data = b"".join(corpus.gzip_member(p, bgzf=True) for p in parts)     # 6 MiB of output: above the 4 MiB the result cache starts at
    assert swc.GzipArchive.multi_unarchive(data) == parts
    lib = swc._lib.load()
    assert lib.swc_set_tuning(b"result_cache_mib", 0) == 0 and lib.swc_set_tuning(b"pinned_keep_mib", 0) == 0
    assert swc.GzipArchive.multi_unarchive(data) == parts
    assert swc.trim() == 0
    assert swc.GzipArchive.multi_unarchive(data) == parts
    assert lib.swc_set_tuning(b"result_cache_mib", 512) == 0 and lib.swc_set_tuning(b"pinned_keep_mib", 512) == 0
    assert swc.trim() == 0
