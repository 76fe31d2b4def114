"""Multi-GPU path on CPU: world_size-2 `gloo` process group.  The sharding + bookkeeping collectives of
swcompression_amd/shard.py run for real; the per-rank decode is the host emulation of the lane decoder
(test infrastructure -- on the GPU box the same call decodes through DeviceBatch / HIP)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_ranges_cover_and_balance():
    from swcompression_amd.shard import balanced_ranges
    rng = np.random.default_rng(7)
    for n in (0, 1, 2, 7, 100, 4097):
        costs = rng.integers(1, 100000, size=n)
        for world in (1, 2, 3, 8):
            r = balanced_ranges(costs, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert all(lo <= hi for lo, hi in r)
            if n >= 50 * world:
                sums = [costs[lo:hi].sum() for lo, hi in r]
                assert max(sums) - min(sums) <= 2 * costs.max()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, with_error, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _emu
        from swcompression_amd import corpus
        from swcompression_amd.shard import decode_sharded
        plains = [corpus.p_text(3000 + 517 * i, 40 + i) for i in range(23)] + [b"", corpus.p_rand(999, 3)]
        units = [corpus.deflate_raw(p) for p in plains]
        if with_error:
            units[17] = units[17][: len(units[17]) // 2]  # truncated stream -> DeflateError.symbolNotFound class
        caps = [max(len(p), 1) for p in plains]

        class R:
            pass

        def emu_decode(codec, u, c, **kw):
            res = _emu.inflate(u, c)
            r = R()
            r.statuses = np.array([x[0] for x in res])
            r.out_lens = np.array([x[3] for x in res])
            r.outs = [x[1] for x in res]
            return r

        lo, hi, local, book = decode_sharded("deflate", units, caps, decode_fn=emu_decode)
        outs = local.outs if local is not None else []
        ok = all(outs[i] == plains[lo + i] for i in range(hi - lo) if not (with_error and lo + i == 17))
        q.put((rank, lo, hi, ok, book.bytes_per_rank, book.base, book.first_error_unit, book.error_status, book.total_bytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_error", [False, True])
def test_world2_gloo_shards_and_bookkeeping(with_error):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, with_error, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, ok0, bpr0, base0, fe0, es0, tot0), (r1, lo1, hi1, ok1, bpr1, base1, fe1, es1, tot1) = res
    assert (lo0, hi1) == (0, 25) and hi0 == lo1 and 0 < hi0 < 25     # contiguous cover, both ranks got work
    assert ok0 and ok1                                                # every shard bit-exact
    assert bpr0 == bpr1 and base0 == base1 == [0, bpr0[0]] and tot0 == tot1 == sum(bpr0)   # identical on all ranks
    if with_error:
        assert fe0 == fe1 == 17 and es0 == es1 and es0 != 0
    else:
        assert fe0 is None and fe1 is None and es0 == 0
        sys.path.insert(0, ROOT)
        from swcompression_amd import corpus
        assert tot0 == sum(len(corpus.p_text(3000 + 517 * i, 40 + i)) for i in range(23)) + 999


def _worker_few_units(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _emu
        from swcompression_amd import corpus
        from swcompression_amd.shard import decode_sharded
        # three units for four ranks, one of them carrying nearly all the bytes: at least one rank gets an EMPTY range
        plains = [corpus.p_text(200, 1), corpus.p_text(90000, 2), corpus.p_text(300, 3)]
        units = [corpus.deflate_raw(p) for p in plains]
        caps = [len(p) for p in plains]

        class R:
            pass

        def emu_decode(codec, u, c, **kw):
            res = _emu.inflate(u, c)
            r = R()
            r.statuses = np.array([x[0] for x in res])
            r.out_lens = np.array([x[3] for x in res])
            r.outs = [x[1] for x in res]
            return r

        lo, hi, local, book = decode_sharded("deflate", units, caps, decode_fn=emu_decode)
        outs = local.outs if local is not None else []
        ok = all(outs[i] == plains[lo + i] for i in range(hi - lo))
        q.put((rank, lo, hi, ok, book.bytes_per_rank, book.base, book.first_error_unit, book.error_status, book.total_bytes))
    finally:
        dist.destroy_process_group()


def test_world4_gloo_with_an_empty_range():
    """Four ranks, three units: ranks with nothing to decode still take part in every bookkeeping collective, and everyone
    ends up with the same table (bytes per rank, global output bases, no error)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_few_units, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(4))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ranges = [(lo, hi) for _, lo, hi, *_ in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == 3 and all(ranges[i][1] == ranges[i + 1][0] for i in range(3))
    assert any(lo == hi for lo, hi in ranges)                          # somebody had nothing to do
    assert all(r[3] for r in res)
    tables = {(tuple(r[4]), tuple(r[5]), r[6], r[7], r[8]) for r in res}
    assert len(tables) == 1
    bpr, base, fe, es, tot = next(iter(tables))
    assert tot == 200 + 90000 + 300 and fe is None and es == 0 and list(base) == [sum(bpr[:i]) for i in range(4)]
