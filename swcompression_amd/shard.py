"""Multi-GPU sharding of a batch of independent units.

The decode path has no exchange step: units are independent, so the unit list is cut into `world`
contiguous ranges balanced by the algorithmic bytes sum(C_i + U_i) and every rank decodes its own range on
its own GPU.  `torch.distributed` (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests) carries bookkeeping only:

  * all_gather of every rank's decompressed byte count  -> each rank's base offset in the global output,
  * all_reduce(MAX) of an error flag + all_reduce(MIN) of the first failing global unit index,
  * optionally all_gather of the per-unit (status, out_len) table when the caller wants it everywhere.

Output bytes stay on the GPU that produced them (a gather into one root would be bound by that root's
seven inbound xGMI links); callers get a scatter list (rank, device offset, length) per unit.
"""
import numpy as np

__all__ = ["balanced_ranges", "my_range", "Bookkeeping", "exchange_bookkeeping", "decode_sharded"]


def balanced_ranges(costs, world):
    """Cut range(len(costs)) into `world` contiguous ranges whose cost sums are as equal as a prefix-sum
    split allows.  Returns [(lo, hi)] * world; ranges may be empty when there are fewer units than ranks."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if world <= 0:
        raise ValueError("world must be positive")
    if n == 0:
        return [(0, 0)] * world
    pre = np.concatenate([[0.0], np.cumsum(costs)])
    total = pre[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        # first index whose prefix reaches the target, then pick the closer of the two neighbours
        k = int(np.searchsorted(pre, target, side="left"))
        k = min(max(k, cuts[-1]), n)
        if k > cuts[-1] and k <= n and abs(pre[k - 1] - target) <= abs(pre[k] - target):
            k -= 1
        cuts.append(max(k, cuts[-1]))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def my_range(costs, rank, world):
    return balanced_ranges(costs, world)[rank]


class Bookkeeping:
    """Result of exchange_bookkeeping(): identical on every rank."""

    def __init__(self, bytes_per_rank, first_error_unit, error_status):
        self.bytes_per_rank = [int(x) for x in bytes_per_rank]
        self.base = [int(x) for x in np.concatenate([[0], np.cumsum(self.bytes_per_rank)[:-1]])]
        self.total_bytes = int(sum(self.bytes_per_rank))
        self.first_error_unit = first_error_unit  # global unit index or None
        self.error_status = error_status          # swc_status of that unit or 0

    @property
    def ok(self):
        return self.first_error_unit is None


def exchange_bookkeeping(statuses, out_lens, lo, device="cpu", group=None):
    """statuses/out_lens: this rank's per-unit results for global units [lo, lo + len).  Collective."""
    import torch
    import torch.distributed as dist
    statuses = np.asarray(statuses, dtype=np.int64)
    out_lens = np.asarray(out_lens, dtype=np.int64)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = torch.tensor([int(out_lens[statuses == 0].sum())], dtype=torch.int64, device=device)
    bad = np.nonzero(statuses != 0)[0]
    big = np.iinfo(np.int64).max
    first = torch.tensor([int(lo + bad[0]) if len(bad) else big], dtype=torch.int64, device=device)
    if world > 1:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        dist.all_reduce(first, op=dist.ReduceOp.MIN, group=group)
        per_rank = [int(g.item()) for g in gathered]
    else:
        per_rank = [int(mine.item())]
    first_unit = int(first.item())
    status = torch.tensor([0], dtype=torch.int64, device=device)
    if first_unit != big:
        if lo <= first_unit < lo + len(statuses):
            status[0] = int(statuses[first_unit - lo])
        if world > 1:
            dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    return Bookkeeping(per_rank, None if first_unit == big else first_unit, int(status.item()))


def decode_sharded(codec, units, caps, decode_fn=None, device=None, group=None, **batch_kw):
    """Decode this rank's share of `units` and exchange the bookkeeping.

    Returns (lo, hi, local, book): `local` is whatever decode_fn returned for units[lo:hi]; by default a
    launched `DeviceBatch` (HIP path).  `decode_fn(codec, units, caps, **kw)` must return an object with
    `.statuses` and `.out_lens` arrays, or a DeviceBatch.  Tests inject the host emulation here; the product
    default is the GPU and fails loudly without one."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    costs = [len(u) + int(c) for u, c in zip(units, caps)]
    lo, hi = my_range(costs, rank, world)
    if decode_fn is None:
        from .batch import DeviceBatch
        if device is None:
            device = "cuda:%d" % torch.cuda.current_device()

        def decode_fn(codec, u, c, **kw):
            b = DeviceBatch(codec, u, c, device=device, **kw)
            b.launch(sync=True)
            return b
    kw = {k: (v[lo:hi] if isinstance(v, (list, tuple, np.ndarray)) and len(v) == len(units) else v) for k, v in batch_kw.items()}
    local = decode_fn(codec, units[lo:hi], caps[lo:hi], **kw) if hi > lo else None
    if local is None:
        st, ol = np.zeros(0, np.int64), np.zeros(0, np.int64)
    elif hasattr(local, "results"):
        r = local.results()
        st, ol = r["status"].astype(np.int64), r["out_len"].astype(np.int64)
    else:
        st, ol = np.asarray(local.statuses), np.asarray(local.out_lens)
    bdev = device if (device is not None and dist.is_initialized() and dist.get_backend(group) == "nccl") else "cpu"
    book = exchange_bookkeeping(st, ol, lo, device=bdev, group=group)
    return lo, hi, local, book
