#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X decode engine.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches it under torch.distributed.run, one rank per GPU.  A "step" is ONE pass of the hot path (one
batched launch) over the whole synthetic batch with inputs already resident in HBM.  Default workload =
BASELINE.json configs[1]: 100,000 independent gzip members of 64 KiB (Deflate, dynamic Huffman), host-side
framing done before the timed region.  Units shard across ranks with no data-path collective
("weak" scaling: every rank decodes its own 100,000 members); RCCL is used only for the barrier and
the max-over-ranks time.

Prints ONE JSON line: decompressed GiB/s (sum of U over all ranks / max time), plus
  roofline     -- HBM roofline of the dominant kernel: algorithmic bytes (C + U per unit)
                  / mean launch duration measured with HIP events on the launch stream, vs 8 TB/s;
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm) timed on a bounded sample of the
                  same workload on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (codec, unit kind for corpus.build_units, n_distinct, tile, unit_size, description)
    "deflate64k": ("deflate", "gzip", 4000, 25, 65536, "100000 x 64 KiB gzip members (BASELINE configs[1])"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="deflate64k", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the batch (debug only; <1 is not a valid headline run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def strip_framing(kind, units):
    """Host-side block discovery for the bench corpus: returns the raw codec units and per-unit aux."""
    if kind == "gzip":
        # corpus.gzip_member writes a fixed 10-byte header and an 8-byte trailer (CRC-32, ISIZE)
        return [u[10:-8] for u in units]
    return units


def cpu_baseline(workload, units_raw, plains, seconds):
    """Times the oracle (single thread, like the reference) on as many units as fit in `seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    fn = {"deflate": O.deflate}[WORKLOADS[workload][0]]
    t0 = time.perf_counter()
    done = 0
    nbytes = 0
    while done < len(units_raw) and time.perf_counter() - t0 < seconds:
        st, out, _ = fn(units_raw[done])
        assert st == 0 and len(out) == len(plains[done])
        nbytes += len(out)
        done += 1
    dt = time.perf_counter() - t0
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s decompressed", "cores": 1, "kind": "port",
            "sample": "%d of the %d distinct units, %.1f s, oracle/librefcpu.so single thread" % (done, len(units_raw), dt),
            "compressed_MBps": sum(len(u) for u in units_raw[:done]) / dt / 1e6}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")  # "nccl" IS RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    from swcompression_amd import corpus
    from swcompression_amd.batch import DeviceBatch

    codec, kind, n_distinct, tile, unit_size, desc = WORKLOADS[args.workload]
    n_distinct = max(64, int(n_distinct * args.scale))
    # every rank decodes its own, differently seeded, batch: independent units, no exchange step
    units, plains = corpus.build_units(kind, n_distinct, unit_size, seed=2 + 100003 * rank)
    raw = strip_framing(kind, units)
    batch = DeviceBatch(codec, raw, [unit_size] * n_distinct, tile=tile, device=device)
    sum_u = sum(len(p) for p in plains) * tile
    sum_c = sum(len(r) for r in raw) * tile

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.launch()
    barrier()
    r = batch.results()
    if not ((r["status"] == 0).all() and (r["out_len"] == unit_size).all()):
        raise SystemExit("decode failed: statuses %s" % sorted(set(r["status"].tolist())))

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        batch.launch()
        e.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [s.elapsed_time(e) for s, e in ev]

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # parity spot check outside the timed region: one tile against the plain payloads
    if rank == 0:
        for i in range(0, n_distinct, max(1, n_distinct // 16)):
            assert batch.output(i, unit_size) == plains[i], "bit-exactness violated on unit %d" % i

    if rank == 0:
        total_u = sum_u * world * args.steps
        mean_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = (sum_c + sum_u) / (mean_ms * 1e-3) / 1e9
        line = {
            "metric": "decompressed GiB/s", "value": total_u / dt_max / 2**30, "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc if args.scale == 1.0 else desc + " (scaled x%g, not a headline run)" % args.scale,
                       "codec": codec, "units_per_gpu": batch.n, "unit_bytes": unit_size,
                       "compressed_bytes_per_gpu": sum_c, "decompressed_bytes_per_gpu": sum_u,
                       "payload": "P-text (Zipf pseudo-words), zlib level 6 as encoder", "parallelism": "%d x independent shards" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "swc_inflate_kernel", "kernel_ms": mean_ms,
                         "algorithmic_bytes_per_launch": sum_c + sum_u},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, raw, plains, args.cpu_seconds)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
