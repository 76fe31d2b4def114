#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X decode engine.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches it under torch.distributed.run, one rank per GPU.  A "step" is ONE pass of the hot path (one
batched launch) over the whole synthetic batch with inputs already resident in HBM.  Default workload =
BASELINE.json configs[1]: 100,000 independent gzip members of 64 KiB (Deflate, dynamic Huffman), host-side
framing done before the timed region.  Units shard across ranks with no data-path collective
("weak" scaling: every rank decodes its own batch); RCCL is used only for the barrier and the
max-over-ranks time.  The other BASELINE configs are parity-test cases; `--workload` times them too
(lz4_4m, bzip2_900k, lzma2_256k) but they are not the headline line.

Prints ONE JSON line: decompressed GiB/s (sum of U over all ranks / max time), plus
  roofline     -- HBM roofline: algorithmic bytes (C + U per unit) / mean kernel time of a
                  launch, measured with HIP events on the launch stream (per kernel for the two Deflate
                  kernels), vs 8 TB/s; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc
                  passes of this same command (profiles/*_traffic.json), or null;
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm) timed on a bounded sample of the
                  same workload on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: codec, corpus kind, n_distinct, tile, unit bytes, description
    "deflate64k": dict(codec="deflate", kind="gzip", n_distinct=4000, tile=25, unit=65536,
                       desc="100000 x 64 KiB gzip members (BASELINE configs[1])", kernels="swc_inflate_kernel + swc_lz_resolve_kernel"),
    "lz4_4m": dict(codec="lz4_block", kind="lz4_block", n_distinct=32, tile=256, unit=4 << 20,
                   desc="8192 x 4 MiB independent LZ4 blocks (BASELINE configs[2], resident micro-config of SURVEY 8d)",
                   kernels="swc_lz4_parse_kernel + swc_lz4_resolve_kernel"),
    "bzip2_900k": dict(codec="bzip2_block", kind="bzip2", n_distinct=32, tile=320, unit=899000,
                       desc="10240 x 900 kB bzip2 blocks (BASELINE configs[3])", kernels="swc_bzip2_stage1/2/3_kernel"),
    "lzma2_256k": dict(codec="lzma2", kind="lzma2", n_distinct=256, tile=40, unit=262144,
                       desc="10240 x 256 KiB raw-LZMA2 units (BASELINE configs[4] shape)", kernels="swc_lzma_kernel"),
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
    ap.add_argument("--tuning", action="append", default=[], help="key=value for swc_set_tuning (comparison runs only)")
    return ap.parse_args()


def make_batch(name, w, n_distinct, seed, device):
    """Host-side block discovery for the bench corpus + the device-resident batch."""
    from swcompression_amd import corpus
    from swcompression_amd.batch import DeviceBatch
    units, plains = corpus.build_units(w["kind"], n_distinct, w["unit"], seed=seed)
    if name == "deflate64k":
        raw = [u[10:-8] for u in units]  # corpus.gzip_member: fixed 10-byte header, 8-byte trailer (CRC-32, ISIZE)
        b = DeviceBatch("deflate", raw, [w["unit"]] * n_distinct, tile=w["tile"], device=device)
    elif name == "lz4_4m":
        raw = units
        b = DeviceBatch("lz4_block", raw, [w["unit"]] * n_distinct, tile=w["tile"], device=device)
    elif name == "bzip2_900k":
        raw = units  # whole one-block streams: "BZh9" (32 bits) + block magic (48) + block CRC (32) => body at bit 112
        b = DeviceBatch("bzip2_block", raw, [w["unit"] + 64] * n_distinct, extra=[112] * n_distinct,
                        dict_values=[int.from_bytes(s[10:14], "big") for s in raw], tile=w["tile"], device=device)
    else:
        raw = units
        db = corpus.lzma2_dict_byte(1 << 20)
        b = DeviceBatch("lzma2", raw, [w["unit"]] * n_distinct, aux=[db] * n_distinct, tile=w["tile"], device=device)
    return b, raw, plains


def cpu_baseline(name, raw, plains, seconds):
    """Times the oracle (single thread, like the reference) on as many units as fit in `seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from swcompression_amd import corpus
    if name == "deflate64k":
        fn = lambda u: O.deflate(u)[:2]
    elif name == "lz4_4m":
        O.lib.refcpu_set_max_output(1 << 23)
        fn = lambda u: O.lz4_block(u)[:2]
    elif name == "bzip2_900k":
        fn = lambda u: O.bzip2(u)[:2]
    else:
        db = corpus.lzma2_dict_byte(1 << 20)
        fn = lambda u: O.lzma2(u, db)[:2]
    t0 = time.perf_counter()
    done = nbytes = cbytes = 0
    while time.perf_counter() - t0 < seconds:   # cycle through the distinct units until the time budget is used
        i = done % len(raw)
        st, out = fn(raw[i])
        assert st == 0 and len(out) == len(plains[i])
        nbytes += len(out)
        cbytes += len(raw[i])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s decompressed", "cores": 1, "kind": "port",
            "sample": "%d unit decodes (cycling over the %d distinct units of the workload), %.1f s, oracle/librefcpu.so, one thread"
                      % (done, len(raw), dt),
            "compressed_MBps": cbytes / dt / 1e6}


def cpu_context(name, raw, plains, seconds):
    """SURVEY.md 8(d) context lines next to the single-thread baseline: the oracle on all host cores (block-parallel, one
    unit per task: oracle/rc_pool.c) and, for the gzip workload, the system zlib on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from swcompression_amd import corpus
    codec = {"deflate64k": 1, "lz4_4m": 2, "bzip2_900k": 3, "lzma2_256k": 4}[name]
    aux = corpus.lzma2_dict_byte(1 << 20) if codec == 4 else 0
    fn = O.lib.refcpu_timed_pool
    fn.restype = C.c_double
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_double,
                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    cores = os.cpu_count() or 1
    n = len(raw)
    ins = (C.c_char_p * n)(*[bytes(r) for r in raw])
    lens = (C.c_size_t * n)(*[len(r) for r in raw])
    ob, ib, un = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dt = fn(codec, aux, ins, lens, n, cores, seconds, C.byref(ob), C.byref(ib), C.byref(un))
    if dt <= 0:
        raise SystemExit("oracle pool failed")
    ctx = {"oracle_all_cores": {"value": ob.value / dt / 2**30, "unit": "GiB/s decompressed", "cores": cores, "kind": "port",
                                "compressed_MBps": ib.value / dt / 1e6,
                                "sample": "%d unit decodes in %.1f s, one unit per task over %d threads (oracle/rc_pool.c)"
                                          % (un.value, dt, cores)}}
    if name == "deflate64k":
        import zlib
        t0 = time.perf_counter()
        nbytes = i = 0
        while time.perf_counter() - t0 < min(seconds, 3.0):
            nbytes += len(zlib.decompress(raw[i % len(raw)], -15))
            i += 1
        dt = time.perf_counter() - t0
        ctx["system_zlib_one_thread"] = {"value": nbytes / dt / 2**30, "unit": "GiB/s decompressed", "cores": 1}
    return ctx


def config1_latency(lib, raw, plains, reps=20):
    """BASELINE configs[0]: ONE 64 KiB Deflate block through the single-shot C ABI (host buffer in, host buffer out, so
    the figure includes both PCIe copies, the launch and the synchronisation)."""
    data = raw[0]
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    used = C.c_size_t()
    ts = []
    for k in range(reps + 2):
        t0 = time.perf_counter()
        st = lib.swc_deflate_decompress(data, len(data), C.byref(out), C.byref(n), C.byref(used))
        dt = time.perf_counter() - t0
        assert st == 0 and n.value == len(plains[0]) and C.string_at(out, n.value) == plains[0]
        lib.swc_free(out)
        if k >= 2:
            ts.append(dt)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"workload": "1 x 64 KiB dynamic-Huffman block, swc_deflate_decompress (BASELINE configs[0])", "median_ms": med * 1e3,
            "compressed_MBps": len(data) / med / 1e6, "decompressed_MiBps": n.value / med / 2**20, "reps": reps}


def committed_traffic(name):
    """HBM bytes per launch from the rocprofv3 --pmc passes of this command (tools/pmc_bench.sh), if committed."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_%s_traffic.json" % name):
                best = os.path.join(pdir, f)
    if not best:
        return None, None
    try:
        d = json.load(open(best))
        return d.get("hbm_bytes_per_launch"), os.path.relpath(best, ROOT)
    except Exception:
        return None, None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(device))  # "nccl" IS RCCL on ROCm

    from swcompression_amd import _lib
    lib = _lib.load()
    for kv in args.tuning:
        k, v = kv.split("=")
        if lib.swc_set_tuning(k.encode(), int(v)) != 0:
            raise SystemExit("unknown tuning " + kv)
    w = WORKLOADS[args.workload]
    n_distinct = max(8, int(w["n_distinct"] * args.scale))
    # every rank decodes its own, differently seeded, batch: independent units, no exchange step
    batch, raw, plains = make_batch(args.workload, w, n_distinct, 2 + 100003 * rank, device)
    unit = w["unit"]
    sum_u = sum(len(p) for p in plains) * w["tile"]
    sum_c = sum(len(r) for r in raw) * w["tile"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.launch()
    barrier()
    r = batch.results()
    if not ((r["status"] == 0).all() and (r["out_len"] == unit).all()):
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
    launch_ms = [s.elapsed_time(e) for s, e in ev]

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    # outside the timed region: per-kernel durations of one more launch (HIP events inside the library, on the launch stream)
    phases = None
    if args.workload == "deflate64k":
        lib.swc_set_tuning(b"phase_timing", 1)
        acc = [0.0, 0.0]
        reps = 3
        for _ in range(reps):
            batch.launch(sync=True)
            buf = (C.c_float * 4)()
            if lib.swc_last_phase_ms(buf, 4) == 2:
                acc[0] += buf[0] / reps
                acc[1] += buf[1] / reps
        lib.swc_set_tuning(b"phase_timing", 0)
        phases = {"swc_inflate_kernel_ms": acc[0], "swc_lz_resolve_kernel_ms": acc[1]}

    # parity checks outside the timed region: one tile against the plain payloads, and (gzip workload) EVERY member
    # against the CRC-32 of its gzip trailer, computed on the device (swc_batch_crc32)
    crc_check = None
    if rank == 0:
        for i in range(0, n_distinct, max(1, n_distinct // 16)):
            assert batch.output(i, unit) == plains[i], "bit-exactness violated on unit %d" % i
        if args.workload == "deflate64k":
            import numpy as np
            from swcompression_amd import corpus
            units, _ = corpus.build_units(w["kind"], n_distinct, w["unit"], seed=2 + 100003 * rank)
            want = np.tile(np.array([np.frombuffer(u[-8:-4], dtype="<u4")[0] for u in units], dtype=np.uint32), w["tile"])
            batch.crc32()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            got = batch.crc32()
            e1.record()
            torch.cuda.synchronize()
            assert (got == want).all(), "device CRC-32 differs from the gzip trailers"
            ms = e0.elapsed_time(e1)
            crc_check = {"members_verified": int(batch.n), "kernel": "swc_crc32_kernel", "ms_incl_readback": ms,
                         "GBps": sum_u / (ms * 1e-3) / 1e9}

    if rank == 0:
        total_u = sum_u * world * args.steps
        mean_ms = sum(launch_ms) / len(launch_ms)
        achieved = (sum_c + sum_u) / (mean_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic(args.workload)
        desc = w["desc"] if args.scale == 1.0 else w["desc"] + " (scaled x%g, not a headline run)" % args.scale
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": w["kernels"], "kernel_ms": mean_ms, "algorithmic_bytes_per_launch": sum_c + sum_u}
        if phases:
            roof["per_kernel_ms"] = phases
            dom = max(phases, key=phases.get)
            roof["dominant_kernel"] = dom[:-3]
            if phases[dom] > 0:
                roof["dominant_kernel_achieved_GBps"] = (sum_c + sum_u) / (phases[dom] * 1e-3) / 1e9
        line = {
            "metric": "decompressed GiB/s", "value": total_u / dt_max / 2**30, "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "codec": w["codec"], "units_per_gpu": batch.n, "unit_bytes": unit,
                       "compressed_bytes_per_gpu": sum_c, "decompressed_bytes_per_gpu": sum_u,
                       "payload": "P-text (Zipf pseudo-words), system encoder (zlib 6 / liblz4 / bz2 9 / xz 6)",
                       "parallelism": "%d x independent shards" % world},
            "roofline": roof,
        }
        if crc_check:
            line["crc32_check"] = crc_check
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, raw, plains, args.cpu_seconds)
            line["cpu_context"] = cpu_context(args.workload, raw, plains, min(args.cpu_seconds, 5.0))
        if world == 1 and args.workload == "deflate64k" and not args.no_cpu_baseline:
            # (skipped together with the CPU legs: the profiling commands of tools/gpu_round.sh want nothing but the
            # batch launches in their kernel statistics)
            line["config1_latency"] = config1_latency(lib, raw, plains)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
