#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X decode engine.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches it under torch.distributed.run, one rank per GPU.  A "step" is ONE pass of the hot path over the whole
synthetic batch with inputs already resident in HBM.  Headline workload = BASELINE.json configs[1]: 100,000
independent gzip members of 64 KiB (Deflate, dynamic Huffman; 4,096 distinct members tiled at distinct device
addresses), host-side framing done before the timed region; a step is what GzipArchive.unarchive does per member:
the batched Deflate launch AND the CRC-32 of every member (device kernel), both inside the timed region.

Units shard across ranks with no data-path collective.  `--scaling weak` (default): every rank decodes a full per-GPU
batch of its own; `--scaling strong`: ONE unit list is cut into contiguous ranges balanced by sum(C + U)
(swcompression_amd/shard.py) and every rank decodes its range.  RCCL carries only the barrier, the max-over-ranks time
and the byte totals.

Prints ONE compact JSON line (< 4 KB, `compact_line`; the whole object goes to bench_full.json): decompressed GiB/s (sum of U over all ranks / max time), plus
  roofline     -- HBM roofline of the headline: algorithmic bytes (C + U per unit) / mean duration of a
                  launch, measured with HIP events on the launch stream (per kernel for the Deflate kernels), vs 8 TB/s;
                  `traffic` / `l2_hit_rate` = HBM bytes per launch and TCC_HIT / (TCC_HIT + TCC_MISS) per kernel from the
                  committed rocprofv3 --pmc passes of this same command (profiles/*_traffic.json), or null;
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm) timed on a bounded sample of the same workload
                  on this box's host cores (rank 0, N = 1 only);
  per_codec    -- (N = 1) the other BASELINE configs on their stated sizes -- LZ4 8,192 x 4 MiB, BZip2 10,240 x 900 kB,
                  LZMA2 32,768 x 256 KiB -- each with value / roofline / cpu_baseline, every unit verified on the device.
`--workload X` times workload X alone as the headline line.
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
    "deflate64k": dict(codec="deflate", kind="gzip", parts=[("text", 4096)], n_units=100000, unit=65536,
                       desc="100000 x 64 KiB gzip members (BASELINE configs[1]; 4096 distinct, tiled at distinct addresses)",
                       kernels=["swc_inflate_sync_kernel", "swc_lz_copy_kernel"], steps=None),
    "deflate64k_mix": dict(codec="deflate", kind="gzip", parts=[("text", 768), ("mix", 256)], n_units=100000, unit=65536,
                           desc="100000 x 64 KiB gzip members, 3/4 P-text + 1/4 P-mix (stored blocks, literal-only stretches; a per_codec line, not the headline; 1024 distinct)",
                           kernels=["swc_inflate_sync_kernel", "swc_lz_copy_kernel"], steps=10),
    "lz4_4m": dict(codec="lz4_block", kind="lz4_block", parts=[("text", 192), ("mix", 64)], n_units=8192, unit=4 << 20,
                   desc="8192 x 4 MiB independent LZ4 blocks (BASELINE configs[2], resident micro-config of SURVEY 8d; 256 distinct: 192 P-text + 64 P-mix)",
                   kernels=["swc_lz4_lane_kernel", "swc_lz4_parse_kernel", "swc_lz4_copy_kernel"], steps=10),
    "lz4_compress_4m": dict(codec="lz4_compress", kind="lz4_plain", parts=[("text", 192), ("mix", 64)], n_units=8192, unit=4 << 20,
                            desc="ENCODE: 8192 x 4 MiB blocks compressed to LZ4 blocks (LZ4.compress(block:), SURVEY 8f row 4; 256 distinct: 192 P-text + 64 P-mix); "
                                 "value = INPUT GiB/s; every block decoded again on the device and checked",
                            kernels=["swc_lz4_compress_kernel"], steps=10),
    "deflate_compress_64k": dict(codec="deflate_compress", kind="deflate_plain", parts=[("text", 768), ("mix", 256)], n_units=100000, unit=65536,
                                 desc="ENCODE: 100000 x 64 KiB buffers compressed to raw Deflate streams (Deflate.compress(data:), SURVEY 8f row 4; 1024 distinct: "
                                      "768 P-text + 256 P-mix); value = INPUT GiB/s; every stream decoded again on the device and checked",
                                 kernels=["swc_deflate_compress_kernel"], steps=10),
    "bzip2_900k": dict(codec="bzip2_block", kind="bzip2", parts=[("text", 256)], n_units=10240, unit=899000,
                       desc="10240 x 900 kB bzip2 blocks (BASELINE configs[3]; 256 distinct P-text payloads as SURVEY 8d states)",
                       kernels=["swc_bzip2_block_kernel", "swc_bzip2_team_prep+walk_kernel", "swc_bzip2_team_finish_kernel", "swc_bzip2_expand_kernel", "swc_bzip2_crc_kernel"], steps=10),
    "lzma2_256k": dict(codec="lzma2", kind="lzma2", parts=[("text", 256)], n_units=32768, unit=262144,
                       desc="32768 x 256 KiB raw-LZMA2 units (BASELINE configs[4]; 256 distinct P-text payloads as SURVEY 8d states)",
                       kernels=["swc_lzma_kernel"], steps=10),
    "lzma2_256k_bin": dict(codec="lzma2", kind="lzma2", parts=[("bin", 128)], n_units=8192, unit=262144,
                           desc="8192 x 256 KiB raw-LZMA2 units of BINARY RECORDS (corpus.p_bin: all eight classes of the literal coder's context in use -- the worst "
                                "case of the coder cache in LDS; a per_codec line next to the text of BASELINE configs[4]; 128 distinct)",
                           kernels=["swc_lzma_kernel"], steps=3),
}
PAYLOAD_NOTE = ("P-text = Zipf pseudo-words, P-mix = 4 KiB pieces of text / repeated phrase / uniform random bytes (the random pieces come out as "
                "stored blocks, literal-only sequences and incompressible chunks); system encoders (zlib 6 / liblz4 / bz2 9 / xz 6)")


def stats(samples_ms, warmup):
    """The reference's statistics (Sources/swcomp/Benchmarks/RunBenchmarkCommand.swift:66-101): `warmup` discarded iterations,
    then the mean and the POPULATION standard deviation of the timed ones."""
    n = len(samples_ms)
    mean = sum(samples_ms) / n
    var = sum((x - mean) ** 2 for x in samples_ms) / n
    return {"iterations": n, "warmup_discarded": warmup, "mean_ms": mean, "sigma_ms": var ** 0.5, "min_ms": min(samples_ms), "max_ms": max(samples_ms),
            "method": "mean +- population sigma over the timed iterations, per-iteration HIP events / wall clock (RunBenchmarkCommand.swift:66-101)"}


def _r(x, nd=4):
    """Round floats for the compact line (the full precision is in bench_full.json)."""
    return round(x, nd) if isinstance(x, float) else x


def compact_line(line):
    """The LAST line of stdout: the headline object the driver parses, kept well under 4 KB (round 5's full object had grown
    to 24.7 KB on one line and the driver's record came back `parsed: null`).  Like the reference's harness, which prints one
    short line per benchmark (RunBenchmarkCommand.swift:66-101).  Everything else goes to bench_full.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: _r(line[k]) for k in keep if k in line}
    for k in ("rehearsal", "ranks"):
        if k in line:
            out[k] = line[k]
    cfg = line.get("config", {})
    out["config"] = {k: cfg[k] for k in ("workload", "codec", "units_per_gpu", "unit_bytes", "compressed_bytes_per_gpu", "decompressed_bytes_per_gpu", "parallelism")
                     if k in cfg}
    roof = line.get("roofline", {})
    r = {k: _r(roof[k], 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms",
                                      "algorithmic_bytes_per_launch", "dominant_kernel", "dominant_kernel_frac") if k in roof}
    if roof.get("per_kernel_ms"):
        r["per_kernel_ms"] = {k[:-3] if k.endswith("_ms") else k: _r(v, 3) for k, v in roof["per_kernel_ms"].items()}
    out["roofline"] = r
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
    ca = line.get("cpu_baseline_all_cores")
    if ca:
        out["cpu_baseline_all_cores"] = {k: _r(ca[k]) for k in ("value", "unit", "cores", "kind") if k in ca}
    st = line.get("stats")
    if st:
        out["sigma_ms"] = _r(st.get("sigma_ms"))
    if "verify" in line:
        out["units_verified"] = line["verify"].get("units_verified")
    per = line.get("per_codec")
    if per:   # {name: [ms_per_step, GiB/s, roofline frac]}
        out["per_codec_summary"] = {n: [_r(v["ms_per_step"], 2), _r(v["value"], 1), _r(v["roofline"]["frac"], 5)] for n, v in per.items()}
    c1 = line.get("config1_latency")
    if c1:
        out["config1_latency_ms"] = _r(c1.get("median_ms"))
    if "full" in line:
        out["full"] = line["full"]
    return json.dumps(out, separators=(",", ":"))


def write_full(line):
    """The whole object (per_codec, archive_paths, cpu_context, lz4_streamed, L2 hit rates ...) next to the script -- and under
    gpurun_out/ when that exists, so that it comes back from the GPU box."""
    names = [os.path.join(ROOT, "bench_full.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        names.append(os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    wrote = None
    for n in names:
        try:
            with open(n, "w") as f:
                json.dump(line, f, indent=1)
            wrote = wrote or os.path.relpath(n, ROOT)
        except OSError:
            continue
    return wrote


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="deflate64k", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the batch (debug only; <1 is not a valid headline run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-codec", action="store_true")
    ap.add_argument("--streamed-only", action="store_true", help="only the streamed configs[2] mode (waves through pinned host buffers)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--tuning", action="append", default=[], help="key=value for swc_set_tuning (comparison runs only)")
    ap.add_argument("--parts", default=None, help="payload classes of the distinct units, e.g. text:256 or text:192,mix:64 (comparison runs only; "
                                                    "the default is the workload's own mixture)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank REHEARSAL on a box with one GPU: all ranks share cuda:0 and the bookkeeping collectives run over gloo "
                         "(exercises sharding, barriers and the max-over-ranks clock; the figure it prints is not a scaling result)")
    return ap.parse_args()


def scaled_parts(w, scale, override=None):
    parts = [(c, int(k)) for c, k in (x.split(":") for x in override.split(","))] if override else list(w["parts"])
    return [(cls, max(2, int(n * scale))) for cls, n in parts] if scale != 1.0 else parts


def make_batch(name, w, parts, seed, device, select):
    """Host-side block discovery for the bench corpus + the device-resident batch of the units select = (lo, hi) of the
    tiled unit list (unit i of the list is distinct unit i % n_distinct at its own device address)."""
    from swcompression_amd import corpus
    from swcompression_amd.batch import DeviceBatch
    if w["kind"] in ("lz4_plain", "deflate_plain"):   # the encode workloads: the units ARE the plain payloads
        _, plains = corpus.build_units_mixed("lz4_block" if w["kind"] == "lz4_plain" else "gzip", parts, w["unit"], seed=seed)
        units = plains
    else:
        units, plains = corpus.build_units_mixed(w["kind"], parts, w["unit"], seed=seed)
    n_distinct = len(units)
    trailers = None
    kw = {}
    if w["kind"] == "gzip":
        raw = [u[10:-8] for u in units]  # corpus.gzip_member: fixed 10-byte header, 8-byte trailer (CRC-32, ISIZE)
        trailers = [u[-8:] for u in units]
        caps = [w["unit"]] * n_distinct
    elif name == "lz4_4m":
        raw = units
        caps = [w["unit"]] * n_distinct
    elif name == "lz4_compress_4m":
        raw = units
        caps = [len(u) + len(u) // 255 + 16 for u in units]
    elif name == "deflate_compress_64k":
        raw = units
        caps = [len(u) + len(u) // 8 + 16 for u in units]
    elif name == "bzip2_900k":
        raw = units  # whole one-block streams: "BZh9" (32 bits) + block magic (48) + block CRC (32) => body at bit 112
        caps = [w["unit"] + 64] * n_distinct
        kw = dict(extra=[112] * n_distinct, dict_values=[int.from_bytes(s[10:14], "big") for s in raw])
    else:
        raw = units
        caps = [w["unit"]] * n_distinct
        kw = dict(aux=[corpus.lzma2_dict_byte(1 << 20)] * n_distinct)
    b = DeviceBatch(w["codec"], raw, caps, device=device, select=select, **kw)
    return b, raw, plains, trailers


def cpu_baseline(name, raw, plains, seconds):
    """Times the oracle (single thread, like the reference) on as many units as fit in `seconds`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from swcompression_amd import corpus
    if name.startswith("deflate64k"):
        fn = lambda u: O.deflate(u)[:2]
    elif name == "lz4_4m":
        O.lib.refcpu_set_max_output(1 << 23)
        fn = lambda u: O.lz4_block(u)[:2]
    elif name == "lz4_compress_4m":
        fn = lambda u: (lambda r: (r[0], u))(O.lz4_compress_block(u))   # (oracle/rc_lz4c.c; "output" counted = the input bytes)
    elif name == "deflate_compress_64k":
        fn = lambda u: (lambda z: (0, u))(O.deflate_compress(u))        # (oracle/rc_deflatec.c; "output" counted = the input bytes)
    elif name == "bzip2_900k":
        fn = lambda u: O.bzip2(u)[:2]
    elif name.startswith("lzma2_256k"):
        db = corpus.lzma2_dict_byte(1 << 20)
        fn = lambda u: O.lzma2(u, db)[:2]
    else:
        db = corpus.lzma2_dict_byte(1 << 20)
        fn = lambda u: O.lzma2(u, db)[:2]
    t0 = time.perf_counter()
    done = nbytes = cbytes = 0
    while time.perf_counter() - t0 < seconds:   # cycle through the distinct units until the time budget is used
        i = done % len(raw)
        st, out = fn(raw[i])
        assert st == 0 and len(out) == len(plains[i])
        if name.startswith("deflate64k"):   # GzipArchive.unarchive checks the member's CRC-32 (CheckSums.swift:21-28) inside the call
            O.crc32(out)
        nbytes += len(out)
        cbytes += len(raw[i])
        done += 1
    dt = time.perf_counter() - t0
    enc = name in ("lz4_compress_4m", "deflate_compress_64k")   # (the ENCODE workloads time the oracle's ENCODER over the plain payloads)
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s of input" if enc else "GiB/s decompressed", "cores": 1, "kind": "port",
            "sample": "%d unit %s (cycling over the %d distinct units of the workload), %.1f s, oracle/librefcpu.so, one thread"
                      % (done, "encodes" if enc else "decodes", len(raw), dt),
            "compressed_MBps": None if enc else cbytes / dt / 1e6}


def host_cpus():
    """What the box really grants this process: logical CPUs, the affinity mask, and the cgroup CPU quota (a container can
    show 256 logical CPUs and be throttled to a dozen: the all-core line scales with the quota, not with the thread count)."""
    info = {"logical": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = info["logical"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_quota_cpus"] = quota
    info["usable"] = int(min(info["affinity"], quota)) if quota else info["affinity"]
    return info


def cpu_context(name, raw, plains, seconds):
    """SURVEY.md 8(d) lines next to the single-thread baseline: `all_cores` = the oracle with one unit per task on every host
    thread of the box (oracle/rc_pool.c; the reference decodes one unit per call on one thread, so "all host cores" is one
    independent unit per thread) and `system_codec_one_thread` = the system's own decoder (zlib / liblz4 / libbz2 / liblzma)
    on the same units on one thread -- a tuned CPU decoder for orientation, not the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from swcompression_amd import corpus
    if name == "lz4_compress_4m":
        return cpu_context_lz4_compress(raw, seconds)
    if name == "deflate_compress_64k":
        return cpu_context_deflate_compress(raw, seconds)
    codec = {"deflate64k": 1, "deflate64k_mix": 1, "lz4_4m": 2, "bzip2_900k": 3, "lzma2_256k": 4, "lzma2_256k_bin": 4}[name]
    aux = corpus.lzma2_dict_byte(1 << 20) if codec == 4 else 0
    fn = O.lib.refcpu_timed_pool
    fn.restype = C.c_double
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_double,
                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    cpus = host_cpus()
    cores = max(1, cpus["usable"])   # one thread per core the cgroup really grants (256 threads on 16 cores thrash the caches)
    n = min(len(raw), 512)
    ins = (C.c_char_p * n)(*[bytes(r) for r in raw[:n]])
    lens = (C.c_size_t * n)(*[len(r) for r in raw[:n]])
    ob, ib, un = C.c_uint64(), C.c_uint64(), C.c_uint64()
    dt = fn(codec, aux, ins, lens, n, cores, seconds, C.byref(ob), C.byref(ib), C.byref(un))
    ctx = {}
    if dt > 0:
        ctx["all_cores"] = {"value": ob.value / dt / 2**30, "unit": "GiB/s decompressed", "cores": cpus["usable"], "threads": cores, "host_cpus": cpus,
                            "kind": "port", "compressed_MBps": ib.value / dt / 1e6,
                            "sample": "%d unit decodes in %.1f s, one unit per task over %d threads (oracle/rc_pool.c)" % (un.value, dt, cores)}
    import bz2
    import lzma
    import zlib
    if name.startswith("deflate64k"):
        dec, what = (lambda u: zlib.decompress(u, -15)), "zlib %s" % zlib.ZLIB_RUNTIME_VERSION
    elif name == "lz4_4m":
        l4 = corpus._liblz4()
        l4.LZ4_decompress_safe.restype = C.c_int
        l4.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        l4.LZ4_versionString.restype = C.c_char_p
        dst = C.create_string_buffer(4 << 20)

        def dec(u):
            k = l4.LZ4_decompress_safe(u, dst, len(u), 4 << 20)
            return memoryview(dst)[:k]
        what = "liblz4 %s" % l4.LZ4_versionString().decode()
    elif name == "bzip2_900k":
        dec, what = bz2.decompress, "libbz2 (python bz2)"
    else:
        flt = [{"id": lzma.FILTER_LZMA2, "dict_size": 1 << 20}]
        dec, what = (lambda u: lzma.decompress(u, format=lzma.FORMAT_RAW, filters=flt)), "liblzma (python lzma)"
    t0 = time.perf_counter()
    nbytes = cbytes = i = 0
    while time.perf_counter() - t0 < min(seconds, 3.0):
        u = raw[i % len(raw)]
        nbytes += len(dec(u))
        cbytes += len(u)
        i += 1
    dt = time.perf_counter() - t0
    ctx["system_codec_one_thread"] = {"value": nbytes / dt / 2**30, "unit": "GiB/s decompressed", "cores": 1, "decoder": what,
                                      "compressed_MBps": cbytes / dt / 1e6}
    return ctx


def cpu_context_lz4_compress(raw, seconds):
    """liblz4's own block compressor (LZ4_compress_default) on one thread: a tuned CPU encoder for orientation."""
    from swcompression_amd import corpus
    l4 = corpus._liblz4()
    l4.LZ4_compress_default.restype = C.c_int
    l4.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    l4.LZ4_versionString.restype = C.c_char_p
    dst = C.create_string_buffer((4 << 20) + (4 << 20) // 255 + 64)
    t0 = time.perf_counter()
    nbytes = i = 0
    while time.perf_counter() - t0 < min(seconds, 3.0):
        u = raw[i % len(raw)]
        assert l4.LZ4_compress_default(u, dst, len(u), len(dst)) > 0
        nbytes += len(u)
        i += 1
    dt = time.perf_counter() - t0
    return {"system_codec_one_thread": {"value": nbytes / dt / 2**30, "unit": "GiB/s of input", "cores": 1,
                                        "decoder": "liblz4 %s LZ4_compress_default" % l4.LZ4_versionString().decode()}}


def cpu_context_deflate_compress(raw, seconds):
    """zlib at level 1 (its static / dynamic blocks over a fast greedy parse) on one thread, for orientation."""
    import zlib
    t0 = time.perf_counter()
    nbytes = i = 0
    while time.perf_counter() - t0 < seconds:
        u = raw[i % len(raw)]
        zlib.compress(u, 1)
        nbytes += len(u)
        i += 1
    dt = time.perf_counter() - t0
    return {"system_codec_one_thread": {"value": nbytes / dt / 2**30, "unit": "GiB/s of input", "cores": 1, "decoder": "zlib %s compress level 1" % zlib.ZLIB_VERSION}}


def verify_compressed_units(batch, plains, torch, codec="lz4_block"):
    """Every compressed block of the launch is DECODED again on the device (the engine's own LZ4 decoder, reading the
    compressor's output where it lies) and the XXH32 of what comes out is compared with that of the payload; sizes too."""
    import numpy as np
    from swcompression_amd.batch import DeviceBatch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    r = batch.results()
    if not (r["status"] == 0).all():
        raise SystemExit("compression failed: statuses %s" % sorted(set(r["status"].tolist())))
    nd = len(plains)
    dec = DeviceBatch(codec, [b"\x00"] * nd, [len(p) for p in plains], tile=-(-batch.n // nd), device=str(batch.device), select=(0, batch.n))
    jobs = dec._jobs_host.copy()
    jobs["in"] = r["out"]
    jobs["in_len"] = r["out_len"]
    dec._jobs_host = jobs
    dec.d_jobs.copy_(torch.from_numpy(jobs.view(np.uint8)).to(dec.device))
    dec.launch(sync=True)
    d = dec.results()
    want_len = np.array([len(p) for p in plains], dtype=np.uint64)
    ok = bool((d["status"] == 0).all()) and bool((d["out_len"] == want_len[dec.unit_index]).all())
    want = np.array([O.xxh32(p) for p in plains], dtype=np.uint64)
    ok = ok and bool((dec.checksum("xxh32") == want[dec.unit_index]).all())
    if not ok:
        raise SystemExit("round trip of the compressed blocks failed")
    ratio = float(want_len[dec.unit_index].sum()) / float(r["out_len"].sum())
    # the size of the reference encoder restated (oracle/rc_lz4c.c / rc_deflatec.c) on a sample of the distinct units
    k = min(nd, 16)
    sizes = r["out_len"][:nd]
    if codec == "lz4_block":
        ref = sum(len(O.lz4_compress_block(plains[i])[1]) for i in range(k))
    else:
        ref = sum(len(O.deflate_compress(plains[i])) for i in range(k))
    ours = int(sum(int(sizes[i]) for i in range(k))) if batch.n // nd >= 1 else 0
    return {"units_verified": int(batch.n), "method": "every compressed unit decoded on the device by the engine's own decoder; size and XXH32 == the payload's",
            "compression_ratio": ratio, "size_vs_reference_encoder_restated": {"units": k, "engine_bytes": ours, "oracle_bytes": int(ref), "ratio": ours / ref if ref else None}}


def archive_paths(lib):
    """SURVEY 8d-2 / 8d-5 as the ARCHIVE layer sees them (host buffer in, host buffer out, so both PCIe legs, the block
    discovery and the trailer checks are in the figure): a BGZF file of 4,096 x 64 KiB members through
    GzipArchive.multiUnarchive (BSIZE locates the members, one launch) and an `xz --block-size=256KiB` stream of 512 blocks
    through XZArchive.unarchive (the index locates the blocks, one launch)."""
    import shutil
    import subprocess
    import swcompression_amd as swc
    from swcompression_amd import corpus
    res = {}
    parts = [corpus.p_text(65536, 0x5C0DE + 2 + i) for i in range(512)]
    members = [corpus.gzip_member(p, bgzf=True) for p in parts]
    data = b"".join(members * 8)
    iters = 5
    swc.GzipArchive.multi_unarchive(data)   # ONE discarded warm-up of the full size (the reference's method: the first call pins the staging buffers)
    l0 = lib.swc_stat(b"launches")
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = swc.GzipArchive.multi_unarchive(data)
        ts.append((time.perf_counter() - t0) * 1e3)
    dt = sum(ts) / len(ts) / 1e3
    assert len(out) == 4096 and out[0] == parts[0] and out[-1] == parts[-1] and sum(len(o) for o in out) == 4096 * 65536
    res["bgzf_multi_unarchive"] = {"workload": "BGZF file, 4096 members x 64 KiB (512 distinct), GzipArchive.multiUnarchive, host buffers both ways",
                                   "value": 4096 * 65536 / dt / 2**30, "unit": "GiB/s decompressed (PCIe legs, discovery, CRC-32 / ISIZE checks and the Python list of bytes included)",
                                   "seconds": dt, "stats": stats(ts, 1), "launches_per_call": int(lib.swc_stat(b"launches") - l0) // iters, "compressed_bytes": len(data)}
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = swc.GzipArchive.multi_unarchive(data, views=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    dt = sum(ts) / len(ts) / 1e3
    assert len(out) == 4096 and out[0] == parts[0] and out[-1] == parts[-1] and sum(len(o) for o in out) == 4096 * 65536
    del out
    res["bgzf_multi_unarchive_views"] = {"workload": "the same call, members handed over as views INTO the C result (what a Swift shim does with Data(bytesNoCopy:)) "
                                                     "instead of 4096 Python bytes objects copied from it",
                                         "value": 4096 * 65536 / dt / 2**30, "unit": "GiB/s decompressed (PCIe legs, discovery, CRC-32 / ISIZE checks included)",
                                         "seconds": dt, "stats": stats(ts, 1)}
    res["bzip2_compress"] = bzip2_compress_path(lib)
    if shutil.which("xz") is None:
        res["xz_index_unarchive"] = {"error": "no xz command on this box to write a multi-block stream"}
        return res
    x = b"".join(corpus.p_text(262144, 0x5C0DE + 5 + i) for i in range(64)) * 8
    a = subprocess.run(["xz", "-z", "-c", "-T4", "--block-size=262144", "--check=crc64"], input=x, stdout=subprocess.PIPE, check=True).stdout
    swc.XZArchive.unarchive(a)   # one discarded warm-up of the full size
    l0, h0 = lib.swc_stat(b"launches"), lib.swc_stat(b"xz_cache_hits")
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        y = swc.XZArchive.unarchive(a)
        ts.append((time.perf_counter() - t0) * 1e3)
    dt = sum(ts) / len(ts) / 1e3
    assert y == x
    res["xz_index_unarchive"] = {"workload": "xz -T4 --block-size=256KiB stream of 512 blocks (64 distinct), XZArchive.unarchive, host buffers both ways",
                                 "value": len(x) / dt / 2**30, "unit": "GiB/s decompressed (PCIe legs, index walk and CRC-64 checks included)",
                                 "seconds": dt, "stats": stats(ts, 1), "launches_per_call": int(lib.swc_stat(b"launches") - l0) // 3,
                                 "blocks_from_the_batch_per_call": int(lib.swc_stat(b"xz_cache_hits") - h0) // 3, "compressed_bytes": len(a),
                                 "note": "bound by the latency of ONE LZMA2 stream (a 256 KiB block is a serial range-coder chain of about 0.17 s on a wave), not by the host path"}
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        y = swc.XZArchive.unarchive(a, view=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    assert y == x
    del y
    dt = sum(ts) / len(ts) / 1e3
    res["xz_index_unarchive_view"] = {"workload": "the same call, the result handed over as a view OF the C result instead of a bytes object copied from it",
                                      "value": len(x) / dt / 2**30, "unit": "GiB/s decompressed (PCIe legs, index walk and CRC-64 checks included)",
                                      "seconds": dt, "stats": stats(ts, 1)}
    return res


def bzip2_compress_path(lib):
    """SURVEY 8f row 4, BZip2.compress(data:blockSize: .nine): 64 blocks of 720,000 bytes (one launch group), host buffer in,
    host buffer out.  The stream is checked with libbz2 and the engine's own decoder; libbz2 on one core of this box is the
    CPU figure next to it (a stronger encoder than the reference's, which has no build here)."""
    import bz2
    import swcompression_amd as swc
    from swcompression_amd import corpus
    x = b"".join((corpus.p_mix if i % 4 == 3 else corpus.p_text)(720000, 0xB2C + i) for i in range(64))
    swc.BZip2.compress(x, 9)   # one discarded warm-up of the full size
    l0 = lib.swc_stat(b"launches")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        z = swc.BZip2.compress(x, 9)
        ts.append((time.perf_counter() - t0) * 1e3)
    dt = sum(ts) / len(ts) / 1e3
    assert bz2.decompress(z) == x and swc.BZip2.decompress(z) == x
    sample = x[:8 * 720000]
    t0 = time.perf_counter()
    ref = bz2.compress(sample, 9)
    tref = time.perf_counter() - t0
    ours = len(swc.BZip2.compress(sample, 9))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    small = x[:2 * 720000]
    t0 = time.perf_counter()
    restated = O.bzip2_compress(small, 9)
    trest = time.perf_counter() - t0
    return {"workload": "64 x 720,000 bytes (48 text, 16 mix), BZip2.compress(data:blockSize: .nine), host buffers both ways",
            "value": len(x) / dt / 2**30, "unit": "GiB/s uncompressed (PCIe legs, the host's code-length step and the assembly of the stream included)",
            "seconds": dt, "stats": stats(ts, 1), "calls_counted": int(lib.swc_stat(b"launches") - l0), "compressed_bytes": len(z), "ratio": len(z) / len(x),
            "verified": "libbz2 and the engine's own decoder return the input",
            "cpu_baseline": {"kind": "port", "what": "oracle/rc_bzip2c.c, the reference encoder restated (qsort prefix doubling for its suffix array), one core",
                             "GiBps": len(small) / trest / 2**30, "sample_bytes": len(small),
                             "size_vs_reference_encoder_restated": len(swc.BZip2.compress(small, 9)) / len(restated)},
            "libbz2_one_core": {"GiBps": len(sample) / tref / 2**30, "sample_bytes": len(sample), "size_vs_libbz2": ours / len(ref)}}


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
    st = stats([t * 1e3 for t in ts], 2)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"workload": "1 x 64 KiB dynamic-Huffman block, swc_deflate_decompress (BASELINE configs[0])", "median_ms": med * 1e3, "stats": st,
            "compressed_MBps": len(data) / med / 1e6, "decompressed_MiBps": n.value / med / 2**20, "reps": reps}


def committed_traffic(name):
    """HBM bytes per launch and L2 hit rates per kernel from the rocprofv3 --pmc passes of this command
    (tools/pmc_bench.sh), if committed."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_%s_traffic.json" % name):
                best = os.path.join(pdir, f)
    if not best:
        return None, None, None
    try:
        d = json.load(open(best))
        hit = {}
        for k, v in d.get("per_kernel_counters", {}).items():
            h, m = v.get("TCC_HIT_sum", v.get("TCC_HIT")), v.get("TCC_MISS_sum", v.get("TCC_MISS"))
            if h is not None and m is not None and h + m > 0:
                hit[k.split("::")[-1]] = h / (h + m)
        return d.get("hbm_bytes_per_launch"), os.path.relpath(best, ROOT), (hit or None)
    except Exception:
        return None, None, None


def verify_all_units(name, batch, raw, plains, trailers):
    """Every unit of the launch, on the device: a checksum of each job's output against the value computed on the host from
    the plain payloads (gzip: the CRC-32 of the member's trailer, as GzipArchive.unarchive checks it)."""
    import numpy as np
    import zlib
    r = batch.results()
    ok = bool((r["status"] == 0).all())
    if name.startswith("deflate64k"):
        want = np.array([np.frombuffer(t[:4], dtype="<u4")[0] for t in trailers], dtype=np.uint32)
        isize = np.array([np.frombuffer(t[4:], dtype="<u4")[0] for t in trailers], dtype=np.uint64)
        ok = ok and bool((r["out_len"] == isize[batch.unit_index]).all())
        got = batch.crc32()
        how = "device CRC-32 of every member == its gzip trailer, out_len == ISIZE"
    elif name == "lz4_4m":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        want = np.array([O.xxh32(p) for p in plains], dtype=np.uint64)
        got = batch.checksum("xxh32")
        how = "device XXH32 of every block == XXH32 of its payload"
    else:
        want = np.array([zlib.crc32(p) & 0xFFFFFFFF for p in plains], dtype=np.uint32)
        got = batch.crc32()
        how = "device CRC-32 of every unit == CRC-32 of its payload"
    ok = ok and bool((got == want[batch.unit_index]).all())
    if not ok:
        raise SystemExit("verification failed for workload %s" % name)
    return {"units_verified": int(batch.n), "method": how}


def run_workload(name, args, lib, torch, dist, world, rank, device, steps, warmup, with_cpu):
    w = WORKLOADS[name]
    parts = scaled_parts(w, args.scale, args.parts if name == args.workload else None)
    n_distinct = sum(n for _, n in parts)
    n_total = w["n_units"] if args.scale == 1.0 else max(n_distinct, int(w["n_units"] * args.scale))
    if args.scaling == "strong" and world > 1:
        # ONE unit list for the whole job, cut into contiguous ranges balanced by sum(C + U)
        from swcompression_amd import corpus, shard
        seed = 2
        units, pl = corpus.build_units_mixed(w["kind"], parts, w["unit"], seed=seed)
        costs = [len(units[i % n_distinct]) + len(pl[i % n_distinct]) for i in range(n_total)]
        select = shard.balanced_ranges(costs, world)[rank]
    else:
        seed = 2 + 1000003 * rank   # weak: every rank decodes its own, differently seeded, full batch
        select = (0, n_total)
    batch, raw, plains, trailers = make_batch(name, w, parts, seed, device, select)
    unit = w["unit"]
    sum_u = int(sum(len(plains[i]) for i in batch.unit_index))
    sum_c = int(sum(len(raw[i]) for i in batch.unit_index))
    gzip_crc = w["kind"] == "gzip"

    def step():
        batch.launch()
        if gzip_crc:
            batch.crc32_async()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    r = batch.results()
    if not (r["status"] == 0).all():
        raise SystemExit("decode failed: statuses %s" % sorted(set(r["status"].tolist())))
    if name in ("lz4_compress_4m", "deflate_compress_64k"):   # algorithmic bytes of the encode side: the payload read once, the compressed unit written once
        sum_c = int(r["out_len"].sum())

    # The verification after the timed region must see what the LAST timed step wrote, not what the warm-up left behind:
    # before that step the outputs, the result fields of the job records and the CRC buffer are wiped -- between the events
    # of two steps and with the wall clock stopped, so that the wipe is in neither figure.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    dt = 0.0
    t0 = time.perf_counter()
    for k, (s, e) in enumerate(ev):
        if k == steps - 1:
            barrier()
            dt += time.perf_counter() - t0
            batch.wipe_results()
            barrier()
            t0 = time.perf_counter()
        s.record()
        step()
        e.record()
    barrier()
    dt += time.perf_counter() - t0
    launch_ms = [s.elapsed_time(e) for s, e in ev]

    if world > 1:
        cdev = "cpu" if args.rehearse_on_one_gpu else device
        tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
        ts = torch.tensor([float(sum_u), float(sum_c)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        dt_max, tot_u = float(tm[0].item()), float(ts[0].item())
    else:
        dt_max, tot_u = dt, float(sum_u)

    # every unit of the LAST timed step is checked before anything else launches (the phase-timing launches below rewrite
    # all outputs: checking after them would check them)
    if name == "lz4_compress_4m":
        verify = verify_compressed_units(batch, plains, torch)
    elif name == "deflate_compress_64k":
        verify = verify_compressed_units(batch, plains, torch, codec="deflate")
    else:
        verify = verify_all_units(name, batch, raw, plains, trailers)

    # outside the timed region: per-kernel durations of more launches (HIP events inside the library, on the launch stream)
    lib.swc_set_tuning(b"phase_timing", 1)
    names = w["kernels"]
    acc = [0.0] * len(names)
    reps = 3 if gzip_crc else 1
    got = 0
    for _ in range(reps):
        batch.launch(sync=True)
        buf = (C.c_float * 8)()
        if lib.swc_last_phase_ms(buf, 8) == len(names):
            got += 1
            for i in range(len(names)):
                acc[i] += buf[i]
    lib.swc_set_tuning(b"phase_timing", 0)
    phases = {n + "_ms": a / got for n, a in zip(names, acc)} if got else None
    if phases is not None and gzip_crc:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            batch.crc32_async()
        e1.record()
        torch.cuda.synchronize()
        phases["swc_crc32_kernel_ms"] = e0.elapsed_time(e1) / reps

    mean_ms = sum(launch_ms) / len(launch_ms)
    achieved = (sum_c + sum_u) / (mean_ms * 1e-3) / 1e9
    traffic, traffic_src, l2 = committed_traffic(name)
    desc = w["desc"] if args.scale == 1.0 else w["desc"] + " (scaled x%g, not a headline run)" % args.scale
    if args.parts and name == args.workload:
        desc += " (payload classes overridden: %s)" % args.parts
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src, "l2_hit_rate": l2,
            "kernel": " + ".join(w["kernels"] + (["swc_crc32_kernel"] if gzip_crc else [])), "kernel_ms": mean_ms,
            "algorithmic_bytes_per_launch": sum_c + sum_u}
    if phases:
        roof["per_kernel_ms"] = phases
        dom = max(phases, key=phases.get)
        roof["dominant_kernel"] = dom[:-3]
        if phases[dom] > 0:
            roof["dominant_kernel_achieved_GBps"] = (sum_c + sum_u) / (phases[dom] * 1e-3) / 1e9
            roof["dominant_kernel_frac"] = roof["dominant_kernel_achieved_GBps"] / HBM_PEAK_GBS
        if gzip_crc:
            roof["ms_decode_only"] = phases["swc_inflate_sync_kernel_ms"] + phases["swc_lz_copy_kernel_ms"]
    res = {"value": tot_u * steps / dt_max / 2**30, "unit": "GiB/s", "steps": steps, "warmup": warmup, "ms_per_step": dt_max / steps * 1e3,
           "config": {"workload": desc, "codec": w["codec"], "units_per_gpu": int(batch.n), "unit_bytes": unit,
                      "compressed_bytes_per_gpu": sum_c, "decompressed_bytes_per_gpu": sum_u,
                      "payload_classes": dict(parts), "payload": PAYLOAD_NOTE,
                      "parallelism": "%d x independent shards (%s scaling)" % (world, args.scaling)},
           "roofline": roof, "verify": verify, "stats": stats(launch_ms, warmup)}
    if with_cpu and rank == 0:
        res["cpu_baseline"] = cpu_baseline(name, raw, plains, args.cpu_seconds if name == args.workload else min(args.cpu_seconds, 5.0))
        if name != "deflate64k":   # (the headline's context lines are attached by main())
            ctx = cpu_context(name, raw, plains, min(args.cpu_seconds, 4.0))
            if "all_cores" in ctx:
                res["cpu_baseline_all_cores"] = ctx.pop("all_cores")
            res["cpu_context"] = ctx
    return res, batch, raw, plains


def run_lz4_streamed(args, torch, device, waves=12, wave_units=1024, n_distinct=32):
    """BASELINE configs[2] as the job it really is: 1,000,000 x 4 MiB LZ4 blocks (4 TiB of output) do not fit in HBM, so the
    blocks go through the device in WAVES -- host (pinned) -> HBM copy of a wave's compressed blocks, decode, HBM -> host
    copy of its output -- with two buffer sets, so that the copies of wave k-1 / k+1 run on their own HIP streams while
    wave k decodes.  Reported: the steady-state rate INCLUDING both PCIe legs (this is NOT `value`, which is HBM-resident)
    and the time the whole job's per-GPU share would take at that rate."""
    from swcompression_amd import corpus
    from swcompression_amd.batch import DeviceBatch
    unit = 4 << 20
    units, plains = corpus.build_units("lz4_block", n_distinct, unit, seed=77)
    tile = wave_units // n_distinct
    slots = [DeviceBatch("lz4_block", units, [unit] * n_distinct, tile=tile, device=device) for _ in range(2)]
    n_in, n_out = slots[0].d_in.numel(), slots[0].d_out.numel()
    h_in = torch.empty(n_in, dtype=torch.uint8).pin_memory()
    h_in.copy_(slots[0].d_in.cpu())
    h_out = [torch.empty(n_out, dtype=torch.uint8).pin_memory() for _ in range(2)]
    s_in, s_dec, s_out = torch.cuda.Stream(device), torch.cuda.Stream(device), torch.cuda.Stream(device)
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_dec = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    def run(n_waves):
        for k in range(n_waves):
            b = slots[k & 1]
            with torch.cuda.stream(s_in):
                s_in.wait_event(ev_dec[k & 1])       # the decode that last read this slot's input is done
                b.d_in.copy_(h_in, non_blocking=True)
                ev_in[k & 1].record(s_in)
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(ev_in[k & 1])
                s_dec.wait_event(ev_out[k & 1])      # the copy-out that last read this slot's output is done
                b.launch()
                ev_dec[k & 1].record(s_dec)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_dec[k & 1])
                h_out[k & 1].copy_(b.d_out, non_blocking=True)
                ev_out[k & 1].record(s_out)
        torch.cuda.synchronize()

    run(2)
    t0 = time.perf_counter()
    run(waves)
    dt = time.perf_counter() - t0
    # the bytes that came back to the host are the payloads (last wave of each slot, a sample of blocks each)
    for sl in range(2):
        b = slots[sl]
        r = b.results()
        if not (r["status"] == 0).all():
            raise SystemExit("streamed LZ4 decode failed")
        for i in (0, b.n // 3, b.n - 1):
            o = int(b._out_off[i])
            if h_out[sl][o:o + unit].numpy().tobytes() != plains[b.unit_index[i]]:
                raise SystemExit("streamed LZ4: host copy of block %d differs" % i)
    sum_u = wave_units * unit
    sum_c = int(slots[0].total_in)
    rate = waves * sum_u / dt / 2**30
    share = 1000000 // 8
    return {"value": rate, "unit": "GiB/s (decompressed, PCIe legs included)", "waves": waves, "blocks_per_wave": wave_units,
            "ms_per_wave": dt / waves * 1e3, "h2d_bytes_per_wave": sum_c, "d2h_bytes_per_wave": sum_u,
            "pcie_GBps_both_directions": waves * (sum_c + sum_u) / dt / 1e9,
            "projected_seconds_for_125000_blocks_per_gpu": share * unit / 2**30 / rate,
            "note": "double-buffered: H2D, decode and D2H of consecutive waves overlap on three HIP streams; bound by the D2H leg"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.rehearse_on_one_gpu:
        if torch.cuda.device_count() >= world > 1:
            raise SystemExit("--rehearse-on-one-gpu is for boxes with fewer GPUs than ranks; this one has %d" % torch.cuda.device_count())
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(device))  # "nccl" IS RCCL on ROCm

    from swcompression_amd import _lib
    lib = _lib.load()
    for kv in args.tuning:
        k, v = kv.split("=")
        if lib.swc_set_tuning(k.encode(), int(v)) != 0:
            raise SystemExit("unknown tuning " + kv)

    if args.streamed_only:
        print(json.dumps({"lz4_streamed": run_lz4_streamed(args, torch, device)}))
        return
    with_cpu = world == 1 and not args.no_cpu_baseline
    head, batch, raw, plains = run_workload(args.workload, args, lib, torch, dist, world, rank, device, args.steps, args.warmup, with_cpu)
    line = None
    if rank == 0:
        line = {"metric": "decompressed GiB/s", "value": head["value"], "unit": "GiB/s", "n_gpus": 1 if args.rehearse_on_one_gpu else world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
                "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": head["config"], "roofline": head["roofline"],
                "verify": head["verify"], "stats": head["stats"]}
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = True   # all ranks share ONE GPU: exercises the multi-rank contract, not a scaling result
            line["ranks"] = world
        if "cpu_baseline" in head:
            line["cpu_baseline"] = head["cpu_baseline"]
            ctx = head.pop("cpu_context", None) or cpu_context(args.workload, raw, plains, min(args.cpu_seconds, 4.0))
            if "all_cores" in ctx:
                line["cpu_baseline_all_cores"] = ctx.pop("all_cores")
            elif "cpu_baseline_all_cores" in head:
                line["cpu_baseline_all_cores"] = head["cpu_baseline_all_cores"]
            line["cpu_context"] = ctx
        if with_cpu and args.workload == "deflate64k":
            # (skipped together with the CPU legs: profiling commands want nothing but the batch launches in their statistics)
            line["config1_latency"] = config1_latency(lib, raw, plains)
            if not args.no_per_codec and args.scale == 1.0:
                try:
                    line["archive_paths"] = archive_paths(lib)
                except Exception as e:   # (a line of context, never the reason the headline is lost)
                    line["archive_paths"] = {"error": str(e)[:200]}
    del batch
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and args.workload == "deflate64k" and not args.no_per_codec and args.scale == 1.0:
        per = {}
        for name in ("deflate64k_mix", "lz4_4m", "bzip2_900k", "lzma2_256k", "lzma2_256k_bin", "lz4_compress_4m", "deflate_compress_64k"):
            res, b, _, _ = run_workload(name, args, lib, torch, dist, world, rank, device, WORKLOADS[name]["steps"], 1, with_cpu)
            per[name] = res
            del b
            torch.cuda.empty_cache()
        line["per_codec"] = per
        # in a process of its own: the PCIe figure depends on how much other pinned / mapped memory the process holds
        # (46 GiB/s stand-alone, 22 GiB/s at the end of this run in round 2)
        import subprocess
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--streamed-only"], capture_output=True, text=True, timeout=600)
            line["lz4_streamed"] = json.loads(out.stdout.strip().splitlines()[-1])["lz4_streamed"]
        except Exception as e:   # e.g. not enough pinned host memory on the box: say so instead of dropping the headline
            line["lz4_streamed"] = {"error": str(e)[:200]}
    if rank == 0:
        line["full"] = write_full(line)
        sys.stdout.flush()
        print(compact_line(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
