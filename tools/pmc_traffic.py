#!/usr/bin/env python3
"""Turn the counter_collection CSVs of tools/pmc_bench.sh into HBM bytes per launch.

Per launch = sum over the kernels of ONE pass of the hot path (the last dispatch of each swc:: kernel).
Units/corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-like units of
1024 B; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced streaming reads, so
read bytes = FETCH_SIZE * 1024 * 2 is an UPPER estimate for kernels whose reads are narrower (reported both ways)."""
import collections, csv, glob, json, sys

d, workload, tag = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(dict)   # kernel -> counter -> value of the last dispatch
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "swc" in r["Kernel_Name"]]
    last = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        last[k] = max(last.get(k, 0), int(r["Dispatch_Id"]))
    acc = collections.defaultdict(float)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Dispatch_Id"]) == last[k]:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, c), v in acc.items():
        per[k][c] = v
# gfx950 half-count correction: only kernels whose reads are wide coalesced streams (16 B per lane).  Calibrated on
# this path: the resolve kernel stages U + 4*records bytes with 16-byte loads and reports exactly half of that;
# the lane-per-stream kernels read scattered dwords and report the compressed bytes 1:1.
# (round 5: phase 2 is swc_lz_copy_kernel for launches of 2,560 streams and more; it reads records as dwords, literals as 16
# bytes per lane and far sources as scattered 8-byte loads -- its FETCH_SIZE is taken 1:1, the conservative reading)
WIDE = ("swc_lz_resolve_kernel", "swc_lz4_resolve_kernel")
# Kernels that verify_all_units launches AFTER the clock has stopped are not part of the step: the XXH32 of the LZ4 outputs and
# the CRC-32 of the BZip2 / LZMA2 outputs.  For the gzip workloads the CRC-32 IS part of the step.
VERIFY = () if workload.startswith("deflate64k") else ("swc_xxh32_kernel", "swc_crc32_kernel", "swc_crc32_group_kernel", "swc_crc32_consts_kernel")
verify = {k: per.pop(k) for k in list(per) if any(v in k for v in VERIFY)}
fetch = sum(v.get("FETCH_SIZE", 0.0) * (2 if any(w in k for w in WIDE) else 1) for k, v in per.items()) * 1024
fetch_raw = sum(v.get("FETCH_SIZE", 0.0) for v in per.values()) * 1024
write = sum(v.get("WRITE_SIZE", 0.0) for v in per.values()) * 1024
out = {"workload": workload, "round": tag, "per_kernel_counters": per, "verification_kernels_not_summed": verify,
       "fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected": fetch, "write_bytes": write,
       "hbm_bytes_per_launch": fetch + write,
       "note": "hbm_bytes_per_launch = FETCH_SIZE*1024 (x2 for the wide-streaming resolve kernel: gfx950 counts a 128-byte request as 64) "
               "+ WRITE_SIZE*1024 (uncalibrated); one launch = last dispatch of each swc:: kernel"}
print(json.dumps(out, indent=1))
