#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the --stats style table:
   python tools/rocpd_summary.py gpurun_out/prof_bench/bench_results.db > profiles/r01_bench_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                 "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-70s %6s %14s %14s %14s %14s %6s  %s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "PCT", "vgpr/agpr/sgpr/lds/scratch grid/wg"))
for r in rows:
    print("%-70s %6d %14d %14.0f %14d %14d %6.2f  %s/%s/%s/%s/%s %s/%s" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                                  r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
