#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per (kernel, grid size) calls / total / avg / min /
max — launches of one kernel at different problem sizes (bench workload vs the small parity sample of
the cpu_baseline leg) stay on separate rows.
usage: rocpd_summary.py results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, "
                        "max(end-start)/1e6, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size), "
                        "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name, grid_x order by 3 desc"))
tot = sum(r[2] for r in rows)
# rocprofv3's rocpd `vgpr_count` is HALF the code object's .vgpr_count on gfx950 (dvp_strong_update_v16: 128 here,
# `.vgpr_count 256` / 2 waves per SIMD in the code object; dvp_gen_neighbours_list: 36 vs 71): the column below is
# 2 x rocpd, i.e. the code object's figure rounded up to the allocation granule, so that occupancy can be read off it.
print("# vgpr = 2 x rocpd.vgpr_count (= the code object's .vgpr_count, see tools/rocpd_summary.py)")
print("%-38s %6s %12s %10s %10s %10s %6s %5s %5s %5s %7s %6s %9s %4s" % (
    "kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%", "vgpr", "agpr", "sgpr", "scratch", "lds", "grid_x", "wg"))
for r in rows:
    print("%-38s %6d %12.3f %10.3f %10.3f %10.3f %6.2f %5s %5s %5s %7s %6s %9s %4s" % (
        r[0][:38], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, 2 * (r[6] or 0), r[7], r[8], r[9], r[10], r[11], r[12]))
