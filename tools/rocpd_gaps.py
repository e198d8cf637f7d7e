#!/usr/bin/env python3
"""Idle intervals of the device in a rocprofv3 (rocpd sqlite) kernel trace: between the first and the last dispatch of the
LAST `dvp_depth_to_weak*`/`dvp_sweep_eval` period (one view), every gap longer than --min-ms with the kernels either side.
usage: rocpd_gaps.py results.db [--min-ms 0.3] [--last-ms 1200]"""
import sqlite3
import sys

db = sys.argv[1]
min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 0.3
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 1200.0
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, grid_x from kernels order by start"))
t_end = rows[-1][2]
rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
busy_until = rows[0][1]
total_gap = 0.0
print("# window: the last %.0f ms of the trace, %d dispatches" % (last_ms, len(rows)))
prev = rows[0]
for r in rows:
    if r[1] > busy_until:
        gap = (r[1] - busy_until) / 1e6
        total_gap += gap
        if gap >= min_ms:
            print("%8.3f ms idle at t=%9.3f ms   after %-32s before %-32s (grid %d)" % (gap, (busy_until - rows[0][1]) / 1e6, prev[0][:32], r[0][:32], r[3]))
    if r[2] > busy_until:
        busy_until = r[2]
        prev = r
print("# idle in the window: %.3f ms" % total_gap)
if "--timeline" in sys.argv:
    n = int(sys.argv[sys.argv.index("--timeline") + 1])
    t0 = rows[0][1]
    for r in rows[:n]:
        print("%10.3f .. %10.3f ms  %-36s grid %d" % ((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[0][:36], r[3]))
