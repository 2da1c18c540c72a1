#!/usr/bin/env python3
"""Every kernel of a rocprofv3 --kernel-trace database in start order: start (us from the first), duration, stream, grid, name.
usage: kernel_list.py <rocprof output dir> [skip kernels whose name contains ...]"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
skip = sys.argv[2:]
f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
c = sqlite3.connect(f[0])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((x for x in ("stream_id", "queue_id", "queue") if x in cols), "0")
rows = c.execute(f"select name, start, end, {qcol}, grid_x, workgroup_x from kernels order by start").fetchall()
t0 = rows[0][1]
for name, st, en, q, gx, wx in rows:
    short = name.split("(")[0].replace("void ", "")
    if any(s in short for s in skip):
        continue
    print(f"{(st - t0) / 1e3:11.1f} {(en - st) / 1e3:8.1f} q{q:<3} {gx // max(wx, 1):6d}x{wx:<5d} {short[:90]}")
