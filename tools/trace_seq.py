#!/usr/bin/env python3
"""Kernel sequence of a rocprofv3 --kernel-trace run: (start us, duration us, stream, queue, name) for a window of
dispatches -- what actually ran, in which order, on which stream.
usage: trace_seq.py <rocprof output dir> [first [count]]   (first < 0: counted from the end)"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
c = sqlite3.connect(f[0])
rows = c.execute("select start, end - start, stream_id, queue_id, name from kernels order by start").fetchall()
first = int(sys.argv[2]) if len(sys.argv) > 2 else -200
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
if first < 0:
    first = max(0, len(rows) + first)
t0 = rows[first][0]
for st, du, sid, qid, name in rows[first:first + count]:
    name = name.replace("(anonymous namespace)::", "")
    print(f"{(st - t0) / 1e3:10.1f} {du / 1e3:7.1f} s{sid:<3} q{qid:<3} {name[:70]}")
