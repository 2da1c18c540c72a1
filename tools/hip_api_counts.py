#!/usr/bin/env python3
"""HIP API calls of a rocprofv3 --hip-trace run (rocpd database): name, count, average duration -- which runtime calls
a step makes, and how many.  usage: hip_api_counts.py <rocprof output dir>"""
import glob
import os
import sqlite3
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
c = sqlite3.connect(f[0])
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
for t in tables:
    cols = [r[1] for r in c.execute(f"pragma table_info('{t}')")]
    if "name" in cols and ("start" in cols or "duration" in cols) and t.lower().startswith(("regions", "rocpd_region", "region")):
        dur = "duration" if "duration" in cols else "(end - start)"
        try:
            rows = c.execute(f"select name, count(*), avg({dur}) from '{t}' group by name order by count(*) desc limit 40").fetchall()
        except sqlite3.Error as e:
            print(t, "error", e)
            continue
        print(f"# {t}")
        for name, n, avg in rows:
            print(f"{n:8d} {avg / 1e3 if avg else 0:9.2f} us  {name}")
print("# tables:", [t for t in tables if "region" in t.lower() or "api" in t.lower()][:20])
