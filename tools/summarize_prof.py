#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd .db files under gpurun_out/<dir>) into small text /
json summaries under profiles/.

usage: summarize_prof.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir> [<command profiled>]]

<stats_dir> comes from `rocprofv3 --kernel-trace --stats`, the PMC dirs from separate
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command."""
import collections
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name, w=96):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= w else name[: w // 2 - 2] + " .. " + name[-(w // 2 - 2):]


def db_of(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    if not f:
        raise SystemExit(f"no rocpd .db under {d}")
    return sqlite3.connect(f[0])


def kernel_stats(d):
    """[(name, calls, avg, min, max, total, vgpr, sgpr, lds, grid, wg, median, p99)] (durations in ns), by total time"""
    c = db_of(d)
    rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    durs = collections.defaultdict(list)
    for name, d_ in c.execute("select name, duration from kernels"):
        durs[name].append(d_)
    out = []
    for r in rows:
        v = sorted(durs[r[0]])
        out.append(tuple(r) + (v[len(v) // 2], v[min(len(v) - 1, int(0.99 * len(v)))]))
    return out


def git_sha():
    """the commit the profiled tree was built from: TC_GIT_SHA, the .git_sha file gpurun ships (the box has no .git), or git"""
    import subprocess
    sha = os.environ.get("TC_GIT_SHA")
    if not sha or sha == "unknown":
        try:
            sha = open(os.path.join(ROOT, ".git_sha")).read().strip() or None
        except OSError:
            sha = None
    if not sha:
        try:
            sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            sha = None
    return sha


def pmc(d, counter):
    c = db_of(d)
    agg = collections.OrderedDict()
    for name, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        agg.setdefault(name, []).append(float(v))
    return agg


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "# durations in microseconds", ""]
    rows = kernel_stats(stats_dir)
    tot = sum(r[5] for r in rows) or 1
    out.insert(1, f"# git_sha {git_sha()}")
    out.append(f"{'kernel':96s} {'calls':>6s} {'avg_us':>9s} {'med_us':>9s} {'p99_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} "
               f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>9s} {'wg':>5s}")
    stats = {}
    for name, calls, avg, mn, mx, total, vg, sg, lds, grid, wg, med, p99 in rows:
        out.append(f"{short(name):96s} {calls:6d} {avg/1e3:9.2f} {med/1e3:9.2f} {p99/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*total/tot:6.2f} "
                   f"{vg or 0:5d} {sg or 0:5d} {lds or 0:6d} {grid or 0:9d} {wg or 0:5d}")
        stats[short(name)] = {"calls": calls, "avg_us": avg / 1e3, "median_us": med / 1e3, "p99_us": p99 / 1e3, "min_us": mn / 1e3, "max_us": mx / 1e3}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    if len(sys.argv) >= 5:
        out += ["", "# PMC (separate passes): FETCH_SIZE / WRITE_SIZE in KB per dispatch (raw counter values;",
                "# on gfx950 FETCH_SIZE tallies a 128-B request as 64 B -> hbm_read = 2 x FETCH_SIZE,",
                "# MI355X_MICROARCH.md HBM section; WRITE_SIZE taken as is)", ""]
        f, w = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        out.append(f"{'kernel':96s} {'n':>5s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM_MB(2F+W)':>13s}")
        pm = {}
        for k in f:
            fs, ws = f[k], w.get(k, [0.0])
            fa, wa = sum(fs) / len(fs), sum(ws) / len(ws)
            out.append(f"{short(k):96s} {len(fs):5d} {fa:12.1f} {wa:12.1f} {(2*fa+wa)/1024:13.2f}")
            pm[short(k)] = {"FETCH_SIZE_KB": fa, "WRITE_SIZE_KB": wa, "launches": len(fs)}
        json.dump({"tag": tag, "git_sha": git_sha(), "command": sys.argv[5] if len(sys.argv) > 5 else None, "note": "raw rocprofv3 PMC values per dispatch; gfx950 FETCH_SIZE tallies a 128-B "
                                       "request as 64 B (MI355X_MICROARCH.md HBM section): hbm_read = 2 x FETCH_SIZE",
                   "kernels": pm, "kernel_stats": stats},
                  open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    path = os.path.join(ROOT, "profiles", f"{tag}.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print(path)


if __name__ == "__main__":
    main()
