#!/usr/bin/env python3
"""Timeline of a pipelined run from a rocprofv3 --kernel-trace database: per batch, when its grouping chain (k_hist, the
radix passes) and its evaluation started and ended, how long the evaluation waited after its sort was done (slack) and
how long the engine's stream idled between two evaluations (gap).

usage: timeline.py <rocprof output dir> [first_batch [last_batch]]"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    if not f:
        raise SystemExit(f"no rocpd .db under {d}")
    c = sqlite3.connect(f[0])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("# kernels columns:", cols)
    qcol = next((x for x in ("stream_id", "queue_id", "queue") if x in cols), None)
    rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
    # a batch's grouping chain starts with k_hist (LSD passes) or with k_tile_ranges (range path: + k_finish)
    hist = [r for r in rows if "k_hist" in r[0] or "k_tile_ranges" in r[0] or "k_tile_part" in r[0]]
    sweeps = [r for r in rows if "k_onesweep" in r[0] or "k_finish" in r[0] or "k_hot_gather" in r[0]]
    evals = [r for r in rows if "k_eval_sorted" in r[0] or "k_eval_general" in r[0] or "k_eval_lean_hot" in r[0]]
    print(f"# {len(hist)} chain heads (k_hist / k_tile_ranges), {len(sweeps)} k_onesweep / k_finish, {len(evals)} evaluations")
    # the i-th evaluation belongs to the i-th batch; its sort chain is the i-th k_hist (in START order the chains of
    # different streams interleave, so the passes are matched by stream)
    by_q = {}
    for r in sweeps:
        by_q.setdefault(r[3], []).append(r)
    hist_sorted = sorted(hist, key=lambda r: r[1])
    n = min(len(hist_sorted), len(evals))
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, n - 24)
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else n
    t0 = rows[0][1]
    used = {q: 0 for q in by_q}
    chain_end = []
    heads_by_q = {}
    for h in hist_sorted:
        heads_by_q.setdefault(h[3], []).append(h)
    for h in hist_sorted:
        q = h[3]
        lst = by_q.get(q, [])
        k = used.get(q, 0)
        # the chain's kernels: everything on its stream that starts before the stream's next chain head
        later = [x for x in heads_by_q[q] if x[1] > h[1]]
        nxt = later[0][1] if later else float("inf")
        m = k
        while m < len(lst) and lst[m][1] < nxt:
            m += 1
        mine = lst[k:m]
        used[q] = m
        chain_end.append((mine[-1][2] if mine else h[2], q, [(m[1], m[2]) for m in mine]))
    print(f"{'batch':>5s} {'q':>6s} {'hist_start':>10s} {'hist':>6s} {'p1':>6s} {'p2':>6s} {'p3':>6s} {'sort_end':>9s} {'eval_start':>10s} {'eval':>6s} "
          f"{'slack':>7s} {'gap':>6s} {'step':>6s}   (us)")
    prev_end = prev_start = None
    tot = {"eval": 0.0, "gap": 0.0, "slack": 0.0, "step": 0.0, "n": 0}
    for i in range(n):
        h, (se, q, ps), e = hist_sorted[i], chain_end[i], evals[i]
        if lo <= i < hi:
            gap = (e[1] - prev_end) / 1e3 if prev_end else 0.0
            step = (e[1] - prev_start) / 1e3 if prev_start else 0.0
            pd = [(b - a) / 1e3 for a, b in ps] + [0.0] * 3
            print(f"{i:5d} {str(q)[-6:]:>6s} {(h[1]-t0)/1e3:10.1f} {(h[2]-h[1])/1e3:6.1f} {pd[0]:6.1f} {pd[1]:6.1f} {pd[2]:6.1f} {(se-t0)/1e3:9.1f} "
                  f"{(e[1]-t0)/1e3:10.1f} {(e[2]-e[1])/1e3:6.1f} {(e[1]-se)/1e3:7.1f} {gap:6.1f} {step:6.1f}")
            if prev_end:
                tot["eval"] += (e[2] - e[1]) / 1e3
                tot["gap"] += gap
                tot["slack"] += (e[1] - se) / 1e3
                tot["step"] += step
                tot["n"] += 1
        prev_end, prev_start = e[2], e[1]
    if tot["n"]:
        k = tot["n"]
        print(f"# mean over {k} batches: eval {tot['eval']/k:.1f} us, gap {tot['gap']/k:.1f} us, slack after the sort {tot['slack']/k:.1f} us, "
              f"step {tot['step']/k:.1f} us")


if __name__ == "__main__":
    main()
