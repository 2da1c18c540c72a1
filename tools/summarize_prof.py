#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>) into small text summaries
under profiles/.  usage: summarize_prof.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]"""
import collections
import csv
import glob
import os
import sys


def short(name, w=96):
    name = name.replace("(anonymous namespace)::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::")
    return name if len(name) <= w else name[: w // 2 - 2] + " .. " + name[-(w // 2 - 2):]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", ""]
    f = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0]
    out.append(f"{'kernel':96s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for r in csv.DictReader(open(f)):
        out.append(f"{short(r['Name']):96s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f} "
                   f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
    if len(sys.argv) >= 5:
        out += ["", "# PMC (separate passes): FETCH_SIZE / WRITE_SIZE in KB per dispatch (raw counter values;",
                "# on gfx950 FETCH_SIZE counts a 128-B request as 64 B -> double it for streaming/gather reads)", ""]
        agg = collections.OrderedDict()
        for which, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
            for r in csv.DictReader(open(f)):
                agg.setdefault(short(r["Kernel_Name"]), {}).setdefault(which, []).append(float(r["Counter_Value"]))
        out.append(f"{'kernel':96s} {'n':>5s} {'FETCH_KB':>12s} {'WRITE_KB':>12s}")
        for k, v in agg.items():
            fs, ws = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
            out.append(f"{k:96s} {len(fs):5d} {sum(fs)/len(fs):12.1f} {sum(ws)/len(ws):12.1f}")
    os.makedirs("profiles", exist_ok=True)
    if len(sys.argv) >= 5:
        import json
        pm = {k: {"FETCH_SIZE_KB": sum(v.get("FETCH_SIZE", [0])) / len(v.get("FETCH_SIZE", [0])),
                  "WRITE_SIZE_KB": sum(v.get("WRITE_SIZE", [0])) / len(v.get("WRITE_SIZE", [0])),
                  "launches": len(v.get("FETCH_SIZE", [0]))} for k, v in agg.items()}
        json.dump({"tag": tag, "note": "raw rocprofv3 PMC values per dispatch; gfx950 FETCH_SIZE tallies a 128-B "
                                       "request as 64 B (MI355X_MICROARCH.md HBM section): hbm_read = 2 x FETCH_SIZE",
                   "kernels": pm}, open(os.path.join("profiles", f"{tag}_pmc.json"), "w"), indent=1)
    path = os.path.join("profiles", f"{tag}.txt")
    open(path, "w").write("\n".join(out) + "\n")
    print(path)


if __name__ == "__main__":
    main()
