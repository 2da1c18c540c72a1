#!/usr/bin/env python3
"""bench.py's string-key section alone (configs[4] at its stated configuration, both key sets)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

a = bench.parse()
if os.environ.get("KEYS_NO_VERIFY"):
    a.no_verify = True
torch.cuda.set_device(0)
r = bench.keys_bench(a, torch.device("cuda:0"))
print(json.dumps({k: ({kk: v[kk] for kk in ("value", "ms_per_step", "steps") if kk in v} if isinstance(v, dict) else v) for k, v in r.items()}))
if os.environ.get("KEYS_DETAIL"):
    for k, v in r.items():
        if isinstance(v, dict):
            print(k, {st: (round(1e3 * x["avg_ms"], 1), round(x["launches_per_batch"], 2)) for st, x in v["detail"]["stages"]["pipelined"].items()}, v.get("grouping_path"))
