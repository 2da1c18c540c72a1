#!/usr/bin/env python3
"""Per-stage timing (HIP events inside the engine) for the uniform and zipf
streams; quick iteration tool.  usage: stage_bench.py [steps] [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
piped = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
keys = 10_000_000

KINDS = os.environ.get("TC_STAGE_KINDS", "uniform,zipf").split(",")
MODES = os.environ.get("TC_STAGE_MODES")  # e.g. "bits,dec": only these output modes

for kind in KINDS:
    z = W.Zipf(keys) if kind == "zipf" else None
    hb = [(z.slots(batch, start=i * batch) if z else W.uniform_slots(keys, batch, start=i * batch)) for i in range(steps + 3)]
    db = [torch.from_numpy(b.astype(np.int32)).cuda() for b in hb]
    for want in (("allowed",), ("allowed", "grouped"), t.Engine.ALL_FIELDS, t.Engine.RECORD_FIELDS, t.Engine.DECISION_FIELDS,
                 t.Engine.DECISION_FIELDS + ("grouped",)):
        grouped = "grouped" in want
        label = ('dec' if 'decisions' in want else ('rec' if 'result4' in want else ('full' if len(want) > 2 else 'bits'))) + ('+g' if grouped else '')
        want = tuple(w for w in want if w != "grouped")
        if MODES and label not in MODES.split(","):
            continue
        eng = t.Engine(keys, batch)
        if os.environ.get("TC_OWN_STREAM") != "1":
            eng.use_torch_stream()
        eng.register_params_uniform(*W.REF_PARAMS)
        out = t.BatchResult()
        for i in range(3):
            eng.rate_limit_batch_slots(db[i], registered=True, quantity=1, now_ns=W.T0_NS + i * 10**6, want=want, out=out, inputs_ready=piped, grouped=grouped)
        eng.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.rate_limit_batch_slots(db[3 + i], registered=True, quantity=1, now_ns=W.T0_NS + (3 + i) * 10**6, want=want, out=out, inputs_ready=piped, grouped=grouped)
        t_issue = time.perf_counter() - t0
        eng.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.profile_enable(True)
        for i in range(steps):
            eng.rate_limit_batch_slots(db[3 + i], registered=True, quantity=1, now_ns=W.T0_NS + (30 + i) * 10**6, want=want, out=out, inputs_ready=piped, grouped=grouped)
        prof = eng.profile_read()
        st = {k: round(1e3 * v[0] / max(1, v[1]), 1) for k, v in prof.items() if v[1]}
        print(f"piped={int(piped)} {kind:8s} {label:6s} {steps * batch / dt / 1e9:7.2f} G/s  "
              f"{1e6 * dt / steps:7.1f} us/batch  host-issue {1e6 * t_issue / steps:6.1f} us/batch  stages(us)={st}", flush=True)
        eng.close()
