#!/usr/bin/env python3
"""General-batch timing (per-request timestamps): uniform and zipf slot streams, 10 M keys, 1 Mi requests."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

steps, batch, keys = 10, 1 << 20, 10_000_000
for kind in ("uniform", "zipf"):
    z = W.Zipf(keys) if kind == "zipf" else None
    hb = [(z.slots(batch, start=i * batch) if z else W.uniform_slots(keys, batch, start=i * batch)) for i in range(steps + 2)]
    db = [torch.from_numpy(b.astype(np.int32)).cuda() for b in hb]
    # request i of batch b is stamped b ms + i ns (strictly increasing inside the batch)
    nows = [torch.arange(batch, dtype=torch.int64, device="cuda") + (W.T0_NS + b * 10**6) for b in range(steps + 2)]
    for want in (("allowed",), t.Engine.RECORD_FIELDS):
        eng = t.Engine(keys, batch)
        eng.use_torch_stream()
        eng.register_params_uniform(*W.REF_PARAMS)
        out = t.BatchResult()
        for i in range(2):
            eng.rate_limit_batch_slots(db[i], registered=True, quantity=1, now_ns=nows[i], want=want, out=out, inputs_ready=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2, steps + 2):
            eng.rate_limit_batch_slots(db[i], registered=True, quantity=1, now_ns=nows[i], want=want, out=out, inputs_ready=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = eng.counters()
        print(f"general {kind:8s} {'rec' if len(want) > 1 else 'bits':5s} {steps * batch / dt / 1e9:7.2f} G/s  {1e6 * dt / steps:8.1f} us/batch  "
              f"allowed {c['allowed'] / max(1, c['total']):.3f}", flush=True)
        eng.close()
