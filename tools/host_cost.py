#!/usr/bin/env python3
"""The HOST's cost of enqueueing one pipelined batch, by grouping form: small batches (16 Ki requests: the device is done long
before the host), 2 000 calls, nothing waited for in between.  usage: host_cost.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

cap, n = 2_000_000, 16384
for kind in ("uniform", "skewed"):
    eng = t.Engine(cap, n, fixed_params=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    rng = np.random.default_rng(1)
    bat = []
    for i in range(16):
        s = rng.integers(0, cap, n).astype(np.uint32)
        if kind == "skewed":
            m = rng.random(n) < 0.6
            s[m] = (rng.integers(0, 40, int(m.sum())) * 48_611 + 7) % cap
        bat.append(torch.from_numpy(s.astype(np.int32)).cuda())
    outs = [t.BatchResult() for _ in range(8)]
    for i in range(64):
        eng.rate_limit_batch_slots(bat[i % 16], registered=True, quantity=1, now_ns=W.T0_NS + i, want=("allowed",), out=outs[i % 8], inputs_ready=True, outputs_idle=True)
        if i % 4 == 3:
            torch.cuda.synchronize()
    steps = 2000
    t0 = time.perf_counter()
    for i in range(steps):
        eng.rate_limit_batch_slots(bat[i % 16], registered=True, quantity=1, now_ns=W.T0_NS + 10**6 * (i + 1), want=("allowed",), out=outs[i % 8], inputs_ready=True, outputs_idle=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    info = eng.info()
    print(f"{kind}: [{info['grouping_path']}] enqueue {1e6 * (t1 - t0) / steps:.1f} us/batch, with drain {1e6 * (t2 - t0) / steps:.1f} us/batch, hot batches {info['hot_batches']}")
    eng.close()
