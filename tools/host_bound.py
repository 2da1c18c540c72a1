#!/usr/bin/env python3
"""Is the pipelined stream bound by the HOST's enqueue rate?  Times the call loop alone (no synchronisation) and
the loop + drain.  usage: host_bound.py [steps=200] [layout=fixed] [stream=uniform|zipf]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fixed = (sys.argv[2] if len(sys.argv) > 2 else "fixed") == "fixed"
eng = t.Engine(10_000_000, 1 << 20, fixed_params=fixed)
eng.use_torch_stream()
eng.register_params_uniform(*W.REF_PARAMS)
stream = sys.argv[3] if len(sys.argv) > 3 else "uniform"
if stream == "zipf":
    z = W.Zipf(10_000_000)
    bat = [torch.from_numpy(z.slots(1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(16)]
else:
    bat = [torch.from_numpy(W.uniform_slots(10_000_000, 1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(16)]
out = t.BatchResult()
for piped in (True, False):
    for i in range(40):
        eng.rate_limit_batch_slots(bat[i % 16], registered=True, quantity=1, now_ns=W.T0_NS + i, want=("allowed",), out=out, inputs_ready=piped)
        if i % 4 == 3:
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.rate_limit_batch_slots(bat[i % 16], registered=True, quantity=1, now_ns=W.T0_NS + 10**6 * (i + 1), want=("allowed",), out=out,
                                   inputs_ready=piped)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    info = eng.info()
    print(f"{stream} piped={piped} [{info['grouping_path']}, hot slots {info['hot_slots']}, hot batches {info['hot_batches']}]: enqueue loop {1e6 * (t1 - t0) / steps:.1f} us/batch, with drain {1e6 * (t2 - t0) / steps:.1f} us/batch")
eng.close()
