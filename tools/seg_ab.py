#!/usr/bin/env python3
"""Why is a segmented batch slower?  Same stream, same engine shape: plain slot column vs one / two segments,
max_batch = B vs 2 B."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

keys, B, steps = 10_000_000, 1 << 20, 100
db = [torch.from_numpy(W.uniform_slots(keys, B, start=i * B).astype(np.int32)).cuda() for i in range(16)]
for name, mb, mode in (("plain B", B, "plain"), ("plain 2B", 2 * B, "plain"), ("1 segment 2B", 2 * B, "seg1"), ("2 segments 2B", 2 * B, "seg2"),
                       ("1 segment B", B, "seg1")):
    eng = t.Engine(keys, mb, fixed_params=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    outs = [t.BatchResult() for _ in range(8)]
    it = 0
    best = 1e9
    for rep in range(3):
        n = 10 if rep == 0 else steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            d = db[it % 16]
            kw = dict(registered=True, quantity=1, now_ns=W.T0_NS + it * 1_000_000, want=("allowed",), out=outs[it % 8], inputs_ready=True,
                      outputs_idle=True)
            if mode == "plain":
                eng.rate_limit_batch_slots(d, **kw)
            elif mode == "seg1":
                eng.rate_limit_batch_slots(None, segments=[(d, B)], **kw)
            else:
                eng.rate_limit_batch_slots(None, segments=[(d[:B // 2], B // 2), (d[B // 2:], B // 2)], **kw)
            it += 1
        torch.cuda.synchronize()
        if rep:
            best = min(best, (time.perf_counter() - t0) / n)
    print(f"{name:16s} {best * 1e6:7.1f} us/step", flush=True)
    eng.close()
