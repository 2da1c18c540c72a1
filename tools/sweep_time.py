#!/usr/bin/env python3
"""tc_sweep_expired alone: 10 M slots, a quarter of them live, a quarter expired -- fixed and wide layout, synchronous call timed
from the host (10 repetitions; the state is rewritten before each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import throttlecrab_amd as t
from throttlecrab_amd import workload as W

N, B = 10_000_000, 1 << 20
dev = torch.device("cuda:0")
for fixed in (True, False):
    eng = t.Engine(N, B, fixed_params=fixed)
    eng.use_torch_stream()
    eng.register_params_uniform(10, 100, 60)
    res = t.BatchResult()
    rng = np.random.default_rng(5)
    times = []
    for rep in range(10):
        t0 = W.T0_NS + rep * 10**12
        for k in range(5):  # 5 Mi requests over 10 M slots: ~40 % of the slots written
            sl = torch.from_numpy(rng.integers(0, N, B).astype(np.int32)).to(dev)
            eng.rate_limit_batch_slots(sl, registered=True, now_ns=t0 + (k % 2) * 200 * 10**9, want=("allowed",), out=res)
        torch.cuda.synchronize()
        a = time.perf_counter()
        removed = eng.sweep_expired(t0 + 150 * 10**9)
        times.append(1e6 * (time.perf_counter() - a))
        eng.sweep_expired(t0 + 10**11 * 9)
    print("fixed" if fixed else "wide ", "sweep of 10 M slots: median %.1f us (min %.1f), removed %d" % (float(np.median(times)), min(times), removed))
    eng.close()
