#!/usr/bin/env python3
"""the synchronous host-pointer SLOT batch (4 B in, 1 B out per request): pageable vs pinned arrays, chunked or in one piece"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import throttlecrab_amd as t
from throttlecrab_amd import workload as W
n, cap = 1 << 20, 10_000_000
eng = t.Engine(cap, n, fixed_params=True)
eng.register_params_uniform(*W.REF_PARAMS)
batches = [W.uniform_slots(cap, n, seed=2, start=i * n) for i in range(4)]
pinned = []
for b in batches:
    h = eng.host_alloc(n, np.uint32); h[:] = b
    pinned.append((h, t.BatchResult(allowed=eng.host_alloc(n, np.uint8))))
for label, use_pinned in (("pageable", False), ("pinned", True)):
    res = [t.BatchResult(allowed=np.zeros(n, np.uint8)) for _ in range(4)]
    def one(i):
        if use_pinned:
            eng.rate_limit_batch_slots(pinned[i % 4][0], registered=True, quantity=1, now_ns=W.T0_NS + i * 10**6, want=("allowed",), out=pinned[i % 4][1])
        else:
            eng.rate_limit_batch_slots(batches[i % 4], registered=True, quantity=1, now_ns=W.T0_NS + i * 10**6, want=("allowed",), out=res[i % 4])
    for i in range(24):
        one(i)
    t0 = time.perf_counter()
    for i in range(40):
        one(100 + i)
    dt = time.perf_counter() - t0
    print(f"sync host slot batch, {label}: {1e6 * dt / 40:.1f} us per 1 Mi call, {40 * n / dt / 1e9:.2f} G decisions/s", flush=True)
eng.close()
