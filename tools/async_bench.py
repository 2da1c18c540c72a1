#!/usr/bin/env python3
"""TC_B_ASYNC host-buffer path: PCIe-inclusive rate with a ring of K pinned buffer sets.
usage: async_bench.py [batches] [K] [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 48
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
keys = 10_000_000
eng = t.Engine(keys, B)
if os.environ.get("TC_OWN_STREAM") != "1":
    eng.use_torch_stream()
eng.register_params_uniform(*W.REF_PARAMS)
ring = [(eng.host_alloc(B, np.uint32), t.BatchResult(allowed=eng.host_alloc(B, np.uint8))) for _ in range(K)]
for i in range(K):
    ring[i][0][:] = W.uniform_slots(keys, B, start=i * B)


def feed(i, log=None):
    t0 = time.perf_counter()
    if i >= K:
        eng.wait_batches(K - 1)
    t1 = time.perf_counter()
    sl, ob = ring[i % K]
    eng.rate_limit_batch_slots(sl, registered=True, quantity=1, now_ns=W.T0_NS + i * 10**6, want=("allowed",), out=ob, async_=True)
    if log is not None:
        log.append((1e6 * (t1 - t0), 1e6 * (time.perf_counter() - t1)))


for i in range(2 * K):
    feed(i)
eng.wait_batches(0)
log = []
t0 = time.perf_counter()
for i in range(NB):
    feed(i, log)
eng.wait_batches(0)
dt = time.perf_counter() - t0
w = np.array(log)
print(f"async K={K}: {NB * B / dt / 1e9:.2f} G/s  {1e6 * dt / NB:.1f} us/batch   host: wait {w[:, 0].mean():.1f} us, submit {w[:, 1].mean():.1f} us (max {w[:, 1].max():.1f})")
eng.close()
