#!/usr/bin/env python3
"""Latency of ONE request through the C ABI (tc_rate_limit: what RateLimiter::rate_limit costs when
called singly) and of a small host-pointer batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

for key_mode in (False, True):
    eng = t.Engine(1_000_000, 1 << 16, key_mode=key_mode)
    keys = [(b"user:%d" % i) if key_mode else int(i).to_bytes(4, "little") for i in range(2000)]
    for k in keys[:200]:
        eng.rate_limit(k, 10, 100, 60, 1, W.T0_NS)
    t0 = time.perf_counter()
    for i, k in enumerate(keys):
        eng.rate_limit(k, 10, 100, 60, 1, W.T0_NS + i)
    dt = time.perf_counter() - t0
    print(f"{'string' if key_mode else 'slot  '} mode: tc_rate_limit {1e6 * dt / len(keys):7.1f} us per call")
    eng.close()

# a handful of requests through the batch call (host arrays, string keys, every result field): one launch (k_small_batch)
eng = t.Engine(1_000_000, 1 << 16, key_mode=True)
for n in (1, 16, 256, 1024):
    arenas = [W.string_keys(np.arange(i * n, (i + 1) * n) % 50_000) for i in range(64)]
    for kb, ko in arenas[:16]:
        eng.rate_limit_batch_keys(kb, ko, max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=W.T0_NS)
    t0 = time.perf_counter()
    for i, (kb, ko) in enumerate(arenas):
        eng.rate_limit_batch_keys(kb, ko, max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=W.T0_NS + i)
    dt = time.perf_counter() - t0
    print(f"string mode: tc_rate_limit_batch_keys, {n:4d} requests {1e6 * dt / len(arenas):7.1f} us per call")
eng.close()
