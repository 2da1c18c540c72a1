#!/usr/bin/env python3
"""Latency of small host-pointer batches through the C ABI (what the actor's rate_limit_batch costs when the
queue holds few requests): slot and string mode, decision records out, per-request timestamps."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as O  # noqa: E402  (pack_keys: builds the key arena)
from throttlecrab_amd import workload as W  # noqa: E402

rng = np.random.default_rng(3)
for key_mode in (False, True):
    eng = t.Engine(1_000_000, 1 << 16, key_mode=key_mode)
    for n in (8, 64, 512, 4096, 32768):
        reps = 200 if n <= 4096 else 50
        batches = []
        for r in range(reps + 5):
            ids = rng.integers(0, 200_000, n)
            now = (W.T0_NS + r * 10**6 + rng.integers(0, 10**6, n)).astype(np.int64)
            if key_mode:
                kb, ko = O.pack_keys([b"user:%d" % i for i in ids])
                batches.append((kb, ko, now))
            else:
                batches.append((ids.astype(np.uint32), None, now))
        out = t.BatchResult()

        def one(b):
            if key_mode:
                eng.rate_limit_batch_keys(b[0], b[1], max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=b[2],
                                          want=("decisions",), out=out)
            else:
                eng.rate_limit_batch_slots(b[0], max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=b[2],
                                           want=("decisions",), out=out)
        for b in batches[:5]:
            one(b)
        t0 = time.perf_counter()
        for b in batches[5:]:
            one(b)
        dt = time.perf_counter() - t0
        print(f"{'string' if key_mode else 'slot  '} mode, batch of {n:6d}: {1e6 * dt / reps:8.1f} us per call  ({n * reps / dt / 1e6:7.2f} M requests/s)", flush=True)
    eng.close()
