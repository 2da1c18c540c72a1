#!/usr/bin/env python3
"""The reference-shaped synchronous call from PAGEABLE arrays (what a caller that knows nothing about pinned memory passes):
us per call by batch size; TCGPU_BOUNCE_MAX=0 switches the engine's pinned bounce block off."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
import throttlecrab_amd as t
from throttlecrab_amd import workload as W
n_keys = 400_000
eng = t.Engine(n_keys + (1 << 17), 1 << 17, key_mode=True)
for n in (2048, 4096, 16384, 32768, 65536):
    batches = [bench.abi_shape_stream(n_keys, n, 7 + k, W.T0_NS + 10**9 + k * 10**6) for k in range(4)]
    res = [t.BatchResult(decisions=np.zeros(4 * n, np.int64)) for _ in range(4)]
    def one(i):
        b = batches[i % 4]
        eng.rate_limit_batch_keys(b["key_bytes"], b["key_off"], **{c: b[c] for c in bench.ABI_COLS}, want=("decisions",), out=res[i % 4])
    for i in range(12):
        one(i)
    t0 = time.perf_counter()
    for i in range(40):
        one(i)
    dt = time.perf_counter() - t0
    print(f"pageable sync call, {n:6d} requests: {1e6 * dt / 40:8.1f} us per call  {40 * n / dt / 1e6:7.1f} M decisions/s", flush=True)
eng.close()
