#!/usr/bin/env python3
"""Per-stage timing of the string-key path (configs[4] shape). usage: stage_keys.py [steps] [batch] [keys]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
keys = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
pre = 5
cap = keys + (steps + pre) * B // 5 + B
eng = t.Engine(cap, B, key_mode=True)
eng.use_torch_stream()
rng = np.random.default_rng(5)
seen, batches = 0, []
for s in range(pre + 2 * steps):
    n_new = B if s < pre else B // 5
    new = np.arange(seen, seen + n_new, dtype=np.int64)
    old = rng.integers(0, max(seen, 1), B - n_new)
    ids = np.concatenate([old, new])
    rng.shuffle(ids)
    seen += n_new
    kb, ko = W.string_keys(ids)
    batches.append((torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda()))
out = t.BatchResult()


def one(s):
    kb, ko = batches[s]
    eng.rate_limit_batch_keys(kb, ko, max_burst=10, count_per_period=100, period=60, quantity=1,
                              now_ns=W.T0_NS + s * 10**9, want=("allowed",), out=out, inputs_ready=True)


for s in range(pre):
    one(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(pre, pre + steps):
    one(s)
    if (s - pre) % 4 == 3:
        eng.sweep_expired_async(W.T0_NS + s * 10**9)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"keys: {steps * B / dt / 1e9:.2f} G/s  {1e6 * dt / steps:.1f} us/batch (sweeps included)")
eng.profile_enable(True)
for s in range(pre + steps, pre + 2 * steps):
    one(s)
prof = eng.profile_read()
print({k: round(1e3 * v[0] / max(1, v[1]), 1) for k, v in prof.items() if v[1]})
eng.close()
