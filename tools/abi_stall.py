#!/usr/bin/env python3
"""host-side time of every call of the async ring (where does a call block?)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
import throttlecrab_amd as t
from throttlecrab_amd import workload as W
import torch
n = int(os.environ.get("ABI_N", 65536)); n_keys = int(os.environ.get("ABI_KEYS", 2_000_000)); D = 4
eng = t.Engine(n_keys + n, max(n, 1 << 16), key_mode=True)
dev = torch.device("cuda:0")
for at in range(0, n_keys, 1 << 20):
    ids = np.arange(at, min(at + (1 << 20), n_keys), dtype=np.uint32)
    kb, ko = W.string_keys(ids)
    for a0 in range(0, len(ids), eng.max_batch):
        pass
eng.close()
eng = t.Engine(n_keys + (1 << 20), 1 << 20, key_mode=True)
for at in range(0, n_keys, 1 << 20):
    ids = np.arange(at, min(at + (1 << 20), n_keys), dtype=np.uint32)
    kb, ko = W.string_keys(ids)
    eng.rate_limit_batch_keys(torch.from_numpy(kb).to(dev), torch.from_numpy(ko.astype(np.int32)).to(dev), max_burst=100, count_per_period=1000, period=3600, quantity=1, now_ns=W.T0_NS, want=("allowed",))
    eng.synchronize()
if os.environ.get("ABI_POLICY"):
    eng.set_sweep_policy("adaptive", created_ns=W.T0_NS, min_interval_ns=5 * 10**9, max_interval_ns=300 * 10**9, max_operations=int(os.environ["ABI_POLICY"]))
batches = [bench.abi_shape_stream(n_keys, n, 77 + k, W.T0_NS + 10**9 + k * 10**6) for k in range(D)]
pinned = []
for k in range(D):
    pb = {c: eng.host_alloc(batches[k][c].size, batches[k][c].dtype) for c in ("key_bytes", "key_off") + bench.ABI_COLS}
    for c in pb:
        pb[c][:] = batches[k][c]
    pinned.append((pb, t.BatchResult(decisions=eng.host_alloc(4 * n, np.int64))))
tw, tc = [], []
for i in range(int(os.environ.get("ABI_CALLS", 40))):
    pb, r_ = pinned[i % D]
    a = time.perf_counter()
    eng.wait_batches(D - 1)
    b = time.perf_counter()
    eng.rate_limit_batch_keys(pb["key_bytes"], pb["key_off"], **{c: pb[c] for c in bench.ABI_COLS}, want=("decisions",), out=r_, async_=True)
    c = time.perf_counter()
    tw.append(1e6 * (b - a)); tc.append(1e6 * (c - b))
eng.wait_batches(0)
big = [(i, round(x)) for i, x in enumerate(tc) if x > 1000]
print("calls over 1 ms (index, us):", big)
print("median call us", round(float(np.median(tc))), "median wait us", round(float(np.median(tw))), "mean total us", round(float(np.mean(tc) + np.mean(tw))))
tail = slice(len(tc) // 2, None)
if os.environ.get("ABI_POLICY"):
    print(eng.sweep_stats())
print("second half: mean call", round(float(np.mean(tc[tail]))), "mean wait", round(float(np.mean(tw[tail]))))
eng.close()
