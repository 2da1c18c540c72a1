#!/usr/bin/env python3
"""Where the reference-shaped call's time goes: the async ring of pinned sets at one batch size with the
inputs / outputs switched on one by one.  ABI_N (requests per call), ABI_KEYS (table), ABI_CALLS."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

n = int(os.environ.get("ABI_N", 1 << 20))
n_keys = int(os.environ.get("ABI_KEYS", 2_000_000))
calls = int(os.environ.get("ABI_CALLS", 16))
modes = os.environ.get("ABI_MODES", "all,no_out,no_cols,keys_only,sync").split(",")
eng = t.Engine(n_keys + n, n, key_mode=True)
import torch  # noqa: E402
dev = torch.device("cuda:0")
for at in range(0, n_keys, n):
    ids = np.arange(at, min(at + n, n_keys), dtype=np.uint32)
    kb, ko = W.string_keys(ids)
    eng.rate_limit_batch_keys(torch.from_numpy(kb).to(dev), torch.from_numpy(ko.astype(np.int32)).to(dev), max_burst=100, count_per_period=1000,
                              period=3600, quantity=1, now_ns=W.T0_NS, want=("allowed",))
    eng.synchronize()
D = 4
batches = [bench.abi_shape_stream(n_keys, n, 77 + k, W.T0_NS + 10**9 + k * 10**6) for k in range(D)]
pinned = []
for k in range(D):
    pb = {c: eng.host_alloc(batches[k][c].size, batches[k][c].dtype) for c in ("key_bytes", "key_off") + bench.ABI_COLS}
    for c in pb:
        pb[c][:] = batches[k][c]
    pinned.append((pb, t.BatchResult(decisions=eng.host_alloc(4 * n, np.int64), allowed=eng.host_alloc(n, np.uint8))))


def run(mode):
    asy = mode != "sync"
    def one(i):
        pb, r_ = pinned[i % D]
        if asy:
            eng.wait_batches(D - 1)
        kw = {c: pb[c] for c in bench.ABI_COLS}
        want = ("decisions",)
        if mode in ("no_out", "keys_only"):
            want = ("allowed",)
        if mode in ("no_cols", "keys_only"):
            kw = dict(max_burst=100, count_per_period=1000, period=3600, quantity=1, now_ns=W.T0_NS + 10**9)
        eng.rate_limit_batch_keys(pb["key_bytes"], pb["key_off"], **kw, want=want, out=r_, async_=asy)
    for i in range(2 * D):
        one(i)
    eng.wait_batches(0)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
        one(i)
    eng.wait_batches(0)
    eng.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:10s} n={n}: {1e6 * dt / calls:9.1f} us per call  {calls * n / dt / 1e9:6.3f} G decisions/s", flush=True)


for m in modes:
    run(m)
eng.close()


def device_side():
    """the same batches with every array already in device memory: what the kernels cost without PCIe"""
    e2 = t.Engine(n_keys + n, n, key_mode=True)
    for at in range(0, n_keys, n):
        ids = np.arange(at, min(at + n, n_keys), dtype=np.uint32)
        kb, ko = W.string_keys(ids)
        e2.rate_limit_batch_keys(torch.from_numpy(kb).to(dev), torch.from_numpy(ko.astype(np.int32)).to(dev), max_burst=100, count_per_period=1000,
                                 period=3600, quantity=1, now_ns=W.T0_NS, want=("allowed",))
        e2.synchronize()
    dbs = []
    for k in range(D):
        b = batches[k]
        dbs.append({c: torch.from_numpy(b[c] if c != "key_off" else b[c].astype(np.int32)).to(dev) for c in ("key_bytes", "key_off") + bench.ABI_COLS})
    for label, cols, want in (("general+decisions", True, ("decisions",)), ("general+allowed", True, ("allowed",)), ("uniform+decisions", False, ("decisions",))):
        for piped in (False, True):
            outs = [t.BatchResult() for _ in range(D)]
            def one(i):
                d = dbs[i % D]
                kw = {c: d[c] for c in bench.ABI_COLS} if cols else dict(max_burst=100, count_per_period=1000, period=3600, quantity=1, now_ns=W.T0_NS + 10**9)
                e2.rate_limit_batch_keys(d["key_bytes"], d["key_off"], **kw, want=want, out=outs[i % D], inputs_ready=piped)
            for i in range(4):
                one(i)
            e2.synchronize()
            t0 = time.perf_counter()
            for i in range(calls):
                one(i)
            e2.synchronize()
            dt = time.perf_counter() - t0
            e2.profile_enable(True)
            for i in range(8):
                one(i)
            e2.synchronize()
            pr = {k: (round(1e3 * ms / max(1, c), 1), c // 8) for k, (ms, c) in e2.profile_read().items() if c}
            e2.profile_enable(False)
            print(f"device {label:18s} piped={int(piped)}: {1e6 * dt / calls:8.1f} us per call; stages (us per launch, launches per batch): {pr}", flush=True)
    e2.close()


if os.environ.get("ABI_DEVICE", "1") == "1":
    device_side()
