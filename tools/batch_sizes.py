#!/usr/bin/env python3
"""Decisions/s against the batch size: 10 M keys (fixed layout, one registered plan), uniform stream, decisions only -- device
batches pipelined (TC_B_INPUTS_READY + TC_B_OUTPUTS_IDLE, a ring of 8 result arrays) and in order on one stream (the ABI's
default), host batches through the synchronous call (pinned arrays, copies included).  One engine per mode; max_batch 1 Mi.
The reference's actor drains whatever its channel holds (actor.rs:217-236): batches of 10^2 .. 10^5 requests are its shape."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

N_KEYS, MAXB = 10_000_000, 1 << 20
SIZES = [int(x) for x in os.environ["BS_SIZES"].split(",")] if os.environ.get("BS_SIZES") else [256, 1024, 4096, 16384, 65536, 262144, 1 << 20]
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
print(f"{'batch':>9} | {'pipelined us':>12} {'G/s':>7} | {'in order us':>11} {'G/s':>7} | {'host sync us':>12} {'G/s':>7}")
with torch.cuda.stream(stream):
    for n in SIZES:
        steps = max(50, min(2000, (1 << 24) // n))
        row = []
        for mode in (os.environ["BS_MODES"].split(",") if os.environ.get("BS_MODES") else ("piped", "inorder", "host")):
            if os.environ.get("BS_PRE"):  # (GPU work on torch's stream BEFORE the engine creates its streams)
                torch.zeros(1 << 20, device=dev).sum().item()
            eng = t.Engine(N_KEYS, MAXB, fixed_params=True)
            eng.use_torch_stream()
            eng.register_params_uniform(*W.REF_PARAMS)
            host = [W.uniform_slots(N_KEYS, n, seed=2, start=i * n) for i in range(8)]
            if mode == "host":
                h_slots = [eng.host_alloc(n, np.uint32) for _ in range(8)]
                for a, b in zip(h_slots, host):
                    a[:] = b
                res = t.BatchResult(allowed=eng.host_alloc(n, np.uint8))
                run = lambda i: eng.rate_limit_batch_slots(h_slots[i % 8], registered=True, now_ns=W.T0_NS + i * 1000, want=("allowed",), out=res)
                steps_m = max(20, steps // 4)
            else:
                d_slots = [torch.from_numpy(h.astype(np.int32)).to(dev) for h in host]
                outs = [t.BatchResult(allowed=torch.empty(n, dtype=torch.uint8, device=dev)) for _ in range(8)]
                piped = mode == "piped"
                run = lambda i: eng.rate_limit_batch_slots(d_slots[i % 8], registered=True, now_ns=W.T0_NS + i * 1000, want=("allowed",),
                                                           out=outs[i % 8], inputs_ready=piped, outputs_idle=piped)
                steps_m = steps
            for i in range(10):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps_m):
                run(10 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps_m
            row.append((1e6 * dt, n / dt / 1e9))
            if os.environ.get("BS_STAGES") and mode != "host":
                eng.profile_enable(True)
                for i in range(20):
                    run(i)
                torch.cuda.synchronize()
                pr = eng.profile_read()
                eng.profile_enable(False)
                print(f"   {mode} n={n}: " + ", ".join(f"{k} {1e3 * ms / calls:.1f} us x{calls / 20:.1f}" for k, (ms, calls) in pr.items() if calls))
            assert eng.selfcheck() == 0
            eng.close()
        while len(row) < 3:
            row.append((0.0, 0.0))
        print(f"{n:>9} | {row[0][0]:>12.1f} {row[0][1]:>7.3f} | {row[1][0]:>11.1f} {row[1][1]:>7.3f} | {row[2][0]:>12.1f} {row[2][1]:>7.3f}", flush=True)
