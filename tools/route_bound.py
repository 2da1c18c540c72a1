#!/usr/bin/env python3
"""Where does the time of the multi-GPU loop go on one GPU?  (a) evaluation of pre-routed batches only, (b) router
only, (c) both as bench.py --gpus N issues them (router TC_ROUTE_AHEAD, LOOKAHEAD batches ahead, counts polled).
usage: route_bound.py [world=1] [steps=100] [lookahead=4]"""
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

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B, K = 1 << 20, 10_000_000
G = world * B
LOOKAHEAD = int(sys.argv[3]) if len(sys.argv) > 3 else 4
RING = 2 * LOOKAHEAD
with torch.cuda.stream(torch.cuda.Stream()):
    eng = t.Engine(K, min(G, 4 * B), fixed_params=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    d_global = [torch.from_numpy(W.uniform_slots(world * K, G, seed=2, start=i * G).astype(np.int32)).cuda() for i in range(8)]
    ring = [(torch.empty(G, dtype=torch.int32, device="cuda"), None, torch.zeros(world, dtype=torch.int32, device="cuda")) for _ in range(RING)]
    host = [eng.host_alloc(world + 1, np.uint32) for _ in range(RING)]
    for h in host:
        h[:] = 0
    out = t.BatchResult()
    tag = [0]

    def route(i):
        tag[0] += 1
        host[i % RING][world] = 0
        eng.route_batch(d_global[i % 8], world, only=0, out=ring[i % RING], ahead=True, host_counts=host[i % RING], tag=tag[0])
        return tag[0]

    def wait(i, tg):
        while int(host[i % RING][world]) != tg:
            pass
        return int(host[i % RING][0])

    def evaluate(i, mine):
        eng.rate_limit_batch_slots(ring[i % RING][0][:mine], registered=True, quantity=1, now_ns=W.T0_NS + i * 10**6, want=("allowed",), out=out, inputs_ready=True)

    # (a) evaluation only: the ring routed once
    tags = [route(i) for i in range(RING)]
    mine = [wait(i, tags[i]) for i in range(RING)]
    for i in range(20):
        evaluate(i, mine[i % RING])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        evaluate(i, mine[i % RING])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"(a) evaluation only: enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, drained {1e6 * (t2 - t0) / steps:.1f} us/step")
    # (b) router only
    t0 = time.perf_counter()
    for i in range(steps):
        tg = route(i)
    t1 = time.perf_counter()
    wait(steps - 1, tg)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"(b) router only:     enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, drained {1e6 * (t2 - t0) / steps:.1f} us/step")
    # (c) both
    tags = {}
    for j in range(LOOKAHEAD):
        tags[j] = route(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    polled = 0.0
    for i in range(steps):
        tags[i + LOOKAHEAD] = route(i + LOOKAHEAD)
        p0 = time.perf_counter()
        m = wait(i, tags.pop(i))
        polled += time.perf_counter() - p0
        evaluate(i, m)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"(c) both:            enqueue {1e6 * (t1 - t0) / steps:.1f} us/step (of which polling {1e6 * polled / steps:.1f}), drained {1e6 * (t2 - t0) / steps:.1f} us/step")
    print("selfcheck", eng.selfcheck())
    eng.close()
