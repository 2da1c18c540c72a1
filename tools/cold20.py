#!/usr/bin/env python3
"""Where do the first microseconds of a SHORT timed region go?  The driver times 20 batches between two device
synchronisations; the pipelined stream needs ~46 us per batch in steady state but ~53 in such a region.  Per repetition:
host time to enqueue the 20 calls, time until the engine's stream has drained (tc_synchronize), time until the device
has (torch.cuda.synchronize), and completion times of the 1st / 2nd / 5th / 20th evaluation (events on the engine's stream).
usage: cold20.py [steps=20] [reps=8]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = t.Engine(10_000_000, 1 << 20, fixed_params=True)
eng.use_torch_stream()
eng.register_params_uniform(*W.REF_PARAMS)
bat = [torch.from_numpy(W.uniform_slots(10_000_000, 1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(16)]
outs = [t.BatchResult() for _ in range(8)]
it = 0


def call():
    global it
    eng.rate_limit_batch_slots(bat[it % 16], registered=True, quantity=1, now_ns=W.T0_NS + it * 1_000_000, want=("allowed",),
                               out=outs[it % 8], inputs_ready=True, outputs_idle=True)
    it += 1


for _ in range(10):
    call()
torch.cuda.synchronize()
marks = [0, 1, 4, steps - 1]
print(f"{'rep':>3s} {'host_enq':>9s} {'1st':>7s} {'2nd':>7s} {'5th':>7s} {'last':>7s} {'drained':>8s} {'dev_sync':>9s} {'us/step':>8s}")
for rep in range(reps):
    evs = {k: torch.cuda.Event(enable_timing=True) for k in marks}
    ev0 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(steps):
        call()
        if k in evs:
            evs[k].record()
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    tm = [ev0.elapsed_time(evs[k]) * 1e3 for k in marks]
    print(f"{rep:3d} {1e6*(t1-t0):9.1f} {tm[0]:7.1f} {tm[1]:7.1f} {tm[2]:7.1f} {tm[3]:7.1f} {1e6*(t2-t0):8.1f} {1e6*(t3-t0):9.1f} {1e6*(t3-t0)/steps:8.1f}")
# the same without the event records in the loop (they are marker packets on the engine's stream)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"plain {rep}: host_enq {1e6*(t1-t0):.1f}  total {1e6*(t3-t0):.1f}  us/step {1e6*(t3-t0)/steps:.1f}")
eng.close()
