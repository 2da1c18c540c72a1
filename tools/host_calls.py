#!/usr/bin/env python3
"""Host time of every call of a SHORT timed region (the driver's form: 20 batches between two device synchronisations),
uniform and Zipf streams: is the region bound by the thread that enqueues it?
Then the driver's form itself: a FRESH engine, 5 warmup batches, a device synchronisation, 20 timed batches -- with the hot
list made on the engine's own thread (default) and on the caller's (TCGPU_HOT_THREAD=0).
usage: host_calls.py [steps=20] [reps=6] [streams=uniform,zipf] [fresh: only the driver's form, in a process that has done nothing else]"""
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
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
streams = (sys.argv[3] if len(sys.argv) > 3 else "uniform,zipf").split(",")
fresh_only = len(sys.argv) > 4 and sys.argv[4] == "fresh"
for stream in ([] if fresh_only else streams):
    eng = t.Engine(10_000_000, 1 << 20, fixed_params=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    if stream == "zipf":
        z = W.Zipf(10_000_000)
        bat = [torch.from_numpy(z.slots(1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(16)]
    else:
        bat = [torch.from_numpy(W.uniform_slots(10_000_000, 1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(16)]
    outs = [t.BatchResult() for _ in range(8)]
    it = 0

    def call():
        global it
        eng.rate_limit_batch_slots(bat[it % 16], registered=True, quantity=1, now_ns=W.T0_NS + it * 1_000_000, want=("allowed",),
                                   out=outs[it % 8], inputs_ready=True, outputs_idle=True)
        it += 1

    for k in range(60):
        call()
        if k % 5 == 4:
            torch.cuda.synchronize()
    info = eng.info()
    print(f"== {stream} [{info['grouping_path']}, hot slots {info.get('hot_slots')}]")
    for rep in range(reps):
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for k in range(steps):
            call()
            ts.append(time.perf_counter())
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        per = " ".join(f"{1e6 * (ts[k + 1] - ts[k]):5.1f}" for k in range(steps))
        print(f"rep {rep}: enqueue {1e6 * (ts[-1] - ts[0]):7.1f} us, region {1e6 * (t3 - ts[0]):7.1f} us = {1e6 * (t3 - ts[0]) / steps:5.1f} us/step | {per}")
    eng.close()

# the driver's form: fresh engine, 5 + 20
for stream in streams:
    if stream == "zipf":
        z = W.Zipf(10_000_000)
        bat = [torch.from_numpy(z.slots(1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(25)]
    else:
        bat = [torch.from_numpy(W.uniform_slots(10_000_000, 1 << 20, start=i << 20).astype(np.int32)).cuda() for i in range(25)]
    outs = [t.BatchResult(allowed=torch.empty(1 << 20, dtype=torch.uint8, device="cuda")) for _ in range(8)]
    for threaded in ("1", "0", "1", "0", "1", "0"):
        os.environ["TCGPU_HOT_THREAD"] = threaded
        eng = t.Engine(10_000_000, 1 << 20, fixed_params=True)
        eng.use_torch_stream()
        eng.register_params_uniform(*W.REF_PARAMS)
        paths = []

        def call(i):
            eng.rate_limit_batch_slots(bat[i], registered=True, quantity=1, now_ns=W.T0_NS + i * 1_000_000, want=("allowed",),
                                       out=outs[i % 8], inputs_ready=True, outputs_idle=True)

        for i in range(5):
            call(i)
        torch.cuda.synchronize()
        ts = [time.perf_counter()]
        for k in range(steps):
            call(5 + k)
            ts.append(time.perf_counter())
            paths.append(eng.last_grouping_path_code() if hasattr(eng, "last_grouping_path_code") else 0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        per = " ".join(f"{1e6 * (ts[k + 1] - ts[k]):5.1f}" for k in range(steps))
        print(f"fresh {stream} hot_thread={threaded}: enqueue {1e6 * (ts[-1] - ts[0]):7.1f} us, region {1e6 * (t3 - ts[0]):7.1f} us = {1e6 * (t3 - ts[0]) / steps:5.1f} us/step | {per}")
        print("      info: " + str({k: v for k, v in eng.info().items() if k in ("grouping_path", "hot_slots", "hot_batches")}))
        eng.close()
