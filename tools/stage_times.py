#!/usr/bin/env python3
"""Per-stage kernel times (tc_profile_*: HIP events on the stream each kernel runs on) of the uniform stream,
pipelined and in order, for whatever grouping path the TCGPU_* environment selects.
usage: stage_times.py [steps=100] [layout=fixed] [stream=uniform|zipf]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fixed = (sys.argv[2] if len(sys.argv) > 2 else "fixed") == "fixed"
stream = sys.argv[3] if len(sys.argv) > 3 else "uniform"
eng = t.Engine(10_000_000, 1 << 20, fixed_params=fixed)
eng.use_torch_stream()
eng.register_params_uniform(*W.REF_PARAMS)
if stream == "uniform":
    gen = lambda start: W.uniform_slots(10_000_000, 1 << 20, start=start)
else:
    z = W.Zipf(10_000_000)
    gen = lambda start: z.slots(1 << 20, start=start)
bat = [torch.from_numpy(gen(i << 20).astype(np.int32)).cuda() for i in range(16)]
out = t.BatchResult()
it = 0
for piped in (True, False):
    for i in range(20):
        eng.rate_limit_batch_slots(bat[it % 16], registered=True, quantity=1, now_ns=W.T0_NS + it * 10**6, want=("allowed",), out=out, inputs_ready=piped)
        it += 1
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for i in range(steps):
        eng.rate_limit_batch_slots(bat[it % 16], registered=True, quantity=1, now_ns=W.T0_NS + it * 10**6, want=("allowed",), out=out, inputs_ready=piped)
        it += 1
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    line = ", ".join(f"{k} {1e3 * ms / calls:.1f}us x{calls / steps:.0f}" for k, (ms, calls) in prof.items() if calls)
    print(f"piped={piped}: {line}  | sum {1e3 * sum(ms for ms, c in prof.values()) / steps:.1f} us/batch")
print("selfcheck", eng.selfcheck())
eng.close()
