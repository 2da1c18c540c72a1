#!/usr/bin/env python3
"""us per 1 Mi batch on a FRESH engine per configuration (no engine ever sees another mode's batches): stream x layout x
pipelined / in order, 10 M keys, one registered plan, decisions only; 100 timed batches after 12 of warm-up."""
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

N, B = 10_000_000, 1 << 20
dev = torch.device("cuda:0")
streams = {"uniform": [W.uniform_slots(N, B, seed=2, start=i * B) for i in range(8)]}
if os.environ.get("FM_ONLY") != "uniform":
    z = W.Zipf(N)
    streams["zipf"] = [z.slots(B, seed=3, start=i * B) for i in range(8)]
with torch.cuda.stream(torch.cuda.Stream()):
    for sname, host in streams.items():
        d = [torch.from_numpy(h.astype(np.int32)).to(dev) for h in host]
        for fixed in (True, False):
            for piped in (True, False):
                eng = t.Engine(N, B, fixed_params=fixed)
                eng.use_torch_stream()
                eng.register_params_uniform(*W.REF_PARAMS)
                outs = [t.BatchResult(allowed=torch.empty(B, dtype=torch.uint8, device=dev)) for _ in range(8)]
                run = lambda i: eng.rate_limit_batch_slots(d[i % 8], registered=True, now_ns=W.T0_NS + i * int(os.environ.get("FM_DT", "1000000")), want=("allowed",),
                                                           out=outs[i % 8], inputs_ready=piped, outputs_idle=piped)
                for i in range(12):
                    run(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(int(os.environ.get("FM_STEPS", "100"))):
                    run(12 + i)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / int(os.environ.get("FM_STEPS", "100"))
                eng.profile_enable(True)
                for i in range(8):
                    run(112 + i)
                torch.cuda.synchronize()
                pr = eng.profile_read()
                st = ", ".join(f"{k} {1e3 * ms / calls:.1f}x{calls / 8:.0f}" for k, (ms, calls) in pr.items() if calls)
                print(f"{sname:8s} {'fixed' if fixed else 'wide ':5s} {'pipelined' if piped else 'in order ':9s} {1e6 * dt:7.1f} us  {B / dt / 1e9:6.2f} G/s   [{st}]", flush=True)
                assert eng.selfcheck() == 0
                eng.close()
