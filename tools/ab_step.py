#!/usr/bin/env python3
"""A/B of engine knobs on the headline configuration (10 M keys, 1 Mi-request batches, pipelined, decisions only):
one process, one engine per configuration (the knobs are environment variables read at engine creation).

    python tools/ab_step.py [steps] [config-name ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

KNOBS = ("TCGPU_EVAL_LEAN", "TCGPU_STOP_EVENTS", "TCGPU_EVAL_ITEMS", "TCGPU_SORT_ITEMS_PIPED", "TCGPU_AUX_CU_MASK",
         "TCGPU_DEBUG_NO_DECISION_STORE", "TCGPU_AUX_STREAMS", "TCGPU_PIPE_DEPTH", "TCGPU_AUX_PRIORITY", "TCGPU_GATE", "TCGPU_PREFILL")
HALF = ",".join(["ffff"] * 8)
QUARTER = ",".join(["ff"] * 8)
# "_idle": the batches carry TC_B_OUTPUTS_IDLE (a ring of 8 output arrays instead of one)
CONFIGS = {
    "r2": {"TCGPU_EVAL_LEAN": "0", "TCGPU_STOP_EVENTS": "0", "TCGPU_GATE": "0"},
    "idle_events": {"_idle": "1", "TCGPU_GATE": "0"},
    "idle_gate1": {"_idle": "1", "TCGPU_GATE": "1"},
    "idle_gate2": {"_idle": "1", "TCGPU_GATE": "2"},
    "idle_gate1_d6": {"_idle": "1", "TCGPU_GATE": "1", "TCGPU_PIPE_DEPTH": "6"},
    "idle_gate2_d6": {"_idle": "1", "TCGPU_GATE": "2", "TCGPU_PIPE_DEPTH": "6"},
    "idle_gate1_d8": {"_idle": "1", "TCGPU_GATE": "1", "TCGPU_PIPE_DEPTH": "8"},
    "idle_gate1_d6_items1": {"_idle": "1", "TCGPU_GATE": "1", "TCGPU_PIPE_DEPTH": "6", "TCGPU_EVAL_ITEMS": "1"},
    "gate1_d6": {"TCGPU_GATE": "1", "TCGPU_PIPE_DEPTH": "6"},
    "idle_gate1_d6_nostore": {"_idle": "1", "TCGPU_GATE": "1", "TCGPU_PIPE_DEPTH": "6", "TCGPU_DEBUG_NO_DECISION_STORE": "1"},
}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    names = sys.argv[2:] or list(CONFIGS)
    keys, batch, layout_fixed = 10_000_000, 1 << 20, True
    dev = torch.device("cuda:0")
    z = W.Zipf(keys)
    nb = 32
    streams = {"uniform": [torch.from_numpy(W.uniform_slots(keys, batch, start=i * batch).astype(np.int32)).to(dev) for i in range(nb)],
               "zipf": [torch.from_numpy(z.slots(batch, start=i * batch).astype(np.int32)).to(dev) for i in range(nb)]}
    print(f"{'config':22s} {'stream':8s} {'us/step':>8s} {'G/s':>7s}   (best of 3 x {steps} steps)", flush=True)
    for name in names:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in CONFIGS[name].items() if not k.startswith("_")})
        idle = CONFIGS[name].get("_idle") == "1"
        for sname, db in streams.items():
            eng = t.Engine(keys, batch, fixed_params=layout_fixed)
            eng.use_torch_stream()
            eng.register_params_uniform(*W.REF_PARAMS)
            outs = [t.BatchResult() for _ in range(8 if idle else 1)]
            it = 0
            best = 1e9
            for rep in range(4):
                n = 10 if rep == 0 else steps
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    eng.rate_limit_batch_slots(db[it % nb], registered=True, quantity=1, now_ns=W.T0_NS + it * 1_000_000, want=("allowed",),
                                               out=outs[it % len(outs)], inputs_ready=True, outputs_idle=idle)
                    it += 1
                torch.cuda.synchronize()
                if rep:
                    best = min(best, (time.perf_counter() - t0) / n)
            bad = eng.selfcheck()
            print(f"{name:22s} {sname:8s} {best * 1e6:8.1f} {batch / best / 1e9:7.2f}" + (f"   SELFCHECK {bad}" if bad else ""), flush=True)
            eng.close()


if __name__ == "__main__":
    main()
