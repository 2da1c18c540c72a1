#!/usr/bin/env python3
"""A/B of engine knobs on the headline configuration (10 M keys, 1 Mi-request batches, pipelined, decisions only).
EVERY configuration runs in a process of its own: the knobs are environment variables read at engine creation, and which
hardware queue a HIP stream lands on depends on everything the process created before (a configuration with four
grouping streams left every later engine of the same process at half speed: r03_v33).

    python tools/ab_step.py [steps] [config-name ...]
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

KNOBS = ("TCGPU_EVAL_LEAN", "TCGPU_STOP_EVENTS", "TCGPU_EVAL_ITEMS", "TCGPU_SORT_ITEMS_PIPED", "TCGPU_DEBUG_NO_DECISION_STORE",
         "TCGPU_AUX_STREAMS", "TCGPU_PIPE_DEPTH", "TCGPU_AUX_PRIORITY", "TCGPU_PREFILL", "TCGPU_RANGE", "TCGPU_GENERAL_EARLIER")
# "_idle": the batches carry TC_B_OUTPUTS_IDLE (a ring of 8 output arrays instead of one)
CONFIGS = {
    "default": {"_idle": "1"},
    "default_again": {"_idle": "1"},
    "range0": {"_idle": "1", "TCGPU_RANGE": "0"},          # three LSD passes (round 3's grouping)
    "range0_again": {"_idle": "1", "TCGPU_RANGE": "0"},
    "items4": {"_idle": "1", "TCGPU_EVAL_ITEMS": "4"},        # lean evaluation, 1024 blocks of 1024 positions
    "items4_again": {"_idle": "1", "TCGPU_EVAL_ITEMS": "4"},
    "no_earlier": {"_idle": "1", "TCGPU_GENERAL_EARLIER": "0"},   # (AB_GENERAL=1) k_eval_general without the earlier-state rule
    "no_earlier_again": {"_idle": "1", "TCGPU_GENERAL_EARLIER": "0"},
    "range_aux2": {"_idle": "1", "TCGPU_AUX_STREAMS": "2"},
    "range_sort8": {"_idle": "1", "TCGPU_SORT_ITEMS_PIPED": "8"},
    "range_sort32": {"_idle": "1", "TCGPU_SORT_ITEMS_PIPED": "32"},
    "range_depth4": {"_idle": "1", "TCGPU_PIPE_DEPTH": "4"},
    "aux2": {"_idle": "1", "TCGPU_AUX_STREAMS": "2"},
    "aux4": {"_idle": "1", "TCGPU_AUX_STREAMS": "4"},
    "depth4": {"_idle": "1", "TCGPU_PIPE_DEPTH": "4"},
    "depth8": {"_idle": "1", "TCGPU_PIPE_DEPTH": "8"},
    "sort8": {"_idle": "1", "TCGPU_SORT_ITEMS_PIPED": "8"},
    "sort32": {"_idle": "1", "TCGPU_SORT_ITEMS_PIPED": "32"},
    "items1": {"_idle": "1", "TCGPU_EVAL_ITEMS": "1"},
    "prio": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1"},
    "prio_again": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1"},
    "prio_low": {"_idle": "1", "TCGPU_AUX_PRIORITY": "1"},
    "prio_sort32": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1", "TCGPU_SORT_ITEMS_PIPED": "32"},
    "prio_sort8": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1", "TCGPU_SORT_ITEMS_PIPED": "8"},
    "prio_items1": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1", "TCGPU_EVAL_ITEMS": "1"},
    "prio_aux2": {"_idle": "1", "TCGPU_AUX_PRIORITY": "-1", "TCGPU_AUX_STREAMS": "2"},
    "no_idle": {},
    "r2": {"TCGPU_EVAL_LEAN": "0", "TCGPU_STOP_EVENTS": "0"},
    "nostore": {"_idle": "1", "TCGPU_DEBUG_NO_DECISION_STORE": "1"},
}
CACHE = "/tmp/ab_step_streams.npz"


def child(name, steps):
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W
    keys, batch = 10_000_000, 1 << 20
    idle = CONFIGS[name].get("_idle") == "1"
    z = np.load(CACHE)
    dev = torch.device("cuda:0")
    for sname in ("uniform", "zipf"):
        db = [torch.from_numpy(b).to(dev) for b in z[sname]]
        nb = len(db)
        eng = t.Engine(keys, batch, fixed_params=True)
        eng.use_torch_stream()
        eng.register_params_uniform(*W.REF_PARAMS)
        outs = [t.BatchResult() for _ in range(8 if idle else 1)]
        general = os.environ.get("AB_GENERAL") == "1"   # a timestamp per request (k_eval_general)
        nows = [torch.arange(batch, dtype=torch.int64, device=dev) + (W.T0_NS + b * 1_000_000) for b in range(16)] if general else None
        it, best = 0, 1e9
        for rep in range(4):
            n = 10 if rep == 0 else steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                eng.rate_limit_batch_slots(db[it % nb], registered=True, quantity=1, now_ns=(nows[it % 16] if general else W.T0_NS + it * 1_000_000), want=("allowed",),
                                           out=outs[it % len(outs)], inputs_ready=True, outputs_idle=idle)
                it += 1
            torch.cuda.synchronize()
            if rep:
                best = min(best, (time.perf_counter() - t0) / n)
        bad = eng.selfcheck()
        print(f"{name:22s} {('g_' if general else '') + sname:8s} {best * 1e6:8.1f} {batch / best / 1e9:7.2f}" + (f"   SELFCHECK {bad}" if bad else ""), flush=True)
        eng.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    names = sys.argv[2:] or list(CONFIGS)
    from throttlecrab_amd import workload as W
    keys, batch, nb = 10_000_000, 1 << 20, 16
    z = W.Zipf(keys)
    np.savez(CACHE, uniform=np.stack([W.uniform_slots(keys, batch, start=i * batch).astype(np.int32) for i in range(nb)]),
             zipf=np.stack([z.slots(batch, start=i * batch).astype(np.int32) for i in range(nb)]))
    print(f"{'config':22s} {'stream':8s} {'us/step':>8s} {'G/s':>7s}   (best of 3 x {steps} steps, one process per configuration)", flush=True)
    for name in names:
        env = {k: v for k, v in os.environ.items() if k not in KNOBS}
        env.update({k: v for k, v in CONFIGS[name].items() if not k.startswith("_")})
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, str(steps)], env=env, timeout=120, check=False)


if __name__ == "__main__":
    main()
