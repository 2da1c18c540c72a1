#!/usr/bin/env python3
"""What `replicate` mode's router costs a rank as the world grows (one GPU stands in for rank 0 of `world`): a global batch of
world x 1 Mi ids, the rank keeps what it owns.  Device time per call, alone on the GPU (in a run it overlaps the evaluations)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import throttlecrab_amd as t
B = 1 << 20
eng = t.Engine(10_000_000, B, fixed_params=True)
eng.use_torch_stream()
dev = torch.device("cuda:0")
for world in (1, 2, 4, 8):
    n = world * B
    rng = np.random.default_rng(world)
    ids = torch.from_numpy(rng.integers(0, world * 10_000_000, n, dtype=np.int64).astype(np.uint32).view(np.int32)).to(dev)
    out = (torch.empty(n, dtype=torch.int32, device=dev), None, torch.empty(world, dtype=torch.int32, device=dev))
    for _ in range(5):
        eng.route_batch(ids, world, only=0, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        eng.route_batch(ids, world, only=0, out=out)
    b.record()
    torch.cuda.synchronize()
    kept = int(out[2].cpu()[0])
    print(f"world {world}: {n >> 20} Mi ids routed, {kept} kept: {1e3 * a.elapsed_time(b) / 40:7.1f} us per call ({n * 4 / (1e3 * a.elapsed_time(b) / 40) / 1e3:.0f} GB/s of ids)", flush=True)
eng.close()
