#!/usr/bin/env python3
"""The configs[4] stream (workload.Config4Stream: what bench.py's string_keys_config4 times and
tests/test_gpu_keys_spec.py checks) for rocprofv3: 10 prefill batches, then `steps` mixed batches pipelined,
a sweep every 4.    usage: profile_keys.py [short|long] [steps=8]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

import throttlecrab_amd as t  # noqa: E402
from throttlecrab_amd import workload as W  # noqa: E402

long = len(sys.argv) > 1 and sys.argv[1] == "long"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 1 << 20
dev = torch.device("cuda:0")
st = W.Config4Stream(B, n_prefill=10, long=long, sweep_every=4)
eng = t.Engine(B * 11, B, key_mode=True, key_arena_bytes=(448 << 20) if long else 0)
eng.use_torch_stream()
res = t.BatchResult()
b, c, p = W.Config4Stream.PARAMS
for k in range(10):
    ids, now = st.prefill(k)
    kb, ko = st.keys(ids, device=dev)
    eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now, want=("allowed",), out=res, inputs_ready=True)
    torch.cuda.synchronize()
mixed = []
for s in range(steps):
    ids, now = st.mixed(s)
    mixed.append((st.keys(ids, device=dev), now))
torch.cuda.synchronize()
for s, ((kb, ko), now) in enumerate(mixed):
    eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now, want=("allowed",), out=res, inputs_ready=True)
    if st.sweep_due(s):
        eng.sweep_expired_async(now)
torch.cuda.synchronize()
print(eng.counters())
eng.close()
