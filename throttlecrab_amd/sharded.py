"""Hash-sharded multi-GPU limiter: one process per GPU, one Engine per process.

The reference has no distributed mode ("use client-side sharding by key",
README.md:247-249).  Keys are independent units, so the decision path needs NO
collective: owner(key) = mix64(key id) mod world; every rank decides only the
requests it owns.  The one exchange is observability: the per-GPU counter
blocks (cf. throttlecrab-server/src/metrics.rs:84-94) are all-gathered
(RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import numpy as np

from . import workload as W
from ._lib import TC_CNT_COUNT, TC_CNT_NAMES


def owner_of(global_ids: np.ndarray, world: int) -> np.ndarray:
    return W.shard_of(np.asarray(global_ids), world)


def partition(global_ids: np.ndarray, world: int, rank: int):
    """Requests of a global stream owned by `rank` -> (positions, global ids), in
    stream order (order inside a key is what the sequential semantics needs)."""
    own = owner_of(global_ids, world) == rank
    pos = np.nonzero(own)[0]
    return pos, np.asarray(global_ids)[pos]


class LocalSlots:
    """Dense shard-local slot ids for the global key ids a rank owns (host side
    routing table; in string mode the on-device hash table plays this role)."""

    def __init__(self):
        self.map = {}

    def resolve(self, global_ids: np.ndarray) -> np.ndarray:
        out = np.empty(len(global_ids), np.uint32)
        m = self.map
        for i, g in enumerate(global_ids.tolist()):
            s = m.get(g)
            if s is None:
                s = len(m)
                m[g] = s
            out[i] = s
        return out


class _DevPtrView:
    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def device_counter_view(engine):
    """The engine's counter block as an int64 CUDA tensor aliasing device memory
    (no host round trip before the all-gather)."""
    import torch
    return torch.as_tensor(_DevPtrView(engine.counters_device_ptr(), TC_CNT_COUNT), device=f"cuda:{engine.device}")


def all_gather_counters(local_block, dist, world: int):
    """all-gather of one counter block per rank -> (per_rank [world, 8], totals dict)."""
    import torch
    gathered = torch.zeros(world * local_block.numel(), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(gathered, local_block.contiguous())
    per_rank = gathered.view(world, -1)
    tot = per_rank.sum(dim=0).tolist()
    return per_rank, {k: int(tot[i]) for i, k in enumerate(TC_CNT_NAMES)}
