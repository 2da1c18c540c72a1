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


def route(global_ids: np.ndarray, world: int, keys_per_shard: int):
    """Host mirror of tc_route_batch (tc_route_host: the same bijection of [0, world * keys_per_shard) the device
    kernels use) -> (owner uint32[n], shard-local slot uint32[n]).  Needs no GPU."""
    import ctypes as C

    from . import _lib as L
    ids = np.ascontiguousarray(global_ids, dtype=np.uint32)
    owner, slot = np.empty(len(ids), np.uint32), np.empty(len(ids), np.uint32)
    rc = L.load().tc_route_host(world, keys_per_shard, len(ids), ids.ctypes.data, owner.ctypes.data, slot.ctypes.data)
    if rc != 0:
        raise ValueError(f"tc_route_host({world}, {keys_per_shard}) failed: {rc}")
    return owner, slot


def route_inverse(owner: np.ndarray, slot: np.ndarray, world: int, keys_per_shard: int) -> np.ndarray:
    """(owner, shard-local slot) -> global key id (what the all-gathered top-denied block reports)."""
    from . import _lib as L
    o, s = np.ascontiguousarray(owner, dtype=np.uint32), np.ascontiguousarray(slot, dtype=np.uint32)
    out = np.empty(len(o), np.uint64)
    rc = L.load().tc_route_inverse(world, keys_per_shard, len(o), o.ctypes.data, s.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise ValueError(f"tc_route_inverse failed: {rc}")
    return out


def shard_requests(global_ids: np.ndarray, world: int, rank: int, keys_per_shard: int):
    """What rank `rank` keeps of a global batch: (positions in the batch, shard-local slots), in stream order --
    the host-side equivalent of tc_route_batch(only=rank)."""
    owner, slot = route(global_ids, world, keys_per_shard)
    pos = np.nonzero(owner == rank)[0]
    return pos, slot[pos]


TOPK = 100  # denied keys every rank contributes to the metrics exchange (metrics.rs:24-76: TopDeniedKeys)


def pack_top_denied(top, rank: int, world: int, keys_per_shard: int) -> np.ndarray:
    """[(slot, count)] of one shard -> int64[TOPK, 2] block of (global key id, count), zero-padded."""
    block = np.zeros((TOPK, 2), np.int64)
    if top:
        slots = np.array([s for s, _ in top[:TOPK]], np.uint32)
        block[: len(slots), 0] = route_inverse(np.full(len(slots), rank, np.uint32), slots, world, keys_per_shard).astype(np.int64)
        block[: len(slots), 1] = [c for _, c in top[:TOPK]]
    return block


def merge_top_denied(gathered: np.ndarray, k: int = TOPK):
    """all-gathered [world * TOPK, 2] blocks -> the k most denied keys of the whole system [(global id, count)]
    (a key lives on one shard, so the blocks never overlap)."""
    rows = gathered.reshape(-1, 2)
    rows = rows[rows[:, 1] > 0]
    order = np.lexsort((rows[:, 0], -rows[:, 1]))
    return [(int(rows[i, 0]), int(rows[i, 1])) for i in order[:k]]


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
