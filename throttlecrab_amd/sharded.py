"""Hash-sharded multi-GPU limiter: one process per GPU, one Engine per process.

The reference has no distributed mode ("use client-side sharding by key",
README.md:247-249).  Keys are independent units, so the decision path needs NO
collective: (owner, slot) of a global key id is a bijection of the id space
(tc_route_host / rt::k_route_*); every rank decides only the requests it owns.

Two ways to bring a request to its owner:
  replicate  every rank is handed the whole global batch and keeps what it owns (tc_route_batch, only = rank):
             per-GPU routing work grows with the number of GPUs;
  exchange   every rank routes only ITS 1/N slice of the batch into one segment per destination (only = -1) and puts
             each segment into the destination's inbox (tc_forward_segments: one launch writing peer memory over
             xGMI); a destination evaluates the N inboxes of a step as ONE batch whose slot column comes in pieces
             (tc_batch.seg_slot), sources in rank order, so a key's requests keep the order of the global stream.
             Per-GPU work is independent of N.  No collective: peer copies + one mailbox word per (source,
             destination, ring slot) in shared host memory, written by the source's host once its copies have landed.
The one collective is observability: the per-GPU counter blocks (cf. throttlecrab-server/src/metrics.rs:84-94) and
the top-denied blocks are all-gathered (RCCL over xGMI on GPUs, gloo in the CPU tests).
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


# ---- the exchange (see the module docstring) -------------------------------------------------------------------------
def split_segments(global_slice: np.ndarray, world: int, keys_per_shard: int):
    """Host mirror of tc_route_batch(only = -1): what one source rank sends -- [shard-local slots for destination d,
    in stream order] for d in range(world)."""
    owner, slot = route(global_slice, world, keys_per_shard)
    return [slot[owner == d] for d in range(world)]


class LocalFabric:
    """Inboxes and mailboxes of every shard in ONE process (tests: several engines on one device, where a peer copy
    degenerates to a device-to-device copy)."""

    def __init__(self, world: int, seg_cap: int, ring: int, device):
        import torch
        self.world, self.ring, self.seg_cap = world, ring, seg_cap
        self._inbox = [torch.empty((ring, world, seg_cap), dtype=torch.int32, device=device) for _ in range(world)]
        self.mail = np.zeros((world, ring, world, 2), np.uint32)  # [dst][ring slot][src] = (count, step + 1)
        self.done = np.zeros(world, np.int64)                      # [dst] steps whose inboxes may be overwritten
        self.ranks = []                                            # the ExchangeRanks driven by this one thread

    def inbox(self, dst: int, slot: int, src: int):
        return self._inbox[dst][slot, src]

    def poll(self):
        """one thread drives every shard: a rank that waits for another rank's progress has to note it itself"""
        for r in self.ranks:
            r._publish_done()

    def close(self):
        pass


class IpcFabric:
    """One process per GPU on one node: every rank allocates its inboxes in its own HBM and shares them through IPC
    memory handles (exchanged ONCE, at set-up, with all_gather_object); mailboxes live in a shared-memory segment.
    Nothing on the decision path goes through torch.distributed."""

    def __init__(self, dist, rank: int, world: int, seg_cap: int, ring: int, device):
        import torch
        from multiprocessing import shared_memory
        from torch.multiprocessing.reductions import reduce_tensor
        self.world, self.ring, self.seg_cap, self.rank = world, ring, seg_cap, rank
        mine = torch.empty((ring, world, seg_cap), dtype=torch.int32, device=device)
        handles = [None] * world
        dist.all_gather_object(handles, reduce_tensor(mine))
        self._inbox = []
        for r, (fn, args) in enumerate(handles):
            self._inbox.append(mine if r == rank else fn(*args))  # (peer access is enabled lazily by the IPC open)
        words = world * ring * world * 2
        nbytes = words * 4 + world * 8
        name = [None]
        if rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
            self._shm.buf[:nbytes] = bytes(nbytes)
            name[0] = self._shm.name
        dist.broadcast_object_list(name, src=0)
        if rank != 0:
            self._shm = shared_memory.SharedMemory(name=name[0])
        self.mail = np.ndarray((world, ring, world, 2), np.uint32, buffer=self._shm.buf, offset=0)
        self.done = np.ndarray((world,), np.int64, buffer=self._shm.buf, offset=words * 4)
        dist.barrier()

    def inbox(self, dst: int, slot: int, src: int):
        return self._inbox[dst][slot, src]

    def close(self):
        self.mail = self.done = None
        try:
            self._shm.close()
            if self.rank == 0:
                self._shm.unlink()
        except (OSError, BufferError):
            pass


class ExchangeRank:
    """One rank's side of the exchange.  Per step i (the caller pipelines them: route a few steps ahead of post, post
    ahead of collect):
        route(i, slice)   tc_route_batch(only = -1, out_dst = the destinations' inboxes) of this rank's slice on a
                          grouping stream: routing and forwarding in ONE pass, counts -> pinned memory
        post(i)           once the router's tag is in (every segment has landed): (count, i + 1) into every
                          destination's mailbox
        collect(i)        wait for every source's mailbox word of step i -> the segments to evaluate
        evaluate(i, ...)  one batch, slot column in pieces, sources in rank order (chunks of at most max_batch)"""

    def __init__(self, engine, fabric, rank: int, world: int, slice_len: int, route_ring: int = 8):
        import torch
        self.eng, self.fab, self.rank, self.world = engine, fabric, rank, world
        dev = torch.device(f"cuda:{engine.device}")
        self.route_ring = route_ring
        self.counts_dev = [torch.zeros(world, dtype=torch.int32, device=dev) for _ in range(route_ring)]
        self.counts_host = [engine.host_alloc(world + 1, np.uint32) for _ in range(route_ring)]
        for c in self.counts_host:
            c[:] = 0
        self.eval_events = {}
        self._event_pool = []
        if hasattr(fabric, "ranks"):
            fabric.ranks.append(self)
        self.host_s = {"route": 0.0, "post": 0.0, "collect": 0.0, "evaluate": 0.0}  # host seconds per phase (diagnostics)

    def route(self, step: int, global_slice):
        r, k = step % self.route_ring, step % self.fab.ring
        # flow control: a destination's inbox slot is free once it has evaluated step - ring
        while any(int(self.fab.done[d]) < step - self.fab.ring + 1 for d in range(self.world)):
            self.fab.poll() if hasattr(self.fab, "poll") else self._publish_done()
        if global_slice.numel() > self.fab.seg_cap:
            raise RuntimeError(f"a slice of {global_slice.numel()} requests exceeds the inbox capacity {self.fab.seg_cap}")
        # (the router's output goes to the inboxes, never to a batch's slot column: it need not wait for any batch in flight)
        self.eng.route_batch(global_slice, self.world, only=-1, out=(None, None, self.counts_dev[r]), ahead=True,
                             host_counts=self.counts_host[r], tag=step + 1, no_readers=True,
                             out_dst=[self.fab.inbox(d, k, self.rank) for d in range(self.world)])

    def post(self, step: int):
        r, k = step % self.route_ring, step % self.fab.ring
        while int(self.counts_host[r][self.world]) != step + 1:  # routed a few steps ago: no wait in steady state
            pass
        for d in range(self.world):
            self.fab.mail[d, k, self.rank, 0] = int(self.counts_host[r][d])
        for d in range(self.world):   # (count before tag: a reader that sees the tag sees the count)
            self.fab.mail[d, k, self.rank, 1] = step + 1

    def collect(self, step: int):
        k = step % self.fab.ring
        mail = self.fab.mail[self.rank, k]
        while any(int(mail[s, 1]) != step + 1 for s in range(self.world)):
            pass
        return [(self.fab.inbox(self.rank, k, s), int(mail[s, 0])) for s in range(self.world)]

    def evaluate(self, step: int, segments, now_ns: int, outs, **kw):
        """-> requests decided.  outs: a list of BatchResult to cycle through (one per chunk in flight)."""
        import torch
        cap = self.eng.max_batch
        total = sum(cnt for _, cnt in segments)
        if total <= cap:
            pieces = [segments]  # (the common case: no slicing, no copies)
        else:
            pieces, chunk, size = [], [], 0
            for tns, cnt in segments:  # cut the concatenation into chunks of at most max_batch requests
                at = 0
                while cnt - at > 0:
                    take = min(cnt - at, cap - size)
                    chunk.append((tns[at:] if at else tns, take))
                    size += take
                    at += take
                    if size == cap:
                        pieces.append(chunk)
                        chunk, size = [], 0
            if chunk:
                pieces.append(chunk)
        for j, ch in enumerate(pieces):
            self.eng.rate_limit_batch_slots(None, segments=ch, registered=True, quantity=1, now_ns=now_ns, want=("allowed",),
                                            out=outs[(step * 4 + j) % len(outs)], inputs_ready=True, outputs_idle=True, **kw)
        # completion of this step's evaluation frees its inbox slots (events are recycled: creating one costs microseconds)
        ev = self._event_pool.pop() if self._event_pool else torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.eng.device))
        self.eval_events[step] = ev
        self._publish_done()
        return total

    def timed(self, name, *args, **kw):
        """run phase `name` and book its host time (bench.py reports where a step's host time goes)"""
        import time
        t0 = time.perf_counter()
        r = getattr(self, name)(*args, **kw)
        self.host_s[name] += time.perf_counter() - t0
        return r

    def _publish_done(self):
        """steps whose evaluation has finished free their inbox slots"""
        for st in sorted(self.eval_events):
            if not self.eval_events[st].query():
                break
            self.fab.done[self.rank] = st + 1
            self._event_pool.append(self.eval_events.pop(st))
