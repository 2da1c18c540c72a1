"""N>1 path on CPU: world_size-2 gloo processes.  Each rank decides only the keys
it owns (with the oracle standing in for the per-GPU engine -- no GPU here) and
the counter blocks are all-gathered exactly as bench.py does over RCCL.  The
union of the shard results must equal one sequential pass over the whole
stream (keys are independent units)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from throttlecrab_amd import sharded, workload as W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_keys, n = 5000, 40000
    gids = W.Zipf(n_keys).slots(n).astype(np.uint64) + np.uint64(10**9)  # global key ids
    now = W.T0_NS + (np.arange(n) // 1000) * 1_000_000
    pos, mine = sharded.partition(gids, world, rank)
    slots = sharded.LocalSlots().resolve(mine)
    orc = O.DenseOracle(n_keys)
    res = orc.batch_slots(slots, 5, 50, 60, 1, now[pos])
    block = torch.zeros(8, dtype=torch.int64)
    block[0], block[1], block[2] = len(pos), int(res.allowed.sum()), int((1 - res.allowed).sum())
    per_rank, totals = sharded.all_gather_counters(block, dist, world)
    allowed_global = np.zeros(n, np.int64)
    allowed_global[pos] = res.allowed
    t = torch.from_numpy(allowed_global)
    dist.all_reduce(t)  # disjoint shards: sum == union
    cover = torch.zeros(n, dtype=torch.int64)
    cover[torch.from_numpy(pos)] = 1
    dist.all_reduce(cover)
    if rank == 0:
        q.put((totals, per_rank.tolist(), t.numpy().tolist(), cover.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_pass():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    totals, per_rank, allowed, cover = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_keys, n = 5000, 40000
    gids = W.Zipf(n_keys).slots(n)
    now = W.T0_NS + (np.arange(n) // 1000) * 1_000_000
    ref = O.DenseOracle(n_keys).batch_slots(gids, 5, 50, 60, 1, now)
    assert all(c == 1 for c in cover), "every request is owned by exactly one rank"
    assert np.array_equal(np.array(allowed), ref.allowed.astype(np.int64))
    assert totals["total"] == n and totals["allowed"] == int(ref.allowed.sum())
    assert totals["allowed"] + totals["denied"] == n
    assert len(per_rank) == 2 and per_rank[0][0] + per_rank[1][0] == n
    assert min(per_rank[0][0], per_rank[1][0]) > 0.2 * n  # both shards get real work


def test_owner_is_stable_and_balanced():
    from throttlecrab_amd import sharded
    ids = np.arange(100000, dtype=np.uint64)
    for world in (2, 4, 8):
        o = sharded.owner_of(ids, world)
        assert np.array_equal(o, sharded.owner_of(ids, world))
        cnt = np.bincount(o, minlength=world)
        assert cnt.min() > 0.9 * len(ids) / world
