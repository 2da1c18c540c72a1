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


N_KEYS, N_REQ = 5000, 40000  # keys per shard, requests in the global stream


def _global_stream(world):
    """one global Zipf stream over the world * N_KEYS global key ids (the same on every rank)"""
    from throttlecrab_amd import workload as W
    gids = W.Zipf(world * N_KEYS).slots(N_REQ).astype(np.uint32)
    now = W.T0_NS + (np.arange(N_REQ) // 1000) * 1_000_000
    return gids, now


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids, now = _global_stream(world)
    # routing: the host mirror of the device partition kernel (tc_route_host == rt::k_route_scatter's map)
    pos, slots = sharded.shard_requests(gids, world, rank, N_KEYS)
    orc = O.DenseOracle(N_KEYS)
    res = orc.batch_slots(slots, 5, 50, 60, 1, now[pos])
    block = torch.zeros(8, dtype=torch.int64)
    block[0], block[1], block[2] = len(pos), int(res.allowed.sum()), int((1 - res.allowed).sum())
    per_rank, totals = sharded.all_gather_counters(block, dist, world)
    # the optional part of the metrics payload: every shard's most denied keys, as (global key id, count)
    denied = np.bincount(slots[res.allowed == 0], minlength=N_KEYS)
    top = sorted(((int(s), int(c)) for s, c in enumerate(denied) if c), key=lambda t: (-t[1], t[0]))[: sharded.TOPK]
    mine = torch.from_numpy(sharded.pack_top_denied(top, rank, world, N_KEYS))
    gathered = torch.zeros(world * sharded.TOPK, 2, dtype=torch.int64)
    dist.all_gather_into_tensor(gathered, mine)
    allowed_global = np.zeros(N_REQ, np.int64)
    allowed_global[pos] = res.allowed
    t = torch.from_numpy(allowed_global)
    dist.all_reduce(t)  # disjoint shards: sum == union
    cover = torch.zeros(N_REQ, dtype=torch.int64)
    cover[torch.from_numpy(pos)] = 1
    dist.all_reduce(cover)
    if rank == 0:
        q.put((totals, per_rank.tolist(), t.numpy().tolist(), cover.numpy().tolist(), sharded.merge_top_denied(gathered.numpy(), 20)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_pass():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    totals, per_rank, allowed, cover, top = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gids, now = _global_stream(2)
    ref = O.DenseOracle(2 * N_KEYS).batch_slots(gids, 5, 50, 60, 1, now)  # one pass, keyed by the global id
    assert all(c == 1 for c in cover), "every request is owned by exactly one rank"
    assert np.array_equal(np.array(allowed), ref.allowed.astype(np.int64))
    assert totals["total"] == N_REQ and totals["allowed"] == int(ref.allowed.sum())
    assert totals["allowed"] + totals["denied"] == N_REQ
    assert len(per_rank) == 2 and per_rank[0][0] + per_rank[1][0] == N_REQ
    assert min(per_rank[0][0], per_rank[1][0]) > 0.2 * N_REQ  # both shards get real work
    # the merged top-denied block == the most denied GLOBAL keys of the single pass
    denied = np.bincount(gids[ref.allowed == 0], minlength=2 * N_KEYS)
    want = sorted(((int(g), int(c)) for g, c in enumerate(denied) if c), key=lambda t: (-t[1], t[0]))[:20]
    assert top == want


def test_route_is_a_bijection_onto_dense_shards():
    """tc_route_host: every global id of [0, world * keys_per_shard) gets its own (owner, slot), every shard exactly
    keys_per_shard slots; the inverse map returns the id; consecutive ids spread over the owners"""
    from throttlecrab_amd import sharded
    for world, cap in ((1, 97), (2, 5000), (3, 77), (8, 4096), (64, 31)):
        ids = np.arange(world * cap, dtype=np.uint32)
        owner, slot = sharded.route(ids, world, cap)
        assert owner.max() == world - 1 and slot.max() == cap - 1
        assert len(np.unique(owner.astype(np.uint64) * cap + slot)) == world * cap
        assert np.array_equal(np.bincount(owner, minlength=world), np.full(world, cap))
        assert np.array_equal(sharded.route_inverse(owner, slot, world, cap), ids.astype(np.uint64))
        if world > 1:
            assert (owner[:-1] != owner[1:]).mean() > 0.5
    # a large key space: balanced owners for a random draw
    owner, _ = sharded.route(np.random.default_rng(3).integers(0, 8 * 10**7, 200000).astype(np.uint32), 8, 10**7)
    cnt = np.bincount(owner, minlength=8)
    assert cnt.min() > 0.95 * 200000 / 8 and cnt.max() < 1.05 * 200000 / 8
    with pytest.raises(ValueError):
        sharded.route(np.zeros(1, np.uint32), 65, 10)


def _exchange_worker(rank, world, port, q):
    """--route exchange on CPU: every rank routes only ITS slice of each global batch into one segment per destination
    (sharded.split_segments == tc_route_batch(only = -1)), the segments travel point to point (gloo send / recv here,
    peer copies on GPUs), and a destination evaluates what it received as one batch, sources in rank order."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gids, now = _global_stream(world)
    n_batches, B = 8, N_REQ // 8 // world           # a global batch = world slices of B requests
    orc = O.DenseOracle(N_KEYS)
    allowed_global = np.zeros(N_REQ, np.int64)
    cover = np.zeros(N_REQ, np.int64)
    decided = 0
    for b in range(n_batches):
        lo = b * world * B
        my_slice = gids[lo + rank * B: lo + (rank + 1) * B]
        owner, slot = sharded.route(my_slice, world, N_KEYS)
        segs = sharded.split_segments(my_slice, world, N_KEYS)
        pos_of = [lo + rank * B + np.nonzero(owner == d)[0] for d in range(world)]   # (test only: where each request came from)
        assert all(np.array_equal(segs[d], slot[owner == d]) for d in range(world))
        # point-to-point: lower rank sends first (two ranks: no deadlock)
        recv_slots, recv_pos = [None] * world, [None] * world
        recv_slots[rank], recv_pos[rank] = segs[rank], pos_of[rank]
        for peer in range(world):
            if peer == rank:
                continue
            def send():
                hdr = torch.tensor([len(segs[peer])], dtype=torch.int64)
                dist.send(hdr, peer)
                if len(segs[peer]):
                    dist.send(torch.from_numpy(segs[peer].astype(np.int64)), peer)
                    dist.send(torch.from_numpy(pos_of[peer].astype(np.int64)), peer)
            def recv():
                hdr = torch.zeros(1, dtype=torch.int64)
                dist.recv(hdr, peer)
                n = int(hdr[0])
                a, c = torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int64)
                if n:
                    dist.recv(a, peer)
                    dist.recv(c, peer)
                recv_slots[peer], recv_pos[peer] = a.numpy().astype(np.uint32), c.numpy()
            if rank < peer:
                send(); recv()
            else:
                recv(); send()
        # sources in rank order: a key's requests keep the order of the global stream
        slots_cat, pos_cat = np.concatenate(recv_slots), np.concatenate(recv_pos)
        assert np.all(np.diff(pos_cat) > 0)          # the concatenation IS in global order
        res = orc.batch_slots(slots_cat, 5, 50, 60, 1, now[pos_cat])
        allowed_global[pos_cat] = res.allowed
        cover[pos_cat] += 1
        decided += len(pos_cat)
    t, c = torch.from_numpy(allowed_global), torch.from_numpy(cover)
    dist.all_reduce(t)
    dist.all_reduce(c)
    if rank == 0:
        q.put((t.numpy().tolist(), c.numpy().tolist(), n_batches * world * B))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_single_pass():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allowed, cover, n_used = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gids, now = _global_stream(2)
    ref = O.DenseOracle(2 * N_KEYS).batch_slots(gids[:n_used], 5, 50, 60, 1, now[:n_used])  # one pass, keyed by the global id
    assert all(c == 1 for c in cover[:n_used]), "every request reaches exactly one owner"
    assert np.array_equal(np.array(allowed[:n_used]), ref.allowed.astype(np.int64))
    assert 0 < ref.allowed.sum() < n_used


def test_split_segments_is_the_router_with_every_destination():
    from throttlecrab_amd import sharded
    rng = np.random.default_rng(4)
    for world, cap in ((2, 5000), (3, 77), (8, 4096)):
        ids = rng.integers(0, world * cap, 20000).astype(np.uint32)
        segs = sharded.split_segments(ids, world, cap)
        assert sum(len(s) for s in segs) == len(ids)
        for d in range(world):
            pos, slots = sharded.shard_requests(ids, world, d, cap)
            assert np.array_equal(segs[d], slots)


# ---- string keys across shards (BASELINE configs[4] sharded; README.md:247-249 shards by KEY) --------------------------------
T0 = 1_700_000_000 * 10**9


def _key_stream():
    """a stream of string keys with hot keys, `key_<i>` and longer ASCII keys mixed"""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    n = 40_000
    ids = rng.integers(0, 6000, n)
    hot = rng.random(n) < 0.3
    ids[hot] = rng.integers(0, 12, int(hot.sum()))
    keys = [(b"key_%d" % i) if i % 3 else (b"tenant:%d:resource/with/a/longer/path/%d" % (i, i * 7919)) for i in ids.tolist()]
    kb, ko = O.pack_keys(keys)
    now = T0 + np.arange(n, dtype=np.int64) * 200_000
    return kb, ko, now


def _key_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kb, ko, now = _key_stream()
    mine_b, mine_o, pos = sharded.split_keys(kb, ko, world)[rank]   # the front door: this shard's keys, in stream order
    st = O.AdaptiveOracle(capacity=10_000, created_ns=T0, auto_cleanup=False)
    res = st.batch_keys(mine_b, mine_o, 5, 50, 60, 1, now[pos])
    allowed = torch.zeros(len(ko) - 1, dtype=torch.int64)
    cover = torch.zeros(len(ko) - 1, dtype=torch.int64)
    allowed[torch.from_numpy(pos)] = torch.from_numpy(res.allowed.astype(np.int64))
    cover[torch.from_numpy(pos)] = 1
    dist.all_reduce(allowed)
    dist.all_reduce(cover)
    if rank == 0:
        q.put((allowed.numpy().tolist(), cover.numpy().tolist(), len(pos)))
    dist.barrier()
    dist.destroy_process_group()


def test_string_keys_sharded_by_key_match_the_single_pass():
    """tc_route_keys_host: owner(key) = mix64(hash(key) ^ salt) mod world.  Two shards, each handed the keys it owns (in
    stream order) with a store of its own: the union of their decisions is the single sequential pass's."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_key_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allowed, cover, n0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kb, ko, now = _key_stream()
    ref = O.AdaptiveOracle(capacity=10_000, created_ns=T0, auto_cleanup=False).batch_keys(kb, ko, 5, 50, 60, 1, now)
    assert all(c == 1 for c in cover), "every request is owned by exactly one shard"
    assert np.array_equal(np.array(allowed), ref.allowed.astype(np.int64))
    assert 0 < int(ref.allowed.sum()) < len(allowed)
    assert 0.3 * len(allowed) < n0 < 0.7 * len(allowed)   # both shards get real work


def test_route_keys_is_a_function_of_the_key_bytes_and_spreads():
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    keys = [b"", b"a", b"key_1", b"key_1", b"key_2", "🦀🔥💻".encode(), b"x" * 1000, b"key:with:colons/and/slashes\\and\\backslashes"]
    kb, ko = O.pack_keys(keys)
    for world in (1, 2, 8, 64):
        o = sharded.route_keys(kb, ko, world)
        assert o.max() < world and o[2] == o[3]
    kb, ko = O.pack_keys([b"key_%d" % i for i in range(80_000)])
    cnt = np.bincount(sharded.route_keys(kb, ko, 8), minlength=8)
    assert cnt.min() > 0.93 * 10_000 and cnt.max() < 1.07 * 10_000
    with pytest.raises(ValueError):
        sharded.route_keys(kb, ko, 65)
    parts = sharded.split_keys(*O.pack_keys(keys), 3)
    assert sorted(int(p) for _, _, pos in parts for p in pos) == list(range(len(keys)))
    for b, off, pos in parts:   # every owner's arena holds exactly its keys, in order
        assert [bytes(b[off[i]:off[i + 1]]) for i in range(len(pos))] == [keys[p] for p in pos]
