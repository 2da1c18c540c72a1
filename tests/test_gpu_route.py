"""tc_route_batch (csrc/route_kernels.hpp) on the GPU: the device partition of a global batch equals the host
mirror (tc_route_host) -- same owners, same shard-local slots, request order kept inside every destination --
for one destination (`only`) and for all of them, at tile / wave edge sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("passes", ["default", "three_kernels"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 300_000])
def test_device_partition_equals_host_mirror(world, n, passes, monkeypatch):
    if passes == "three_kernels":  # (one destination is otherwise routed in one pass over the ids)
        monkeypatch.setenv("TCGPU_ROUTE_3PASS", "1")
    else:
        monkeypatch.delenv("TCGPU_ROUTE_3PASS", raising=False)
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    cap = 20_000
    rng = np.random.default_rng(world * 1000 + n)
    ids = rng.integers(0, world * cap, n).astype(np.uint32)
    if n > 10:
        ids[n // 2:n // 2 + 5] = ids[0]  # duplicates of one key: their order must survive
    owner, slot = sharded.route(ids, world, cap)
    eng = t.Engine(cap, 1 << 16)
    eng.use_torch_stream()
    d = torch.from_numpy(ids.astype(np.int32)).cuda()
    # every destination, one segment after the other
    slots, pos, counts = eng.route_batch(d, world, only=-1, want_pos=True)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    assert np.array_equal(counts, np.bincount(owner, minlength=world))
    slots, pos = slots.cpu().numpy().astype(np.uint32), pos.cpu().numpy()
    at = 0
    for dest in range(world):
        want = np.nonzero(owner == dest)[0]
        assert np.array_equal(pos[at:at + len(want)], want), dest          # stable: request order
        assert np.array_equal(slots[at:at + len(want)], slot[want]), dest
        at += len(want)
    # every destination into a buffer of its own (what the exchange does with the inboxes): one pass (k_route_split_one) /
    # three kernels
    bufs = [torch.full((n + 1,), -7, dtype=torch.int32, device="cuda") for _ in range(world)]
    _, _, c3 = eng.route_batch(d, world, only=-1, out_dst=bufs)
    torch.cuda.synchronize()
    assert np.array_equal(c3.cpu().numpy(), counts)
    for dest in range(world):
        want = np.nonzero(owner == dest)[0]
        got = bufs[dest].cpu().numpy()
        assert np.array_equal(got[: len(want)].astype(np.uint32), slot[want]), dest   # stable: request order
        assert (got[len(want):] == -7).all(), dest                                      # and nothing behind the segment
    # one destination only
    for dest in sorted({0, world - 1, world // 2}):
        s2, p2, c2 = eng.route_batch(d, world, only=dest, want_pos=True)
        torch.cuda.synchronize()
        want = np.nonzero(owner == dest)[0]
        assert np.array_equal(c2.cpu().numpy(), counts)
        assert np.array_equal(p2.cpu().numpy()[: len(want)], want)
        assert np.array_equal(s2.cpu().numpy().astype(np.uint32)[: len(want)], slot[want])
    eng.close()


@pytest.mark.parametrize("n,world", [(4_100_000, 2), (6_000_000, 8), (8 << 20, 8), (9_000_000, 3)])
def test_one_destination_of_a_big_global_batch(n, world):
    """the one-pass router at its grid limits: 1001 tiles of 4096 ids, 733 and 1024 tiles of 8192, and beyond them
    (count | scan | scatter)"""
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    cap = 10_000_000
    rng = np.random.default_rng(n)
    ids = rng.integers(0, world * cap, n).astype(np.uint32)
    owner, slot = sharded.route(ids, world, cap)
    eng = t.Engine(cap, 1 << 16)
    eng.use_torch_stream()
    d = torch.from_numpy(ids.astype(np.int32)).cuda()
    for dest in (0, world - 1):
        s2, p2, c2 = eng.route_batch(d, world, only=dest, want_pos=True)
        torch.cuda.synchronize()
        want = np.nonzero(owner == dest)[0]
        assert np.array_equal(c2.cpu().numpy(), np.bincount(owner, minlength=world))
        assert np.array_equal(p2.cpu().numpy()[: len(want)], want)
        assert np.array_equal(s2.cpu().numpy().astype(np.uint32)[: len(want)], slot[want])
    assert eng.selfcheck() == 0
    eng.close()


@pytest.mark.parametrize("how", ["side_stream", "ahead"])
def test_router_ahead_of_pipelined_batches(how):
    """The router several global batches ahead of the evaluations (what bench.py --gpus N does): on a caller's side
    stream (tc_route.stream), or on the engine's grouping streams with the counts polled from pinned host memory
    (TC_ROUTE_AHEAD + out_count_host, ring entries reused); the routed batches evaluated as TC_B_INPUTS_READY
    batches give the oracle's results"""
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    from tests.test_gpu_slots import T0, _oracle
    world, me, cap, n, ring_n, nb = 4, 2, 50_000, 120_000, 3, 9
    rng = np.random.default_rng(77)
    eng, orc = t.Engine(cap, n), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(5, 10, 60)
    side = torch.cuda.Stream()
    batches = [rng.integers(0, world * cap, n).astype(np.uint32) for _ in range(nb)]
    d = [torch.from_numpy(b.astype(np.int32)).cuda() for b in batches]
    ring = [(torch.empty(n, dtype=torch.int32, device="cuda"), None, torch.zeros(world, dtype=torch.int32, device="cuda")) for _ in range(ring_n)]
    host = [eng.host_alloc(world + 1, np.uint32) for _ in range(ring_n)]
    for h in host:
        h[:] = 0
    done = [torch.cuda.Event() for _ in range(ring_n)]
    torch.cuda.synchronize()

    def route(i):
        r = i % ring_n
        if how == "ahead":
            eng.route_batch(d[i], world, only=me, out=ring[r], ahead=True, host_counts=host[r], tag=i + 1)
        else:
            side.wait_stream(torch.cuda.current_stream())  # the ring entry's last reader
            eng.route_batch(d[i], world, only=me, out=ring[r], stream=side)
            done[r].record(side)

    outs = []
    for j in range(ring_n - 1):
        route(j)
    for i in range(nb):
        if i + ring_n - 1 < nb:
            route(i + ring_n - 1)  # (its ring entry was read by batch i - 1, which is on the engine's stream)
        r = i % ring_n
        if how == "ahead":
            while int(host[r][world]) != i + 1:
                pass
            mine = int(host[r][me])
        else:
            done[r].synchronize()
            mine = int(ring[r][2].cpu()[me])
        res = eng.rate_limit_batch_slots(ring[r][0][:mine], registered=True, quantity=1, now_ns=T0 + i * 10**8, want=("allowed", "status"),
                                         inputs_ready=True)
        outs.append((mine, res.allowed, res.status))  # (each call has its own result tensors; read after the synchronisation)
    torch.cuda.synchronize()
    for i, (mine, allowed, status) in enumerate(outs):
        owner, slot = sharded.route(batches[i], world, cap)
        want = slot[owner == me]
        assert mine == len(want), i
        ref = orc.batch_slots(want, 5, 10, 60, 1, T0 + i * 10**8)
        assert np.array_equal(allowed.cpu().numpy(), ref.allowed.astype(np.uint8)), i
        assert np.array_equal(status.cpu().numpy(), ref.status.astype(np.uint8)), i
    assert eng.selfcheck() == 0
    eng.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("stream_kind", ["uniform", "zipf"])
def test_exchange_between_engines_equals_one_sequential_pass(world, stream_kind):
    """--route exchange with every shard in ONE process (sharded.LocalFabric: a peer copy degenerates to a device-to-device
    copy): rank r routes only slice r of each global batch (tc_route_batch, only = -1), tc_forward_segments puts the
    segments into the destinations' inboxes, every destination evaluates its `world` inboxes as ONE batch with a
    segmented slot column, sources in rank order.  The union of the shards' decisions == one sequential pass of the
    oracle keyed by the global id, although no shard ever sees the whole stream."""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    from tests.test_gpu_slots import T0
    from throttlecrab_amd import sharded
    from throttlecrab_amd import workload as W
    cap, B, steps, LA = 6000, 20_000, 9, 2
    n_glob = world * cap
    rng = np.random.default_rng(world)
    z = W.Zipf(n_glob)
    glob = [(z.slots(world * B, start=i * world * B) if stream_kind == "zipf" else rng.integers(0, n_glob, world * B)).astype(np.uint32)
            for i in range(steps)]
    dev = torch.device("cuda:0")
    fab = sharded.LocalFabric(world, B, ring=4, device=dev)
    engs, ranks = [], []
    for r in range(world):
        e = t.Engine(cap, 2 * B, fixed_params=True)
        e.use_torch_stream()
        e.register_params_uniform(5, 10, 60)
        engs.append(e)
        ranks.append(sharded.ExchangeRank(e, fab, r, world, B, route_ring=4))
    d_slices = [[torch.from_numpy(g[r * B:(r + 1) * B].astype(np.int32)).to(dev) for r in range(world)] for g in glob]
    outs = [[t.BatchResult() for _ in range(8)] for _ in range(world)]
    got = {}  # (step, rank) -> (segments' counts, result)
    # the same pipeline bench.py runs per rank, all ranks stepped in one loop
    for i in range(steps + LA):
        for r in range(world):
            if i < steps:
                ranks[r].route(i, d_slices[i][r])
        for r in range(world):
            if 0 <= i - 1 < steps:
                ranks[r].post(i - 1)
        for r in range(world):
            st = i - LA
            if 0 <= st < steps:
                segs = ranks[r].collect(st)
                res = t.BatchResult()
                n = ranks[r].evaluate(st, segs, T0 + st * 300_000_000, [res])
                got[(st, r)] = ([c for _, c in segs], res, n)
    torch.cuda.synchronize()
    ref_orc = O.DenseOracle(n_glob)
    for st in range(steps):
        g = glob[st]
        ref = ref_orc.batch_slots(g, 5, 10, 60, 1, T0 + st * 300_000_000)
        owner, _ = sharded.route(g, world, cap)
        for r in range(world):
            counts, res, n = got[(st, r)]
            # what rank r evaluated, in its order: for each source s, the requests of slice s that r owns
            idx = np.concatenate([s * B + np.nonzero(owner[s * B:(s + 1) * B] == r)[0] for s in range(world)])
            assert counts == [int((owner[s * B:(s + 1) * B] == r).sum()) for s in range(world)]
            assert n == len(idx)
            assert np.array_equal(res.allowed.cpu().numpy()[:n], ref.allowed[idx].astype(np.uint8)), (st, r)
    for e in engs:
        assert e.selfcheck() == 0
        e.close()
    fab.close()


def test_segmented_slot_column_equals_the_concatenation():
    """tc_batch.seg_slot: pieces of any size (empty ones included), in order and pipelined, full result"""
    import torch

    import throttlecrab_amd as t
    from tests.test_gpu_slots import FIELDS, T0, _oracle, assert_same
    cap = 3000
    eng, orc = t.Engine(cap, 1 << 16), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(5, 10, 60)
    rng = np.random.default_rng(12)
    for rnd, sizes in enumerate(([7], [0, 5, 0], [1000, 1, 63, 64, 65, 0, 4097], [30000, 20000], [1] * 64)):
        pieces = [rng.integers(0, cap, n).astype(np.uint32) for n in sizes]
        d = [torch.from_numpy(np.concatenate([p, rng.integers(0, cap, 5).astype(np.uint32)]).astype(np.int32)).cuda() for p in pieces]  # (slack behind each piece)
        whole = np.concatenate(pieces)
        ref = orc.batch_slots(whole, 5, 10, 60, 1, T0 + rnd * 10**9)
        res = eng.rate_limit_batch_slots(None, segments=[(x, n) for x, n in zip(d, sizes)], registered=True, quantity=1, now_ns=T0 + rnd * 10**9,
                                         want=FIELDS, inputs_ready=bool(rnd % 2))
        torch.cuda.synchronize()
        assert_same(res, ref, f"round {rnd}")
    with pytest.raises(t.engine.TcError):
        eng.rate_limit_batch_slots(None, segments=[(d[0], 1)] * 65 if False else [(d[0], 1)], registered=True, quantity=1, now_ns=T0, unique=True)
    assert eng.selfcheck() == 0
    eng.close()


def test_forward_segments_copies_every_piece_to_its_destination():
    """tc_forward_segments: the router's contiguous segments -> one destination buffer each, in one launch"""
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    world, cap, n = 5, 1000, 40_000
    ids = np.random.default_rng(2).integers(0, world * cap, n).astype(np.uint32)
    eng = t.Engine(cap, 1 << 16)
    eng.use_torch_stream()
    slots, _, counts = eng.route_batch(torch.from_numpy(ids.astype(np.int32)).cuda(), world, only=-1)
    torch.cuda.synchronize()
    cnt = counts.cpu().numpy().tolist()
    dsts = [torch.full((max(c, 1) + 3,), -1, dtype=torch.int32, device="cuda") for c in cnt]
    eng.forward_segments(slots, cnt, dsts)
    torch.cuda.synchronize()
    segs = sharded.split_segments(ids, world, cap)
    for d in range(world):
        got = dsts[d].cpu().numpy()
        assert np.array_equal(got[:cnt[d]].astype(np.uint32), segs[d]) and (got[cnt[d]:] == -1).all(), d
    eng.close()


@pytest.mark.parametrize("world", [3, 8])   # (8: BASELINE configs[3]'s world, every rank of it on this one GPU -- VERDICT r5 #7)
def test_exchange_step_is_one_library_call_and_matches_the_single_pass(world):
    """tc_exchange_step (route i + 4, post i + 1, collect + evaluate i in ONE call, what bench.py --route exchange times), the
    shards of one world driven by one thread over a LocalFabric; steps larger than the engine's max_batch are evaluated in chunks."""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    from tests.test_gpu_slots import T0
    from throttlecrab_amd import sharded
    from throttlecrab_amd import workload as W
    cap, B, steps, LA_R, LA_P = 5000, 12_000, 14, 4, 1
    n_glob = world * cap
    z = W.Zipf(n_glob)
    glob = [z.slots(world * B, start=i * world * B).astype(np.uint32) for i in range(steps + LA_R)]
    dev = torch.device("cuda:0")
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        fab = sharded.LocalFabric(world, B, ring=8, device=dev)
        engs, ranks = [], []
        for r in range(world):
            e = t.Engine(cap, 13_000, fixed_params=True)   # (a hot key's owner gets more than max_batch requests in a step: chunks)
            e.use_torch_stream()
            e.register_params_uniform(5, 10, 60)
            engs.append(e)
            ranks.append(sharded.ExchangeRank(e, fab, r, world, B))
        d_slices = [[torch.from_numpy(g[r * B:(r + 1) * B].astype(np.int32)).to(dev) for r in range(world)] for g in glob]
        outs = [[t.BatchResult(allowed=torch.zeros(world * B, dtype=torch.uint8, device=dev)) for _ in range(steps)] for _ in range(world)]
        for j in range(LA_R):
            for r in range(world):
                ranks[r].route(j, d_slices[j][r])
        for j in range(LA_P):
            for r in range(world):
                ranks[r].post(j)
        decided = {}
        for i in range(steps):
            for r in range(world):
                decided[(i, r)] = ranks[r].step(i, d_slices[i + LA_R][r], LA_R, LA_P, T0 + i * 300_000_000, outs[r])
        torch.cuda.synchronize()
        ref_orc = O.DenseOracle(n_glob)
        chunked = 0
        for st in range(steps):
            g = glob[st]
            ref = ref_orc.batch_slots(g, 5, 10, 60, 1, T0 + st * 300_000_000)
            owner, _ = sharded.route(g, world, cap)
            for r in range(world):
                idx = np.concatenate([s * B + np.nonzero(owner[s * B:(s + 1) * B] == r)[0] for s in range(world)])
                assert decided[(st, r)] == len(idx)
                chunked += len(idx) > 13_000
                assert np.array_equal(outs[r][st].allowed.cpu().numpy()[:len(idx)], ref.allowed[idx].astype(np.uint8)), (st, r)
        assert chunked > 0
        for x in ranks:
            x.close()
        for e in engs:
            assert e.selfcheck() == 0
            e.close()
        fab.close()


def test_exchange_flow_control_holds_while_the_gpu_is_stalled():
    """an inbox slot is freed by an event on the stream the ENGINE evaluates on.  With the engine's stream stalled
    behind a long sleep, a source that is `ring` steps ahead must not be let into the slot (TC_E_AGAIN in non-blocking mode)
    however the host polls; once the evaluation has run, it is."""
    import torch

    import throttlecrab_amd as t
    from tests.test_gpu_slots import T0
    from throttlecrab_amd import _lib as L
    from throttlecrab_amd import sharded
    cap, B, ring = 3000, 4096, 2
    dev = torch.device("cuda:0")
    for own_stream in (False, True):   # the engine on torch's stream / on its own stream (torch's default stream current)
        ctx = torch.cuda.stream(torch.cuda.Stream(device=dev)) if not own_stream else torch.cuda.stream(torch.cuda.default_stream(dev))
        with ctx:
            fab = sharded.LocalFabric(1, B, ring=ring, device=dev)
            eng = t.Engine(cap, B, fixed_params=True)
            eng.use_torch_stream()
            eng.register_params_uniform(5, 10, 60)
            xr = sharded.ExchangeRank(eng, fab, 0, 1, B)
            ids = torch.from_numpy(np.random.default_rng(1).integers(0, cap, B).astype(np.int32)).to(dev)
            outs = [t.BatchResult(allowed=torch.zeros(B, dtype=torch.uint8, device=dev)) for _ in range(8)]
            for st in range(ring):
                xr.route(st, ids)
                xr.post(st)
            eng.synchronize()
            # stall the stream the engine evaluates on, then enqueue the evaluation of step 0 behind the stall
            stall_stream = torch.cuda.current_stream(dev)
            if not own_stream:   # (the engine's private stream cannot be stalled from torch: there, only that the slot is freed in the end)
                torch.cuda._sleep(int(3e8))   # ~150 ms on the engine's stream
            xr.evaluate(0, xr.collect(0), T0, outs)
            rc = xr._lib.tc_exchange_route(xr._h, ring, ids.data_ptr(), B)   # step `ring` wants the slot step 0 still occupies
            if not own_stream:
                assert rc == L.TC_E_AGAIN, rc
                for _ in range(50):
                    fab.poll()
                    assert xr._lib.tc_exchange_route(xr._h, ring, ids.data_ptr(), B) == L.TC_E_AGAIN
            eng.synchronize()
            stall_stream.synchronize()
            fab.poll()
            assert xr._lib.tc_exchange_route(xr._h, ring, ids.data_ptr(), B) == L.TC_E_OK
            eng.synchronize()
            xr.close()
            assert eng.selfcheck() == 0
            eng.close()
            fab.close()


def test_routers_on_two_caller_streams_do_not_share_scratch_unordered():
    """every router off the grouping streams uses scratch lane 0; two of them on different caller streams are now
    ordered by an event.  Alternate two streams for 60 partitions of different batches and check every one."""
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    world, cap, n = 4, 50_000, 300_000
    eng = t.Engine(cap, 1 << 19)
    eng.use_torch_stream()
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    rng = np.random.default_rng(9)
    ids = [rng.integers(0, world * cap, n).astype(np.uint32) for _ in range(6)]
    d_ids = [torch.from_numpy(x.astype(np.int32)).to(dev) for x in ids]
    torch.cuda.synchronize()
    outs = []
    for i in range(60):
        st = streams[i % 2]
        only = i % world
        slots = torch.empty(n, dtype=torch.int32, device=dev)
        counts = torch.zeros(world, dtype=torch.int32, device=dev)
        eng.route_batch(d_ids[i % 6], world, only=only, out=(slots, None, counts), stream=st)
        outs.append((i % 6, only, slots, counts))
    torch.cuda.synchronize()
    for b, only, slots, counts in outs:
        owner, slot = sharded.route(ids[b], world, cap)
        c = counts.cpu().numpy()
        assert np.array_equal(c, np.bincount(owner, minlength=world))
        assert np.array_equal(slots.cpu().numpy()[:c[only]].astype(np.uint32), slot[owner == only])
    assert eng.selfcheck() == 0
    eng.close()


@pytest.mark.parametrize("n,world", [(1024 * 4096, 8), (1024 * 4096 + 1, 8), (1_500_000, 64), (2_000_003, 5)])
def test_split_router_at_its_grid_limit(n, world):
    """k_route_split_one takes grids of at most 1 024 tiles of 4 096 ids (the tagged count words of a call: [world][tiles]);
    one id more and the three kernels take over.  Every destination's buffer against the host mirror."""
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    cap = 1_000_000
    rng = np.random.default_rng(n % 1000 + world)
    ids = rng.integers(0, min(world * cap, 2**31 - 1), n).astype(np.uint32)
    owner, slot = sharded.route(ids, world, cap)
    eng = t.Engine(cap, 1 << 16)
    eng.use_torch_stream()
    d = torch.from_numpy(ids.astype(np.int32)).cuda()
    counts = np.bincount(owner, minlength=world)
    bufs = [torch.full((int(counts[k]) + 1,), -7, dtype=torch.int32, device="cuda") for k in range(world)]
    for rep in range(2):  # (twice: the words of the first call are still there, tagged with another sequence number)
        _, _, c = eng.route_batch(d, world, only=-1, out_dst=bufs)
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), counts)
        for dest in range(world):
            got = bufs[dest].cpu().numpy()
            assert np.array_equal(got[:-1].astype(np.uint32), slot[owner == dest]), (rep, dest)
            assert got[-1] == -7
    assert eng.selfcheck() == 0
    eng.close()


@pytest.mark.parametrize("world", [3, 8])   # (8: BASELINE configs[3]'s world, every rank of it on this one GPU)
@pytest.mark.parametrize("stream_kind", ["uniform", "zipf"])
def test_replicate_mode_as_one_library_call_per_step(stream_kind, world):
    """tc_shard_step (csrc/shard.hip): route step i + 3 on the grouping streams, poll the router's tag, evaluate the
    rank's share of step i in chunks of at most max_batch -- three ranks of one world stepped in one loop, each seeing the WHOLE
    global stream; the union of their decisions == one sequential pass of the oracle keyed by the global id."""
    import torch
    import throttlecrab_amd as t
    from oracle import oracle as O
    from tests.test_gpu_slots import T0
    from throttlecrab_amd import sharded, workload as W
    cap, G, steps, LA = 40_000, 30_000 * world, 10, 3
    rng = np.random.default_rng(5)
    z = W.Zipf(world * cap)
    glob = [(z.slots(G, start=i * G) if stream_kind == "zipf" else rng.integers(0, world * cap, G)).astype(np.uint32) for i in range(steps)]
    d = [torch.from_numpy(g.astype(np.int32)).cuda() for g in glob]
    engs, ranks, outs = [], [], []
    for r in range(world):
        e = t.Engine(cap, 20_000, fixed_params=True)   # (max_batch below a rank's share: every step is evaluated in chunks)
        e.use_torch_stream()
        e.register_params_uniform(5, 10, 60)
        engs.append(e)
        ranks.append(sharded.ShardRank(e, r, world, G, ring=LA + 2))
        outs.append([t.BatchResult(allowed=torch.zeros(G, dtype=torch.uint8, device="cuda"), status=torch.zeros(G, dtype=torch.uint8, device="cuda"))
                     for _ in range(steps)])
    torch.cuda.synchronize()
    decided = {}
    for r in range(world):
        for j in range(LA):
            ranks[r].route(j, d[j])
    for i in range(steps):
        for r in range(world):
            ahead = d[i + LA] if i + LA < steps else None
            decided[(i, r)] = ranks[r].step(i, ahead, LA, T0 + i * 300_000_000, outs[r], want=("allowed", "status"))
    torch.cuda.synchronize()
    orc = O.DenseOracle(world * cap)
    for i in range(steps):
        ref = orc.batch_slots(glob[i], 5, 10, 60, 1, T0 + i * 300_000_000)
        owner, _ = sharded.route(glob[i], world, cap)
        for r in range(world):
            mine = owner == r
            assert decided[(i, r)] == int(mine.sum()), (i, r)
            k = decided[(i, r)]
            assert np.array_equal(outs[r][i].allowed.cpu().numpy()[:k], ref.allowed[mine].astype(np.uint8)), (i, r)
            assert not outs[r][i].status.cpu().numpy()[:k].any()
    for r in range(world):
        assert engs[r].selfcheck() == 0
        assert ranks[r].wait_us() >= 0.0
        with pytest.raises(t.TcError):
            ranks[r].step(steps, None, LA + 5, T0, outs[r])   # a look-ahead the ring cannot hold
        engs[r].close()


def test_a_share_larger_than_the_output_arrays_is_refused_before_anything_is_written():
    """tc_shard_evaluate decides a rank's SHARE of a global step -- a count that comes from the router, on the device.
    The template's n is what the caller's output arrays hold; a skewed step whose share is larger used to be written past
    them.  Now: TC_E_INVALID_ARG, nothing applied (the same step with arrays that are large enough then decides all of it)."""
    import torch
    import throttlecrab_amd as t
    from tests.test_gpu_slots import T0
    from throttlecrab_amd import sharded
    world, cap, G = 2, 50_000, 40_000
    rng = np.random.default_rng(3)
    glob = rng.integers(0, world * cap, G).astype(np.uint32)
    owner, _ = sharded.route(glob, world, cap)
    share = int((owner == 0).sum())
    e = t.Engine(cap, 65_536, fixed_params=True)
    e.use_torch_stream()
    e.register_params_uniform(5, 10, 60)
    rank = sharded.ShardRank(e, 0, world, G, ring=3)
    d = torch.from_numpy(glob.astype(np.int32)).cuda()
    rank.route(0, d)
    small = [t.BatchResult(allowed=torch.full((share - 100,), 7, dtype=torch.uint8, device="cuda"))]
    guard = torch.full((4096,), 9, dtype=torch.uint8, device="cuda")   # (whatever lies behind the short array)
    with pytest.raises(t.TcError) as err:
        rank.evaluate(0, T0, small, want=("allowed",))
    assert "larger than the output arrays" in str(err.value)
    torch.cuda.synchronize()
    assert int((small[0].allowed != 7).sum()) == 0 and int((guard != 9).sum()) == 0 and e.counters()["total"] == 0
    big = [t.BatchResult(allowed=torch.zeros(G, dtype=torch.uint8, device="cuda"))]
    assert rank.evaluate(0, T0, big, want=("allowed",)) == share
    torch.cuda.synchronize()
    assert e.counters()["total"] == share and e.selfcheck() == 0
    rank.close()
    e.close()
