"""The range path of the grouping stage (csrc/radix_sort.hpp: k_tile_ranges + k_finish) through the C ABI: streams that switch
between uniform and skewed batches (the host chooses the path from a HINT about recent batches: a skewed batch on the range
path must still come out exact -- k_finish sorts an oversized range through global memory), key spaces at the edges of what the
path takes, batches that are tiny, ragged or larger than the path takes, duplicates beyond the counting path's byte counters.
Everything against the oracle applying the requests one by one (rate_limiter.rs:147-205 in queue order)."""
import numpy as np
import pytest

from tests.test_gpu_slots import T0, _oracle

pytestmark = pytest.mark.gpu


def _run(cap, batches, piped, fixed=True, plan=(5, 50, 60), general=False):
    import torch

    import throttlecrab_amd as t
    eng = t.Engine(cap, max(len(b) for b in batches), fixed_params=fixed)
    eng.use_torch_stream()
    eng.register_params_uniform(*plan)
    orc = _oracle(cap)
    held = []
    rng = np.random.default_rng(len(batches))
    for i, slots in enumerate(batches):
        now = T0 + i * 700_000_000
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        if general:
            nowc = now + np.sort(rng.integers(0, 10**6, len(slots)))
            ref = orc.batch_slots(slots, *plan, 1, nowc)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=torch.from_numpy(nowc).cuda(), want=("allowed", "status"), inputs_ready=piped)
        else:
            ref = orc.batch_slots(slots, *plan, 1, now)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=now, want=("allowed", "remaining", "status"), inputs_ready=piped)
        held.append((res, ref, d))
        if not piped or i % 3 == 2:
            torch.cuda.synchronize()   # (lets the hint of the batches so far reach the host: the next call may change path)
    torch.cuda.synchronize()
    for i, (res, ref, _) in enumerate(held):
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed), f"batch {i}: decisions differ"
        assert np.array_equal(res.status.cpu().numpy(), ref.status), f"batch {i}: statuses differ"
        if not general:
            assert np.array_equal(res.remaining.cpu().numpy(), ref.remaining), f"batch {i}: remaining differs"
    assert eng.selfcheck() == 0
    c = eng.counters()
    eng.close()
    return c


def _uniform(rng, cap, n):
    return rng.integers(0, cap, n).astype(np.uint32)


def _skewed(rng, cap, n, share=0.45, keys=3):
    s = rng.integers(0, cap, n).astype(np.uint32)
    hot = rng.random(n) < share
    s[hot] = rng.integers(0, keys, int(hot.sum())) * (cap // 7) + 5
    return s


@pytest.mark.parametrize("piped", [True, False], ids=["pipelined", "in_order"])
def test_a_stream_that_switches_between_uniform_and_skewed_batches(piped):
    """uniform batches put the engine on the range path; the skewed batches that follow are grouped by it until the hint has
    caught up (oversized ranges: k_finish's pass through global memory), then by the LSD passes; then uniform again"""
    rng = np.random.default_rng(5)
    cap, n = 400_000, 150_000
    kinds = "uuuuuu" + "ssssssss" + "uuuuuuuuuu" + "sususususu" + "uuuuuuuuuuuu"
    batches = [(_uniform if k == "u" else _skewed)(rng, cap, n) for k in kinds]
    c = _run(cap, batches, piped)
    assert 0 < c["denied"] < c["total"]


def test_skew_inside_one_range_many_keys():
    """70 % of a batch in ONE range but spread over 3 000 keys: the oversized range is a real sort, not one run"""
    rng = np.random.default_rng(6)
    cap, n = 400_000, 120_000
    batches = []
    for i in range(14):
        s = _uniform(rng, cap, n)
        if i >= 6:
            m = rng.random(n) < 0.7
            s[m] = cap // 3 + rng.integers(0, 3000, int(m.sum()))
        batches.append(s)
    _run(cap, batches, True)
    _run(cap, batches, False, fixed=False)


@pytest.mark.parametrize("cap", [65_537, 70_001, 1_000_003, 16_000_000, 16_777_215, 20_000_000])
def test_key_spaces_at_the_edges_of_the_range_path(cap):
    """65 536 keys and fewer, or ranges wider than 65 536 slots (> 16.7 M keys), are sorted by the LSD passes; everything in
    between takes the range path -- with one digit of the offset (ranges of at most 256 slots) or two"""
    rng = np.random.default_rng(cap % 1000)
    n = 60_000
    batches = [_uniform(rng, cap, n) for _ in range(6)]
    batches[3][: n // 2] = cap - 1          # the last slot, many times
    batches[4][::3] = 0
    _run(cap, batches, True)


def test_batch_sizes_tiny_ragged_and_beyond_the_path():
    rng = np.random.default_rng(8)
    cap = 300_000
    sizes = [1, 255, 256, 257, 4095, 4096, 4097, 70_001, 1, 300_000, 2_000_000, 1_600_000, 100_000, 100_000]
    batches = [_uniform(rng, cap, n) for n in sizes]
    _run(cap, batches, True)
    _run(cap, batches, False)


def test_slots_with_more_requests_than_a_byte_counter_holds():
    """the counting path of k_finish keeps a byte per slot and gives up at 16 requests of one slot in a range (ballot passes
    instead): slots with 17, 100, 300 and 5 000 requests among uniform ones, below the share that sends the stream to the LSD passes"""
    rng = np.random.default_rng(9)
    cap, n = 2_000_000, 400_000
    batches = []
    for i in range(10):
        s = _uniform(rng, cap, n)
        for j, cnt in enumerate((17, 100, 300, 5000 if i % 2 else 16)):
            at = rng.integers(0, n, cnt)
            s[at] = 123_457 * (j + 1) + i
        batches.append(s)
    _run(cap, batches, True, plan=(100, 1000, 3600))
    _run(cap, batches, True, general=True)


def test_fresh_in_order_engine_reaches_the_range_path_and_leaves_it_again():
    """The range hint is written by the grouping kernels of the sort paths; the bucket path, which in-order batches of 16 Ki
    requests or more take when the range path turns them down, writes none.  An engine that is only ever handed in-order batches
    (the ABI's default) therefore never got a hint and never left the bucket path.  Now its first batch, and every 32nd one the
    range path turned down, is grouped by the sort path alone.  Here: which kernels a fresh engine's in-order batches run
    (the engine's per-stage profile), uniform batches first, then skewed ones, then uniform again -- every batch exact."""
    import torch

    import throttlecrab_amd as t
    cap, n, plan = 2_000_000, 1 << 16, (5, 50, 60)
    eng = t.Engine(cap, n, fixed_params=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*plan)
    orc = _oracle(cap)
    rng = np.random.default_rng(11)
    step = [0]

    def run(slots):
        now = T0 + step[0] * 300_000_000
        step[0] += 1
        ref = orc.batch_slots(slots, *plan, 1, now)
        res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=now,
                                         want=("allowed", "remaining"))
        torch.cuda.synchronize()
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed) and np.array_equal(res.remaining.cpu().numpy(), ref.remaining), step[0]

    def stages(batches):
        eng.profile_enable(True)
        for b in batches:
            run(b)
        pr = {k: calls for k, (ms, calls) in eng.profile_read().items() if calls}
        eng.profile_enable(False)
        return pr

    for _ in range(3):
        run(_uniform(rng, cap, n))
    pr = stages([_uniform(rng, cap, n) for _ in range(4)])
    assert pr.get("sort") == 8 and not any(k.startswith("bucket") for k in pr), pr          # k_tile_ranges + k_finish per batch
    for _ in range(10):   # a skewed stream: the worst of the last eight looks keeps it off the range path
        run(_skewed(rng, cap, n))
    pr = stages([_skewed(rng, cap, n) for _ in range(4)])
    assert pr.get("sort", 0) != 8 or any(k.startswith("bucket") for k in pr), pr
    for _ in range(80):   # uniform again: noticed within 32 batches (+ eight looks), although the bucket path writes no hint
        run(_uniform(rng, cap, n))
    pr = stages([_uniform(rng, cap, n) for _ in range(4)])
    assert pr.get("sort") == 8 and not any(k.startswith("bucket") for k in pr), pr
    assert eng.selfcheck() == 0
    eng.close()


def _zipf_like(rng, cap, n, hot_keys=400, s=1.1):
    """a stream with a Zipf(s) head of `hot_keys` scattered keys taking ~2/3 of the requests, the rest uniform"""
    w = np.arange(1, hot_keys + 1, dtype=np.float64) ** (-s)
    head = (np.arange(hot_keys, dtype=np.int64) * 7_919 + 13) % cap
    out = rng.integers(0, cap, n).astype(np.uint32)
    m = rng.random(n) < 0.66
    out[m] = head[rng.choice(hot_keys, int(m.sum()), p=w / w.sum())]
    return out


@pytest.mark.parametrize("kind", ["decisions_only", "full_results", "timestamp_per_request", "decisions_only_wide",
                                  "decisions_only_interleaved", "decisions_only_wide_interleaved"])
def test_hot_slots_are_peeled_out_of_the_range_partition(kind, monkeypatch):
    """Round 6 (csrc/range_part.hpp): a stream whose skew is a few hot keys takes the range path with those keys' requests
    gathered behind the ranges.  Who is hot comes from the evaluations' notes on long runs, through pinned memory -- so the
    first batches go through the LSD passes, and the engine must say when it changed over.  Every batch exact: decisions only
    (the RANK form: the hot slots' requests are not even gathered, the evaluation's hot role ranks them where they stand and a
    last block stores the cells it parked -- on both layouts); batches that ask for full results or carry a timestamp per
    request stay on the LSD passes (a gather form for them was built, measured slower and removed: range_part.hpp); then the
    hot keys MOVE (the list is made afresh) and finally the stream turns uniform (the list empties).  The resident state is
    compared at the end."""
    if kind.endswith("_interleaved"):   # (the rank form over interleaved ranges: k_tile_part<PART_RANK, true>, k_finish<true> with its scan blocks)
        monkeypatch.setenv("TCGPU_RANGE_ILV", "1")
        kind = kind[:-len("_interleaved")]
    general, lean = kind == "timestamp_per_request", kind.startswith("decisions_only")
    import torch

    import throttlecrab_amd as t
    rng = np.random.default_rng(21)
    cap, n, plan = 3_000_000, 200_000, (20, 100, 60)
    eng = t.Engine(cap, n, fixed_params=kind != "decisions_only_wide")
    eng.use_torch_stream()
    eng.register_params_uniform(*plan)
    orc = _oracle(cap)
    paths, held = [], []
    streams = [lambda: _zipf_like(rng, cap, n)] * 30 + [lambda: (_zipf_like(rng, cap, n) + 1_000_003) % cap] * 30 + [lambda: _uniform(rng, cap, n)] * 60
    for i, mk in enumerate(streams):
        slots = mk().astype(np.uint32)
        now = T0 + i * 50_000_000
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        if general:
            nowc = now + np.sort(rng.integers(0, 10**6, n))
            ref = orc.batch_slots(slots, *plan, 1, nowc)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=torch.from_numpy(nowc).cuda(), want=("allowed",), inputs_ready=True)
        else:
            ref = orc.batch_slots(slots, *plan, 1, now)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=now, want=("allowed",) if lean else ("allowed", "remaining"), inputs_ready=True)
        paths.append(eng.info()["grouping_path"])
        held.append((res, ref, d))
        if i % 2 == 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for i, (res, ref, _) in enumerate(held):
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed), f"batch {i} ({paths[i]}): decisions differ"
        if not general and not lean:
            assert np.array_equal(res.remaining.cpu().numpy(), ref.remaining), f"batch {i} ({paths[i]}): remaining differs"
    info = eng.info()
    # the resident state after the last batch: one more request per slot of the first batch, full results
    probe = np.unique(streams[0]().astype(np.uint32))[:50_000]
    now = T0 + len(streams) * 50_000_000
    ref = orc.batch_slots(probe, *plan, 1, now)
    res = eng.rate_limit_batch_slots(torch.from_numpy(probe.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=now, want=("allowed", "remaining", "reset_after_ns"))
    torch.cuda.synchronize()
    assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed) and np.array_equal(res.remaining.cpu().numpy(), ref.remaining)
    assert np.array_equal(res.reset_after_ns.cpu().numpy(), ref.reset_after_ns)
    assert eng.selfcheck() == 0
    eng.close()
    hot = "range path, hot slots peeled"
    if lean:
        assert paths[:30].count(hot) >= 15 and paths[30:60].count(hot) >= 10, paths
        assert info["hot_batches"] >= 25
    else:
        assert paths[:60].count(hot) == 0 and paths[:60].count("LSD passes") >= 55, paths
    assert paths[-8:] == ["range path"] * 8 and info["hot_slots"] == 0, (paths[-20:], info)


def _burst(rng, cap, n, share=0.35):
    """a third of the batch on CONSECUTIVE slots (what a key table's free stack hands to one batch's new keys), a tenth on a
    stretch of 1 000 neighbouring slots with ~150 requests each (a benchmark's hot keys), the rest anywhere"""
    s = rng.integers(0, cap, n).astype(np.uint32)
    m = int(n * share)
    first = int(rng.integers(0, max(1, cap - m)))
    s[:m] = (first + np.arange(m)).astype(np.uint32)
    h = n // 10
    s[m:m + h] = (int(rng.integers(0, max(1, cap - 1000))) + rng.integers(0, min(1000, cap), h)).astype(np.uint32)
    return rng.permutation(s)


@pytest.mark.parametrize("cap", [65_537, 70_001, 1_000_003, 11_534_336, 20_000_000])
@pytest.mark.parametrize("general", [False, True], ids=["one_timestamp", "timestamp_per_request"])
def test_interleaved_ranges_of_a_string_mode_engine(cap, general, monkeypatch):
    """String mode (round 6, late): the 512 key ranges are INTERLEAVED chunk by chunk (radix_sort.hpp: RANGE_ILV -- slot s lies in
    range (s >> 3) mod 512) so that the neighbouring slots a key table hands out spread over all ranges.  A string-mode engine
    takes no slot batches; TCGPU_RANGE_ILV=1 gives a slot engine the same ranges, and its batches go through the same grouping
    kernels (k_tile_part<., true>, k_finish<true>, k_hist's range row): bursts on
    consecutive slots, a hot stretch, uniform batches, slots with more requests than a byte counter holds, out-of-range slots,
    ragged sizes -- each against the oracle in queue order; pipelined batches of such a stream take the range path."""
    import torch

    import throttlecrab_amd as t
    from tests.test_gpu_slots import _oracle as dense
    rng = np.random.default_rng(cap % 1000 + general)
    n = 1 << 17
    monkeypatch.setenv("TCGPU_RANGE_ILV", "1")
    eng = t.Engine(cap, n)
    eng.use_torch_stream()
    orc = dense(cap)
    plan = (5, 50, 60)
    batches = [_burst(rng, cap, n), _burst(rng, cap, n), _uniform(rng, cap, n), _burst(rng, cap, n - 4097), _burst(rng, cap, n)]
    many = _burst(rng, cap, n)
    many[rng.random(n) < 0.02] = np.uint32(cap // 3)                       # one slot, ~2 600 requests: past the byte counters
    bad = _burst(rng, cap, n)
    bad[rng.random(n) < 0.01] = np.uint32(cap) + rng.integers(0, 4)         # out-of-range slots: status Internal, nothing applied
    batches += [many, bad, _burst(rng, cap, 300), _burst(rng, cap, n)]
    held, paths = [], []
    for i, slots in enumerate(batches):
        now = T0 + i * 700_000_000
        d = torch.from_numpy(slots.astype(np.int64).astype(np.int32)).cuda()
        if general:
            nowc = now + np.sort(rng.integers(0, 10**6, len(slots)))
            ref = orc.batch_slots(slots, *plan, 1, nowc)
            res = eng.rate_limit_batch_slots(d, max_burst=plan[0], count_per_period=plan[1], period=plan[2], quantity=1,
                                             now_ns=torch.from_numpy(nowc).cuda(), want=("allowed", "remaining", "status"), inputs_ready=True)
        else:
            ref = orc.batch_slots(slots, *plan, 1, now)
            res = eng.rate_limit_batch_slots(d, max_burst=plan[0], count_per_period=plan[1], period=plan[2], quantity=1, now_ns=now,
                                             want=("allowed", "remaining", "status"), inputs_ready=True)
        paths.append(eng.info()["grouping_path"])
        held.append((res, ref, d))
        torch.cuda.synchronize()   # (lets the hint of the batches so far reach the host)
    for i, (res, ref, _) in enumerate(held):
        assert np.array_equal(res.status.cpu().numpy(), ref.status), f"batch {i}: statuses differ"
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed), f"batch {i}: decisions differ"
        assert np.array_equal(res.remaining.cpu().numpy(), ref.remaining), f"batch {i}: remaining differs"
    assert eng.selfcheck() == 0
    # the first batch has no hint to go by; the bursts behind it are taken by the range path (300 requests: below the path's 256 .. no:
    # it takes them; the batch with 2 600 requests on one slot may send its successor back to the passes for a while)
    assert paths[0] == "LSD passes" and paths[1] == "range path" and paths[3] == "range path", paths
    eng.close()
