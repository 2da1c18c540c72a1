"""TC_B_PLAN_DICT (round 6): the reference-shaped batch -- rate_limit(key, max_burst, count_per_period, period,
quantity, now) per request, rate_limiter.rs:102-110 -- with its (burst, count, period) triples as 16-bit indices into a dictionary
the batch carries and its quantities as u32: 6 bytes per request over PCIe instead of 32.  The SHIMS encode (here
Engine.encode_plans; rust/throttlecrab-gpu batch_chunk, include/throttlecrab_gpu.hpp submit_batch); the results must be the wide
form's, request by request, on every path a batch can take: the one-launch small batch, synchronous batches from pageable and
pinned arrays, the chunked pipeline, TC_B_ASYNC rings, device-pointer batches; slots and string keys; the retry of rejected
requests; indices beyond the dictionary (status InvalidRateLimit, like any non-positive triple)."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
S = 10**9
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")
PLANS = np.array([(5, 10, 60), (100, 1000, 3600), (1, 10, 1), (40, 7, 3), (2, 2, 2), (0, 10, 60)], dtype=np.int64)   # (the last: an invalid triple)


@pytest.fixture(autouse=True)
def small_chunks(monkeypatch):
    monkeypatch.setenv("TCGPU_HOST_CHUNK", "65536")


def _stream(rng, n, n_keys, rnd):
    ids = rng.integers(0, n_keys, n)
    pl = PLANS[(ids * 7 + rnd) % len(PLANS)]
    q = rng.integers(0, 3, n).astype(np.int64)
    now = (T0 + rnd * S + np.sort(rng.integers(0, S, n))).astype(np.int64)
    return ids, pl, q, now


def _same(res, ref, ctx):
    for f in FIELDS:
        got = getattr(res, f)
        got = got.cpu().numpy() if hasattr(got, "cpu") else got
        bad = np.nonzero(got.astype(np.int64) != getattr(ref, f).astype(np.int64))[0]
        assert bad.size == 0, f"{ctx}: {f} differs at {bad[:8]}"


@pytest.mark.parametrize("n", [700, 5000, 200_001])
@pytest.mark.parametrize("mode", ["pageable", "pinned", "async_ring", "device"])
def test_slot_batches_in_the_dictionary_form(n, mode):
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    cap = 30_000
    eng, orc = t.Engine(cap, 1 << 18), O.DenseOracle(cap)
    eng.check_on_close = True
    rng = np.random.default_rng(n)
    held = []
    for rnd in range(4):
        ids, pl, q, now = _stream(rng, n, cap, rnd)
        slots = ids.astype(np.uint32)
        ref = orc.batch_slots(slots, pl[:, 0], pl[:, 1], pl[:, 2], q, now, threads=O.host_threads())
        dic, pid = t.Engine.encode_plans(pl[:, 0], pl[:, 1], pl[:, 2])
        assert len(dic) <= len(PLANS)
        q32 = q.astype(np.uint32)
        if mode == "device":
            d = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype != dt else a).cuda()   # noqa: E731
            res = eng.rate_limit_batch_slots(d(slots.astype(np.int32), np.int32), now_ns=d(now, np.int64), want=FIELDS,
                                             plan_dict=(d(dic, np.int64), d(pid.view(np.int16), np.int16), d(q32.view(np.int32), np.int32)))
            torch.cuda.synchronize()
            _same(res, ref, f"round {rnd}")
            continue
        if mode == "pageable":
            res = eng.rate_limit_batch_slots(slots, now_ns=now, want=FIELDS, plan_dict=(dic, pid, q32))
            _same(res, ref, f"round {rnd}")
            continue
        pin = lambda a: _pinned(eng, a)   # noqa: E731
        out = t.BatchResult(**{f: eng.host_alloc(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in FIELDS})
        args = dict(now_ns=pin(now), want=FIELDS, out=out, plan_dict=(pin(dic.reshape(-1)), pin(pid), pin(q32)))
        if mode == "pinned":
            res = eng.rate_limit_batch_slots(pin(slots), **args)
            _same(res, ref, f"round {rnd}")
        else:
            res = eng.rate_limit_batch_slots(pin(slots), async_=True, **args)
            held.append((res, ref, args))
    if held:
        eng.wait_batches()
        for i, (res, ref, _) in enumerate(held):
            _same(res, ref, f"async batch {i}")
    eng.close()


def _pinned(eng, a):
    h = eng.host_alloc(a.size, a.dtype)
    h[:] = a.reshape(-1)
    return h


@pytest.mark.parametrize("n", [900, 4096, 150_000])
@pytest.mark.parametrize("mode", ["pageable", "pinned", "async_ring"])
def test_key_batches_in_the_dictionary_form(n, mode):
    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    eng = t.Engine(60_000, 1 << 18, key_mode=True)
    eng.check_on_close = True
    orc = O.AdaptiveOracle(capacity=60_000, created_ns=T0, auto_cleanup=False)
    rng = np.random.default_rng(n + 1)
    held = []
    for rnd in range(3):
        ids, pl, q, now = _stream(rng, n, 20_000, rnd)
        kb, ko = W.string_keys(ids, prefix=b"user:")
        ref = orc.batch_keys(kb, ko, pl[:, 0], pl[:, 1], pl[:, 2], q, now)
        dic, pid = t.Engine.encode_plans(pl[:, 0], pl[:, 1], pl[:, 2])
        q32 = q.astype(np.uint32)
        if mode == "pageable":
            res = eng.rate_limit_batch_keys(kb, ko, now_ns=now, want=FIELDS + ("decisions",), plan_dict=(dic, pid, q32))
            _same(res, ref, f"round {rnd}")
            continue
        pin = lambda a: _pinned(eng, a)   # noqa: E731
        out = t.BatchResult(**{f: eng.host_alloc(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in FIELDS})
        args = dict(now_ns=pin(now), want=FIELDS, out=out, plan_dict=(pin(dic.reshape(-1)), pin(pid), pin(q32)))
        if mode == "pinned":
            _same(eng.rate_limit_batch_keys(pin(kb), pin(ko), **args), ref, f"round {rnd}")
        else:
            held.append((eng.rate_limit_batch_keys(pin(kb), pin(ko), async_=True, **args), ref, args))
    if held:
        eng.wait_batches()
        for i, (res, ref, _) in enumerate(held):
            _same(res, ref, f"async batch {i}")
    assert eng.debug_check_keys() == 0
    eng.close()


def test_an_index_beyond_the_dictionary_and_inconsistent_batches():
    import throttlecrab_amd as t
    from oracle import oracle as O
    cap, n = 5000, 3000
    eng, orc = t.Engine(cap, 1 << 16), O.DenseOracle(cap)
    rng = np.random.default_rng(2)
    slots = rng.integers(0, cap, n).astype(np.uint32)
    dic = PLANS[:3].copy()
    pid = rng.integers(0, 5, n).astype(np.uint16)          # 3 and 4 are beyond the dictionary: (0, 0, 0)
    wide = np.where((pid < 3)[:, None], dic[np.minimum(pid, 2)], 0)
    ref = orc.batch_slots(slots, wide[:, 0], wide[:, 1], wide[:, 2], 1, T0)
    res = eng.rate_limit_batch_slots(slots, now_ns=T0, quantity=1, want=FIELDS, plan_dict=(dic, pid, None))
    _same(res, ref, "ids beyond the dictionary")
    assert (res.status[pid >= 3] == 2).all()               # TC_INVALID_RATE_LIMIT
    total = eng.counters()["total"]
    with pytest.raises(ValueError):
        eng.rate_limit_batch_slots(slots, now_ns=T0, max_burst=np.full(n, 5), count_per_period=10, period=60, plan_dict=(dic, pid, None))
    with pytest.raises(t.TcError):                          # a dictionary of no plans
        eng.rate_limit_batch_slots(slots, now_ns=T0, want=FIELDS, plan_dict=(np.zeros((0, 3), np.int64), pid, None))
    assert eng.counters()["total"] == total and eng.selfcheck() == 0
    eng.close()


def test_rejected_requests_of_a_dictionary_batch_are_applied_again_after_the_sweep():
    """the store cleans itself (tc_set_sweep_policy): a synchronous key batch that ran out of slots sweeps and applies the
    rejected requests once more -- from the dictionary form as from the wide one"""
    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    capn, n = 4096, 3000
    eng = t.Engine(capn, 1 << 14, key_mode=True)
    eng.set_sweep_policy("adaptive", created_ns=T0)
    orc = O.AdaptiveOracle(capacity=1 << 20, created_ns=T0, auto_cleanup=False)
    plan = np.array([(3, 30, 1)], dtype=np.int64)           # entries live for about a second
    for rnd in range(6):                                      # 6 x 3 000 new keys through a table of 4 096: every round needs the room of the one before
        ids = np.arange(rnd * n, (rnd + 1) * n)
        kb, ko = W.string_keys(ids, prefix=b"r:")
        now = np.full(n, T0 + rnd * 5 * S, np.int64)
        ref = orc.batch_keys(kb, ko, 3, 30, 1, 1, now)
        res = eng.rate_limit_batch_keys(kb, ko, now_ns=now, want=FIELDS, plan_dict=(plan, np.zeros(n, np.uint16), np.ones(n, np.uint32)))
        _same(res, ref, f"round {rnd}")
    assert eng.counters()["errors"] == 0
    eng.close()
