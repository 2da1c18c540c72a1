"""GPU parity of the TC_CFG_FIXED_PARAMS layout (8 bytes per key: the stored TAT alone, SURVEY.md App. A
"fixed-params shortcut"): every evaluation path, the sweep, snapshots and tc_read_state must give exactly what
the 16-byte {tat, expiry} layout and the oracle give, as long as a key's plan never changes -- which the mode
enforces (registration before the first request, registered batches only)."""
import numpy as np
import pytest

from tests.test_gpu_slots import FIELDS, T0, _oracle, assert_same, assert_state_same

pytestmark = pytest.mark.gpu

PLANS = np.array([(100, 1000, 3600), (5, 10, 60), (3, 7, 60), (20, 600, 60), (2, 120, 60), (2**31, 10**9, 1)], dtype=np.int64)


def _fixed(capacity, max_batch, **kw):
    import throttlecrab_amd as t
    e = t.Engine(capacity, max_batch, fixed_params=True, **kw)
    e.check_on_close = True
    return e


def _plans_for(rng, cap, unregistered=0):
    idx = rng.integers(0, len(PLANS), cap)
    b, c, p = PLANS[idx, 0].copy(), PLANS[idx, 1].copy(), PLANS[idx, 2].copy()
    return idx, b, c, p


@pytest.fixture(params=["default", "bucket_all", "no_bucket"])
def grouping(request, monkeypatch):
    for k in ("TCGPU_BUCKET", "TCGPU_BUCKET_MIN_N", "TCGPU_BUCKET_SKEW"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "bucket_all":
        monkeypatch.setenv("TCGPU_BUCKET_MIN_N", "1")
    elif request.param == "no_bucket":
        monkeypatch.setenv("TCGPU_BUCKET", "0")
    return request.param


@pytest.mark.parametrize("plans", ["uniform", "per_slot"])
def test_uniform_batches(grouping, plans):
    """closed-form paths (sorted + direct / parked stores, bucket path), heavy duplicates, q = 0 / 1 / 3, time going back"""
    cap, n = 40_000, 50_000
    rng = np.random.default_rng(7 + (plans == "uniform"))
    eng, orc = _fixed(cap, n), _oracle(cap)
    if plans == "uniform":
        eng.register_params_uniform(5, 10, 60)
        b = c = p = None
    else:
        idx, b, c, p = _plans_for(rng, cap)
        eng.register_params(b[: cap - 100], c[: cap - 100], p[: cap - 100])  # the last 100 keys: no plan -> InvalidRateLimit
        b[cap - 100:] = c[cap - 100:] = p[cap - 100:] = 0
    z = rng.zipf(1.3, n).astype(np.int64)
    for rnd in range(6):
        slots = ((z * 2654435761 + rnd * 17) % 900).astype(np.uint32) if rnd % 2 == 0 else rng.integers(0, cap + 3, n).astype(np.uint32)
        now = T0 + rnd * 700_000_000 if rnd != 4 else T0 - 10**9
        q = (1, 1, 0, 3, 1, 1)[rnd]
        if plans == "uniform":
            ref = orc.batch_slots(slots, 5, 10, 60, q, now)
        else:
            # (an out-of-range slot has no plan to be invalid: the oracle needs a valid triple to get as far as its store)
            sc, out = np.minimum(slots, cap - 1), slots >= cap
            ref = orc.batch_slots(slots, np.where(out, 5, b[sc]), np.where(out, 10, c[sc]), np.where(out, 60, p[sc]), q, now)
        res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
        assert_same(res, ref, f"{grouping}/{plans} round {rnd}")
        assert_state_same(eng, orc, slots[::7])
    eng.close()


def test_general_unique_small_and_pipelined_batches(grouping):
    """per-request timestamps / quantities (k_eval_general), unique-slot batches, one-launch small batches and
    TC_B_INPUTS_READY device batches on one engine"""
    import torch
    cap, n = 30_000, 40_000
    rng = np.random.default_rng(11)
    eng, orc = _fixed(cap, n), _oracle(cap)
    eng.use_torch_stream()
    idx, b, c, p = _plans_for(rng, cap)
    eng.register_params(b, c, p)
    tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
    for rnd in range(6):
        kind = ("general", "unique", "small", "piped", "general_host", "piped")[rnd]
        base = T0 + rnd * 500_000_000
        if kind == "unique":
            slots = rng.permutation(cap)[:20_000].astype(np.uint32)
            ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], 1, base)
            res = eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=base, unique=True)
        elif kind == "small":
            slots = rng.integers(0, 50, 700).astype(np.uint32)
            q, now = rng.integers(0, 3, 700), base + rng.integers(0, 10**9, 700)
            ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], q, now)
            res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
        elif kind == "piped":
            slots = ((rng.zipf(1.2, n) * 2654435761) % cap).astype(np.uint32)
            ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], 1, base)
            res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=base,
                                             inputs_ready=True)
            torch.cuda.synchronize()
        else:
            slots = ((rng.zipf(1.3, n) * 2654435761) % 3000).astype(np.uint32)
            q = rng.choice(np.array([0, 1, 1, 2, -1, 2**62], dtype=np.int64), n)
            now = base + rng.integers(0, 10**9, n)
            now[rng.random(n) < 0.003] = -5
            ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], q, now)
            if kind == "general":
                res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=tt(q), now_ns=tt(now))
                torch.cuda.synchronize()
            else:
                res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
        assert_same(res, ref, f"{grouping} {kind}")
        assert_state_same(eng, orc, slots[::5])
    eng.close()


def test_sweep_and_snapshot(tmp_path):
    """tc_sweep_expired == cleanup on the 8-byte layout; a snapshot continues bit-identically"""
    cap, n = 20_000, 30_000
    rng = np.random.default_rng(5)
    eng, orc = _fixed(cap, n), _oracle(cap)
    idx, b, c, p = _plans_for(rng, cap)
    eng.register_params(b, c, p)
    now = T0
    for rnd in range(5):
        slots = rng.integers(0, cap, n).astype(np.uint32)
        now += rng.integers(1, 40) * 10**9
        ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], 1, now)
        assert_same(eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=now), ref, f"round {rnd}")
        if rnd % 2 == 1:
            assert eng.sweep_expired(now) == orc.sweep(now)
            assert eng.counters()["live_slots"] == orc.live()
            assert_state_same(eng, orc, np.arange(cap))
    path = str(tmp_path / "fixed.snap")
    eng.snapshot_save(path)
    eng2 = _fixed(cap, n)
    eng2.snapshot_load(path)
    import throttlecrab_amd as t
    with pytest.raises(t.engine.TcError):   # a 16-byte engine does not take an 8-byte snapshot
        wide = t.Engine(cap, n)
        try:
            wide.snapshot_load(path)
        finally:
            wide.close()
    slots = rng.integers(0, cap, n).astype(np.uint32)
    ref = orc.batch_slots(slots, b[slots], c[slots], p[slots], 1, now + 10**9)
    for e in (eng, eng2):
        assert_same(e.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=now + 10**9), ref, "after the snapshot")
        assert_state_same(e, orc, np.arange(0, cap, 3))
        e.close()


def test_mode_rules():
    """what the mode refuses: plans it cannot represent, late registration, unregistered batches, single calls,
    Store operations, string keys; and its stricter timestamp domain"""
    import throttlecrab_amd as t
    from throttlecrab_amd import _lib as L
    eng = _fixed(1000, 4096)
    for bad in ((1, 10, 60), (2**32 + 1, 10, 60), (10, 2**62, 60), (2**31 + 1, 1, 1)):  # burst 1; u32-wrapped burst; ei = 0; dvt >= 2^60
        with pytest.raises(t.engine.TcError) as ei:
            eng.register_params_uniform(*bad)
        assert ei.value.code == L.TC_E_UNSUPPORTED, bad
    eng.register_params_uniform(5, 10, 60)
    with pytest.raises(t.engine.TcError) as ei:
        eng.rate_limit_batch_slots(np.arange(10, dtype=np.uint32), max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0)
    assert ei.value.code == L.TC_E_UNSUPPORTED
    for call in (lambda: eng.rate_limit((3).to_bytes(4, "little"), 5, 10, 60, 1, T0),
                 lambda: eng.get((3).to_bytes(4, "little"), T0),
                 lambda: eng.set_if_not_exists_with_ttl((3).to_bytes(4, "little"), 1, 10, T0),
                 lambda: eng.compare_and_swap_with_ttl((3).to_bytes(4, "little"), 1, 2, 10, T0)):
        with pytest.raises(t.engine.TcError) as ei:
            call()
        assert ei.value.code == L.TC_E_UNSUPPORTED
    # timestamps: 2^62 - 1 is decided, 2^62 is Internal (the 16-byte layout takes it)
    orc = _oracle(1000)
    slots = np.arange(64, dtype=np.uint32)
    late = 2**62 - 1
    ref = orc.batch_slots(slots, 5, 10, 60, 1, late)
    assert_same(eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=late), ref, "2^62 - 1")
    res = eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=2**62)
    assert np.all(res.status == L.TC_INTERNAL) and not res.allowed.any()
    with pytest.raises(t.engine.TcError) as ei:   # sealed by the first request
        eng.register_params_uniform(3, 7, 60)
    assert ei.value.code == L.TC_E_UNSUPPORTED
    eng.close()
    with pytest.raises(t.engine.TcError) as ei:
        t.Engine(1000, 1024, key_mode=True, fixed_params=True)
    assert ei.value.code == L.TC_E_UNSUPPORTED


def test_same_results_as_the_wide_layout_at_full_size():
    """10 M keys, 1 Mi requests per batch, uniform and skewed, in order and pipelined: the two layouts agree on
    every decision and on tc_read_state"""
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W
    cap, n = 10_000_000, 1 << 20
    wide, fix = t.Engine(cap, n), _fixed(cap, n)
    z = W.Zipf(cap)
    for e in (wide, fix):
        e.use_torch_stream()
        e.register_params_uniform(*W.REF_PARAMS)
    for rnd in range(6):
        slots = W.uniform_slots(cap, n, start=rnd * n) if rnd % 2 == 0 else z.slots(n, start=rnd * n)
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        outs = [e.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=W.T0_NS + rnd * 10**6, want=("allowed", "remaining"),
                                         inputs_ready=(rnd >= 3)) for e in (wide, fix)]
        torch.cuda.synchronize()
        for f in ("allowed", "remaining"):
            assert torch.equal(getattr(outs[0], f), getattr(outs[1], f)), (rnd, f)
    a, b = wide.read_state(0, cap), fix.read_state(0, cap)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert wide.counters()["allowed"] == fix.counters()["allowed"]
    wide.close()
    fix.close()


def _fixed_eligible(sc):
    """A reference scenario can be replayed on the 8-byte layout when every key keeps ONE valid plan with burst >= 2
    and nothing in it expects an error (the layout takes registered plans only)."""
    plans = {}
    for st in sc["steps"]:
        if st["expect"].get("status") not in (None, 0) or st["q"] < 0:
            return None
        plan = (st["burst"], st["count"], st["period"])
        if plans.setdefault(st["key"], plan) != plan:
            return None
        b, c, p = plan
        if not (2 <= b < 2**31 and 0 < c and 0 < p and p * 10**9 // c > 0 and p * 10**9 // c * (b - 1) < 2**60 and p < 2**30):
            return None
    return plans


_KAT_FIXED = [s for s in __import__("tests.kat", fromlist=["load"]).load()["scenarios"] if _fixed_eligible(s)]


@pytest.mark.parametrize("sc", _KAT_FIXED, ids=[s["name"] for s in _KAT_FIXED])
def test_reference_known_answers_fixed_layout(sc):
    """the reference's own assertions (tests/golden/reference_kat.json) through the 8-byte TAT column"""
    from tests import kat
    plans = _fixed_eligible(sc)
    keys = sorted(plans)
    slot_of = {k: i for i, k in enumerate(keys)}
    eng = _fixed(max(64, len(keys)), 64)
    tr = np.array([plans[k] for k in keys], dtype=np.int64)
    eng.register_params(tr[:, 0], tr[:, 1], tr[:, 2], slots=np.arange(len(keys), dtype=np.uint32))

    class L:
        def rate_limit(self, key, burst, count, period, q, now):
            r = eng.rate_limit_batch_slots(np.array([slot_of[key.decode("utf-8")]], np.uint32), registered=True, quantity=q, now_ns=now,
                                           want=FIELDS)
            return (int(r.status[0]), bool(r.allowed[0]), int(r.limit[0]), int(r.remaining[0]), int(r.reset_after_ns[0]),
                    int(r.retry_after_ns[0]))
    kat.replay_scenario(sc, L())
    eng.close()


def test_most_reference_scenarios_fit_the_fixed_layout():
    assert len(_KAT_FIXED) >= 25, len(_KAT_FIXED)


@pytest.mark.parametrize("shape", ["steady_denials", "tokens_trickle_in", "jitter_and_errors", "expired_between"])
def test_general_decisions_only_hot_keys(shape):
    """decisions-only batches with a timestamp per request on the 8-byte layout: waves of a hot key's run that find the
    resident state live and deny everything under it do not wait for the state that reaches them (strong
    transparency, k_eval_general).  Hot keys whose runs cross hundreds of waves, tokens becoming available in the
    middle of a run, timestamps that jump back, quantities 0 / 1 / 2 / huge, error requests, entries that expire
    between batches -- decisions, statuses and the resident state must equal the oracle's after every batch."""
    import torch
    cap, n = 4_000, 60_000
    rng = np.random.default_rng({"steady_denials": 1, "tokens_trickle_in": 2, "jitter_and_errors": 3, "expired_between": 4}[shape])
    eng, orc = _fixed(cap, n), _oracle(cap)
    eng.use_torch_stream()
    plan = (5, 600, 60) if shape != "expired_between" else (3, 60, 60)   # ei = 0.1 s / 1 s
    eng.register_params_uniform(*plan)
    tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
    hot = rng.permutation(cap)[:6]
    base = T0
    for rnd in range(7):
        slots = rng.integers(0, cap, n).astype(np.uint32)
        heavy = rng.random(n) < 0.85
        slots[heavy] = hot[rng.integers(0, len(hot), int(heavy.sum()))]   # ~8 500 requests per hot key: 130+ waves each
        if shape == "steady_denials":
            base += 1_000_000
            now = base + np.arange(n, dtype=np.int64)                       # 60 us across the batch: no token in sight
            q = np.ones(n, np.int64)
        elif shape == "tokens_trickle_in":
            base += 50_000_000
            now = base + np.arange(n, dtype=np.int64) * 20_000              # 1.2 s across the batch: a dozen tokens per key
            q = np.ones(n, np.int64)
        elif shape == "jitter_and_errors":
            base += 300_000_000
            now = base + rng.integers(0, 400_000_000, n)
            now[rng.random(n) < 0.01] -= 3 * 10**9                         # the clock jumps back
            q = rng.choice(np.array([0, 1, 1, 1, 2, 7, -1, 2**61], dtype=np.int64), n)
        else:
            base += 9 * 10**9                                               # every entry has expired since the last batch
            now = base + np.sort(rng.integers(0, 3 * 10**9, n))
            q = rng.choice(np.array([1, 1, 2, 4], dtype=np.int64), n)       # 4 > burst: denied fresh, allowed on a stale live entry
        ref = orc.batch_slots(slots, *plan, q, now)
        res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=tt(q), now_ns=tt(now),
                                         want=("allowed", "status"), inputs_ready=bool(rnd % 2))
        torch.cuda.synchronize()
        assert_same(res, ref, f"{shape} round {rnd}")
        assert_state_same(eng, orc, np.concatenate([hot.astype(np.uint32), slots[::7]]))
    c = eng.counters()
    assert c["denied"] > c["allowed"] > 0
    eng.close()
