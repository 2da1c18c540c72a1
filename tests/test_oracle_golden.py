"""Pins the CPU oracle (oracle/gcra_oracle.c) against the reference's own
known-answer tests (tests/golden/reference_kat.json)."""
import pytest

from oracle import oracle as O
from tests import kat

KAT = kat.load()
T0 = KAT["t0_ns"]


@pytest.mark.parametrize("sc", KAT["scenarios"], ids=[s["name"] for s in KAT["scenarios"]])
def test_scenario_adaptive_store(sc):
    kat.replay_scenario(sc, O.AdaptiveOracle(capacity=1000, created_ns=T0))


@pytest.mark.parametrize("case", KAT["store_contract"], ids=[c["name"] for c in KAT["store_contract"]])
def test_store_contract_adaptive(case):
    kat.replay_store_contract(case, O.AdaptiveOracle(capacity=100, created_ns=T0), T0)


def test_rate_from_count_and_period():
    for c in KAT["rates"]["cases"]:
        assert O.emission_interval(c["count"], c["period"]) == c["period_ns"]


def test_exact_numbers_survey_appendix_b():
    """Exact values implied by the reference's arithmetic (rate_limiter.rs:207-238)."""
    lim = O.AdaptiveOracle(created_ns=T0)
    # (10,100,60): ei = 0.6 s, dvt = 5.4 s; first request: reset_after = 5.4 s -> 5
    st, allowed, limit, rem, reset_ns, retry_ns = lim.rate_limit(b"k", 10, 100, 60, 1, T0)
    assert (st, allowed, limit, rem, reset_ns, retry_ns) == (0, True, 10, 9, 5_400_000_000, 0)
    # burst exhausted -> retry_after = ei exactly for (5,10,60): 6 s
    lim = O.AdaptiveOracle(created_ns=T0)
    for _ in range(5):
        lim.rate_limit(b"b", 5, 10, 60, 1, T0)
    st, allowed, limit, rem, reset_ns, retry_ns = lim.rate_limit(b"b", 5, 10, 60, 1, T0)
    assert (allowed, rem, retry_ns) == (False, 0, 6_000_000_000)
    # burst 10 then 5 on one key -> allowed, remaining exactly 3 (clamp observable)
    lim = O.AdaptiveOracle(created_ns=T0)
    lim.rate_limit(b"d", 10, 100, 60, 1, T0)
    st, allowed, limit, rem, *_ = lim.rate_limit(b"d", 5, 100, 60, 1, T0)
    assert (allowed, limit, rem) == (True, 5, 3)
    # (3,7,60): ei = 8 571 428 571 ns
    assert O.emission_interval(7, 60) == 8_571_428_571


def test_error_precedence_and_state_untouched():
    lim = O.AdaptiveOracle(created_ns=T0)
    assert lim.rate_limit(b"e", 0, 0, 0, -1, T0)[0] == O.NEGATIVE_QUANTITY  # q<0 checked first
    assert lim.rate_limit(b"e", 0, 10, 60, 1, T0)[0] == O.INVALID_RATE_LIMIT
    assert len(lim) == 0
    assert lim.get(b"e", T0) is None


def test_burst_one_expires_immediately():
    """burst=1 => dvt=0, ttl=0 => entry dead at the same `now` (expiry > now is strict,
    adaptive_cleanup.rs:248): every request at one timestamp is allowed."""
    lim = O.AdaptiveOracle(created_ns=T0)
    for _ in range(4):
        st, allowed, limit, rem, reset_ns, retry_ns = lim.rate_limit(b"one", 1, 1, 1, 1, T0)
        assert (st, allowed, limit, rem) == (0, True, 1, 0)


def test_zero_quantity_burst_one_never_expires():
    """q=0 on a fresh burst=1 key: ttl = (i64)(-ei) as u64 ~ 584 y (rate_limiter.rs:179-183)."""
    lim = O.AdaptiveOracle(created_ns=T0)
    assert lim.rate_limit(b"z", 1, 1, 1, 0, T0)[1] is True
    assert lim.get(b"z", T0 + 200 * 365 * 86400 * 10**9) == T0 - 10**9
    # the key is now live forever with tat clamped to `now`: q=1 is denied
    st, allowed, *_ = lim.rate_limit(b"z", 1, 1, 1, 1, T0 + 5 * 10**9)
    assert (st, allowed) == (0, False)


def test_internal_domain():
    lim = O.AdaptiveOracle(created_ns=T0)
    # pre-1970 now -> reference reads the wall clock; we return Internal
    assert lim.rate_limit(b"i", 5, 10, 60, 1, -1)[0] == O.INTERNAL
    # Duration * u32 overflow (reference panics): ei saturates to u64::MAX ns, mult ~ 4.29e9
    assert lim.rate_limit(b"i", 2**32, 1, 2**62, 1, T0)[0] == O.INTERNAL
    assert len(lim) == 0


def test_adaptive_cleanup_runs_and_is_decision_neutral():
    a = O.AdaptiveOracle(capacity=1000, created_ns=T0, max_operations=50)
    d = O.DenseOracle(4096)
    import numpy as np
    rng = np.random.default_rng(7)
    slots = rng.integers(0, 300, 5000).astype(np.uint32)
    for i, s in enumerate(slots):
        now = T0 + i * 20_000_000
        ka = a.rate_limit(b"key_%d" % s, 3, 30, 60, 1, now)
        kd = d.rate_limit(int(s).to_bytes(4, "little"), 3, 30, 60, 1, now)
        assert ka == kd
    assert a.cleanups > 10


def test_cleanup_is_not_neutral_when_time_goes_back():
    """AdaptiveStore::cleanup is decision-neutral only for non-decreasing timestamps:
    burst=1 leaves expiry == t1; a later request stamped t0 < t1 sees a LIVE entry
    (adaptive_cleanup.rs:248) and is denied -- unless a cleanup at t >= t1 dropped the
    entry first (adaptive_cleanup.rs:176-182), in which case it is allowed."""
    t1, t0 = T0 + 10**9, T0 + 10**9 - 217_000_000
    a = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)  # never cleans
    assert a.rate_limit(b"k", 1, 1, 1, 1, t1)[1] is True
    st, allowed, _, _, reset_ns, retry_ns = a.rate_limit(b"k", 1, 1, 1, 1, t0)
    assert (st, allowed, reset_ns, retry_ns) == (0, False, 217_000_000, 1_217_000_000)
    b = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    assert b.rate_limit(b"k", 1, 1, 1, 1, t1)[1] is True
    b.force_cleanup(t1)
    assert b.rate_limit(b"k", 1, 1, 1, 1, t0)[1] is True


def test_partitioned_dense_driver_equals_the_sequential_one():
    """tco_batch_slots_mt (requests partitioned by slot over threads: the checker of the full-size GPU tests) gives the
    results and the state of the one-by-one pass -- hot keys, per-request timestamps going back, error requests"""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    cap, n = 3000, 40_000
    a, b = O.DenseOracle(cap), O.DenseOracle(cap)
    t0 = KAT["t0_ns"]
    for rnd in range(4):
        slots = rng.integers(0, cap + 3, n).astype(np.uint32)            # (a few out of range: status Internal)
        hot = rng.random(n) < 0.6
        slots[hot] = rng.integers(0, 4, int(hot.sum()))
        now = t0 + rnd * 10**9 + rng.integers(-10**8, 10**9, n)
        q = rng.choice(np.array([0, 1, 1, 2, -1], dtype=np.int64), n)
        burst = rng.choice(np.array([1, 2, 5, 100, 0], dtype=np.int64), n)
        ra = a.batch_slots(slots, burst, 10, 60, q, now)
        rb = b.batch_slots(slots, burst, 10, 60, q, now, threads=7)
        for f, x in ra.fields().items():
            assert np.array_equal(x, rb.fields()[f]), (rnd, f)
    for x, y in zip(a.dump(), b.dump()):
        assert np.array_equal(x, y)
