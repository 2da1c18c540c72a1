"""Denied-key metrics on the device (throttlecrab-server/src/metrics.rs:24-76 analogue): the
per-key denial counters the evaluation kernels keep must equal the denials the oracle reports,
for every evaluation path, and tc_top_denied must return the K most denied keys."""
import collections

import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]


def _top_expected(counter, k):
    items = [(key, c) for key, c in counter.items() if c > 0]
    items.sort(key=lambda kv: (-kv[1], kv[0]))
    return items[:k]


@pytest.mark.parametrize("mode", ["uniform", "general", "unique", "irregular"])
def test_denied_counts_match_oracle_slot_mode(mode):
    import throttlecrab_amd as t
    from oracle import oracle as O
    cap, n = 3000, 60000
    rng = np.random.default_rng(hash(mode) % 1000)
    eng, orc = t.Engine(cap, n, track_denied=True), O.DenseOracle(cap)
    want = collections.Counter()
    for rnd in range(4):
        if mode == "unique":
            slots = rng.permutation(cap)[:2000].astype(np.uint32)
        else:
            slots = ((rng.zipf(1.3, n) * 2654435761) % cap).astype(np.uint32)
        m = slots.size
        if mode == "general":
            q, now = rng.integers(0, 3, m), T0 + rnd * 10**9 + rng.integers(0, 10**9, m)
            burst, count, period = 5, 10, 60
        elif mode == "irregular":
            q, now = 1, T0 + rnd * 10**8          # burst 1: entries expire as they are written -> run walked by its head
            burst, count, period = 1, 1, 1
            slots = (slots % 50).astype(np.uint32)
            m = slots.size
        else:
            q, now = 1, T0 + rnd * 10**9
            burst, count, period = 3, 10, 60
        if mode == "unique":
            q, now = rng.integers(1, 6, m), T0 + rnd * 10**8
            burst = 2
        ref = orc.batch_slots(slots, burst, count, period, q, now)
        res = eng.rate_limit_batch_slots(slots, max_burst=burst, count_per_period=count, period=period, quantity=q,
                                         now_ns=now, unique=(mode == "unique"), want=("allowed", "status"))
        assert np.array_equal(res.allowed, ref.allowed)
        denied = (ref.status == 0) & (ref.allowed == 0)
        for s in slots[denied].tolist():
            want[s] += 1
    for k in (1, 10, 100, 10000):
        got = eng.top_denied(k)
        exp = _top_expected(want, k)
        assert got == exp, (mode, k, got[:5], exp[:5])
    assert eng.counters()["denied"] == sum(want.values())
    eng.denied_reset()
    assert eng.top_denied(10) == []
    eng.close()


def test_top_denied_keys_string_mode():
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    keys = [b"user:%d" % i for i in range(2000)] + [b"very-long-key/" + b"z" * 130 + b"/%d" % i for i in range(20)]
    eng = t.Engine(8192, 50000, key_mode=True, track_denied=True)
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    want = collections.Counter()
    for rnd in range(3):
        idx = np.minimum(rng.zipf(1.2, 40000) - 1, len(keys) - 1)
        idx[:200] = len(keys) - 1 - rng.integers(0, 20, 200)
        kb, ko = O.pack_keys([keys[i] for i in idx])
        now = T0 + rnd * 10**8
        ref = orc.batch_keys(kb, ko, 4, 10, 60, 1, now)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=4, count_per_period=10, period=60, quantity=1, now_ns=now,
                                        want=("allowed", "status"))
        assert np.array_equal(res.allowed, ref.allowed)
        for i in idx[(ref.allowed == 0) & (ref.status == 0)].tolist():
            want[keys[i]] += 1
    for k in (1, 25, 3000):
        assert eng.top_denied(k) == _top_expected(want, k), k   # most denied first, ties by key bytes
    assert any(len(k) > 112 for k, _ in eng.top_denied(3000))   # keys beyond the inline 112 bytes come back intact
    # the counts belong to the KEYS: a sweep that unbinds every key changes nothing (metrics.rs:24-76 counts by key,
    # whatever the store cleans up) ...
    eng.sweep_expired(T0 + 10**12)
    assert eng.counters()["live_slots"] == 0
    assert eng.top_denied(25) == _top_expected(want, 25)
    # ... and tc_denied_reset forgets them all
    eng.denied_reset()
    assert eng.top_denied(10) == []
    eng.close()


class _TopDeniedModel:
    """TopDeniedKeys (throttlecrab-server/src/metrics.rs:24-76) without its size cap: a count per key, bumped on every
    denial, independent of the store; keys over MAX_KEY_LENGTH = 256 bytes are not tracked (metrics.rs:36-39)."""

    def __init__(self):
        self.counts = collections.Counter()

    def update(self, key):
        if len(key) <= 256:
            self.counts[key] += 1

    def get_top(self, k):
        return _top_expected(self.counts, k)


def test_top_denied_follows_keys_through_sweeps_and_rebinding():
    """generations of keys are denied, expire, are swept (their slots go to other keys), come back and
    are denied again; a key's count must be the sum over all its lives, exactly as the reference's map has it."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    pool = [b"k%d" % i for i in range(600)] + [b"tenant/%d/" % i + b"q" * (20 + i % 120) for i in range(200)] + [b"huge/" + b"h" * 300]
    eng = t.Engine(1024, 30000, key_mode=True, track_denied=True)   # fewer slots than keys: slots are reused across generations
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    model = _TopDeniedModel()
    now = T0
    for gen in range(6):
        live = rng.permutation(len(pool))[:700]                      # this generation's keys (overlaps with earlier ones)
        for rnd in range(2):
            idx = live[np.minimum(rng.zipf(1.3, 20000) - 1, len(live) - 1)]
            kb, ko = O.pack_keys([pool[i] for i in idx])
            ref = orc.batch_keys(kb, ko, 3, 6, 60, 1, now)
            res = eng.rate_limit_batch_keys(kb, ko, max_burst=3, count_per_period=6, period=60, quantity=1, now_ns=now,
                                            want=("allowed", "status"), inputs_ready=False)
            assert np.array_equal(res.allowed, ref.allowed), (gen, rnd)
            for i in idx[(ref.allowed == 0) & (ref.status == 0)].tolist():
                model.update(pool[i])
            now += 10**8
        for k in (5, 50, 1000):
            assert eng.top_denied(k) == model.get_top(k), (gen, k)
        now += 200 * 10**9                                           # everything expires ...
        orc.force_cleanup(now)
        assert eng.sweep_expired(now) > 0                            # ... and is swept: every key loses its slot
        assert eng.counters()["live_slots"] == 0
        for k in (5, 50, 1000):
            assert eng.top_denied(k) == model.get_top(k), ("after sweep", gen, k)
    assert model.counts[pool[-1]] == 0                               # (the 305-byte key was denied but is not tracked)
    eng.close()


def test_top_denied_needs_the_flag():
    import throttlecrab_amd as t
    eng = t.Engine(16, 16)
    with pytest.raises(t.TcError) as ei:
        eng.top_denied(5)
    assert ei.value.code == -7
    eng.close()


def test_retired_keys_survive_tombstone_churn_and_compaction():
    """the side table of keys that lost their slot is compacted by the host (tombstones out) before probe chains
    run full -- generations of 9 000 denied keys each retire and come back (every return leaves a tombstone, every retirement
    claims a record): 45 000 claims pass the statistics' threshold; no denial count may be lost on the way."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    n_keys = 9000
    pool = [b"gen-key-%d" % i for i in range(n_keys)]
    eng = t.Engine(16384, 40000, key_mode=True, track_denied=True)
    model = _TopDeniedModel()
    now = T0
    kb, ko = O.pack_keys(pool + pool + pool)   # every key three times per batch: burst 2 -> one denial per key per batch
    for gen in range(6):
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=2, count_per_period=2, period=3600, quantity=1, now_ns=now, want=("allowed", "status"))
        denied = (res.allowed == 0) & (res.status == 0)
        assert int(denied.sum()) == n_keys
        for k in pool:
            model.update(k)
        now += 2 * 3600 * 10**9
        assert eng.sweep_expired(now) == n_keys          # all retire: counts move into the side table (and back next round)
        assert eng.counters()["live_slots"] == 0
        assert eng.top_denied(50) == model.get_top(50), gen
    top = eng.top_denied(10000)
    assert len(top) == n_keys and all(c == 6 for _, c in top)
    assert [k for k, _ in top] == sorted(pool)            # equal counts: key bytes ascending
    eng.close()


def test_top_denied_keys_break_ties_at_the_cut_by_key_bytes():
    """tc_top_denied orders ties by slot, tc_top_denied_keys by key bytes: with 300 keys tied at the cut the k
    returned must be the smallest keys, not the lowest slots (keys are bound in an order unrelated to their bytes)."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    names = [b"tie-%05d" % i for i in rng.permutation(300)]          # bound in shuffled order: slot order != key order
    big = [b"big-%d" % i for i in range(5)]
    eng = t.Engine(4096, 8192, key_mode=True, track_denied=True)
    keys = []
    for nm in names:
        keys += [nm] * 3                                             # burst 2: one denial each
    for j, nm in enumerate(big):
        keys += [nm] * (4 + j)                                       # 2 + j denials
    kb, ko = O.pack_keys(keys)
    eng.rate_limit_batch_keys(kb, ko, max_burst=2, count_per_period=2, period=3600, quantity=1, now_ns=T0, want=("allowed",))
    want = [(b"big-4", 6), (b"big-3", 5), (b"big-2", 4), (b"big-1", 3), (b"big-0", 2)] + [(nm, 1) for nm in sorted(names)]
    for k in (3, 5, 10, 70, 200, 305, 400):
        assert eng.top_denied(k) == want[:k], k
    eng.close()
