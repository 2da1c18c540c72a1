"""Checkpoint / restore: an engine restored from a snapshot must
continue exactly like the one that wrote it -- decisions, resident state, counters, denial
counters, registered plans and (string mode) the key table."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")


def _same(a, b):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_snapshot_slot_mode(tmp_path):
    import throttlecrab_amd as t
    cap, n = 5000, 30000
    rng = np.random.default_rng(4)
    plans = np.array([(5, 10, 60), (100, 1000, 3600), (3, 7, 60)], dtype=np.int64)[rng.integers(0, 3, cap)]
    a = t.Engine(cap, n, track_denied=True)
    a.register_params(plans[:, 0].copy(), plans[:, 1].copy(), plans[:, 2].copy())
    streams = [(rng.integers(0, cap, n).astype(np.uint32), T0 + i * 10**9) for i in range(6)]
    for s, now in streams[:3]:
        a.rate_limit_batch_slots(s, registered=True, quantity=1, now_ns=now)
    path = str(tmp_path / "slots.snap")
    a.snapshot_save(path)
    b = t.Engine(cap, n, track_denied=True)
    b.snapshot_load(path)
    assert a.counters() == b.counters()
    for s, now in streams[3:]:
        _same(a.rate_limit_batch_slots(s, registered=True, quantity=1, now_ns=now),
              b.rate_limit_batch_slots(s, registered=True, quantity=1, now_ns=now))
    ta, xa = a.read_state(0, cap)
    tb, xb = b.read_state(0, cap)
    assert np.array_equal(ta, tb) and np.array_equal(xa, xb)
    assert a.counters() == b.counters() and a.top_denied(50) == b.top_denied(50)
    # a snapshot only fits an engine of the same shape
    c = t.Engine(cap + 1, n, track_denied=True)
    with pytest.raises(t.TcError):
        c.snapshot_load(path)
    for e in (a, b, c):
        e.close()


def test_snapshot_string_mode(tmp_path):
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(6)
    keys = [b"acct:%d" % i for i in range(3000)] + [b"long/" + b"q" * 80 + b"/%d" % i for i in range(30)]

    def batch(lo, hi, n=20000):
        idx = rng.integers(lo, hi, n)
        idx[:40] = len(keys) - 1 - rng.integers(0, 30, 40)
        return O.pack_keys([keys[i] for i in idx])

    a = t.Engine(8192, 20000, key_mode=True)
    batches = [batch(0, 1000), batch(0, 2000), batch(500, 3000), batch(0, 3000), batch(0, 3000)]
    for i, (kb, ko) in enumerate(batches[:3]):
        a.rate_limit_batch_keys(kb, ko, max_burst=4, count_per_period=10, period=60, quantity=1, now_ns=T0 + i * 10**9)
    a.sweep_expired(T0 + 2 * 10**9)   # leaves tombstones and a partly used free stack in the snapshot
    path = str(tmp_path / "keys.snap")
    a.snapshot_save(path)
    b = t.Engine(8192, 20000, key_mode=True)
    b.snapshot_load(path)
    assert a.debug_check_keys() == 0 and b.debug_check_keys() == 0  # (the loaded table is consistent in itself, position column included)
    for i, (kb, ko) in enumerate(batches[3:]):
        now = T0 + (3 + i) * 10**9
        _same(a.rate_limit_batch_keys(kb, ko, max_burst=4, count_per_period=10, period=60, quantity=1, now_ns=now),
              b.rate_limit_batch_keys(kb, ko, max_burst=4, count_per_period=10, period=60, quantity=1, now_ns=now))
    t_end = T0 + 6 * 10**9
    for k in keys[::37] + keys[-30:]:
        assert a.get(k, t_end) == b.get(k, t_end) and a.lookup_slot(k) == b.lookup_slot(k), k
    assert a.counters() == b.counters()
    assert a.sweep_expired(T0 + 10**12) == b.sweep_expired(T0 + 10**12)
    assert a.debug_check_keys() == 0 and b.debug_check_keys() == 0
    a.close()
    b.close()


def test_corrupt_or_truncated_snapshot_is_refused_before_the_engine_is_touched(tmp_path):
    """tc_snapshot_load copied whatever the file held.  The payload now carries its size and a
    checksum, verified in a first pass over the file: a flipped byte or a short file is refused and the engine
    keeps answering from the state it had."""
    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W
    cap, n = 3000, 8000
    rng = np.random.default_rng(6)
    a = t.Engine(cap, n, key_mode=True)
    kb, ko = W.string_keys(rng.integers(0, 2500, n))
    a.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0)
    good = str(tmp_path / "good.snap")
    a.snapshot_save(good)
    raw = bytearray(open(good, "rb").read())
    bad_byte, short = str(tmp_path / "flipped.snap"), str(tmp_path / "short.snap")
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x40
    open(bad_byte, "wb").write(flipped)
    open(short, "wb").write(raw[: len(raw) - 4096])
    b = t.Engine(cap, n, key_mode=True)
    kb2, ko2 = W.string_keys(rng.integers(5000, 6000, n))
    b.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0)
    before = (b.counters(), b.read_state(0, cap))
    for path in (bad_byte, short):
        with pytest.raises(t.TcError):
            b.snapshot_load(path)
        after = (b.counters(), b.read_state(0, cap))
        assert before[0] == after[0] and np.array_equal(before[1][0], after[1][0]) and np.array_equal(before[1][1], after[1][1])
    # ... and it still works, and takes the good file
    r1 = b.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 1)
    assert r1.status.max() == 0
    b.snapshot_load(good)
    _same(a.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 2),
          b.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 2))
    a.close()
    b.close()
