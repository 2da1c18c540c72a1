"""BASELINE configs[4] AT ITS STATED CONFIGURATION (SURVEY.md section 8(d) cfg 5): 10 M string keys through the
on-device key table -- first the keys `key_%d`, then 32..64-byte random ASCII keys (the 64-byte key records and,
for keys over 48 bytes, the overflow arena) --, then mixed batches of 1 Mi requests: 70 % hits of live keys,
20 % new keys, 10 % re-hits of keys whose entries have expired, rate (10, 100 / 60 s), one second per batch,
an expiry sweep every 4 batches (the first one unbinds ~9 M keys: table rebuild + overflow compaction).
Oracle: the AdaptiveStore port with its own cleanup heuristics off and force_cleanup at the sweep points
(adaptive_cleanup.rs:173-279).  All six result fields of every request, live entries after every sweep,
single-key lookups at the end.  Also: tombstones are recycled by inserts; the overflow arena is reclaimed."""
import numpy as np
import pytest

from tests.test_gpu_keys import T0, _engine, assert_same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("key_set", ["key_%d", "ascii_32_64"])
def test_config4_at_spec_10m_keys(key_set):
    import torch
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    assert W.T0_NS == T0
    B, n_prefill, n_mixed = 1 << 20, 10, 8
    long = key_set != "key_%d"
    st = W.Config4Stream(B, n_prefill=n_prefill, long=long, sweep_every=4)
    cap = B * n_prefill + B  # every prefilled key + the new keys of the steps before the first sweep
    eng = _engine(cap, B, key_arena_bytes=(448 << 20) if long else 0)  # ~5.4 M keys over 48 bytes, 64 bytes each
    eng.use_torch_stream()
    orc = O.AdaptiveOracle(capacity=cap, created_ns=T0, auto_cleanup=False)
    b, c, p = W.Config4Stream.PARAMS
    dev = torch.device("cuda:0")

    def run(ids, now, piped, ctx):
        dkb, dko = st.keys(ids, device=dev)
        kb, ko = dkb.cpu().numpy(), dko.cpu().numpy().astype(np.uint32)
        ref = orc.batch_keys(kb, ko, b, c, p, 1, now)
        res = eng.rate_limit_batch_keys(dkb, dko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now, inputs_ready=piped)
        torch.cuda.synchronize()
        assert_same(res, ref, ctx)
        return ref

    for k in range(n_prefill):
        ids, now = st.prefill(k)
        ref = run(ids, now, k % 2 == 1, f"{key_set} prefill {k}")
        assert ref.allowed.all()
    assert eng.counters()["keys_inserted"] == B * n_prefill == len(orc)
    last_ids = None
    for s in range(n_mixed):
        ids, now = st.mixed(s)
        ref = run(ids, now, s % 2 == 0, f"{key_set} mixed step {s}")
        assert 0.7 * B < ref.allowed.sum() < B  # the hot keys run out of burst; new and expired keys start over
        if st.sweep_due(s):
            before = len(orc)
            orc.force_cleanup(now)
            removed = eng.sweep_expired(now)
            assert removed == before - len(orc), (s, removed, before, len(orc))
            assert eng.counters()["live_slots"] == len(orc)
            # entries, key records, position column and free stack agree (tc_debug_check_keys), after the rebuild (s == 3) and
            # after a sweep that wrote tombstones
            assert eng.debug_check_keys() == 0, s
            if s == 3:
                assert removed > 8 * B  # the prefilled keys
        last_ids = ids
    # single-key lookups: live, expired-and-swept, never seen
    t_end = st.t1 + n_mixed * 10**9
    sample = np.concatenate([last_ids[:40], np.arange(0, 40) * 9973, [st.next_id + 5]])
    kb, ko = st.keys(sample)
    for i in range(len(sample)):
        key = bytes(kb[ko[i]:ko[i + 1]])
        assert eng.get(key, t_end) == orc.get(key, t_end), key
    assert eng.debug_check_keys() == 0
    eng.close()


def test_tombstones_are_recycled_and_failed_binds_do_not_pile_up():
    """a full table used to add one tombstone per request for an unseen key, until every miss
    scanned the whole table.  Inserts now claim the first tombstone of their chain; probing an exhausted
    table raises TC_E_TABLE_FULL."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    cap = 4096
    eng = _engine(cap, 8192)
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    kb, ko = W.string_keys(np.arange(cap))
    ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, T0)
    assert_same(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0), ref, "fill")
    for rnd in range(40):  # 40 x 8192 requests for keys that cannot get a slot
        kb, ko = W.string_keys(np.arange(10**6 + rnd * 8192, 10**6 + (rnd + 1) * 8192))
        with pytest.raises(t.TcError) as ei:
            eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 1 + rnd)
        assert ei.value.code == -5
    # the resident keys still answer (and fast: their chains did not fill up with tombstones), in one batch ...
    kb, ko = W.string_keys(np.arange(cap))
    ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, T0 + 100)
    assert_same(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 100), ref, "after the flood")
    # ... and after a sweep the table takes new keys again
    eng.sweep_expired(T0 + 10**12)
    orc.force_cleanup(T0 + 10**12)
    kb, ko = W.string_keys(np.arange(2 * 10**6, 2 * 10**6 + cap))
    ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, T0 + 10**12 + 1)
    assert_same(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 10**12 + 1), ref, "refill")
    assert eng.debug_check_keys() == 0
    eng.close()


def test_overflow_arena_is_reclaimed_by_the_sweep():
    """keys too long for their slot's record (over 112 bytes) live in an arena whose space was never
    given back: churn of long keys exhausted it for good.  Ten generations of 113..240-byte keys, each swept before the
    next arrives, through an arena that holds two of them."""
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    n = 20_000
    eng = _engine(2 * n, n, key_arena_bytes=2 * n * 256)
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    now = T0
    for gen in range(10):
        ids = np.arange(gen * n, (gen + 1) * n)
        kb, ko = W.pack_keys([(b"%d/" % i) + bytes([33 + (i * 7 + j) % 90 for j in range(113 + i % 128)]) for i in ids.tolist()])
        for rep in range(2):
            ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, now)
            res = eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now)
            assert_same(res, ref, f"generation {gen}.{rep}")
            now += 10**9
        now += 100 * 10**9
        orc.force_cleanup(now)
        assert eng.sweep_expired(now) == n
        assert eng.counters()["live_slots"] == len(orc) == 0
        assert eng.debug_check_keys() == 0, gen
    eng.close()


@pytest.mark.parametrize("cap", [257, 5000, 70001, 600_000])
def test_sweep_cycles_at_odd_capacities(cap):
    """The key-mode sweep in three kernels (k_sweep_keys: a stretch of the list per block; k_sweep_decide: one block scans the
    blocks' counts; k_sweep_tombstones: pushes the stretches onto the free stack, writes the tombstones) at capacities that are
    not multiples of anything: one block, a few, a ragged last one, the full grid of 2 048.  Six generations of keys, half of
    each generation refreshed before the sweep so that every sweep unbinds some keys and keeps others; removed / live counts
    and every result against the AdaptiveStore port, the table checked from both sides after every sweep."""
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    n = cap // 3
    eng = _engine(cap, max(n, 256))
    orc = O.AdaptiveOracle(capacity=2 * cap, created_ns=T0, auto_cleanup=False)
    now = T0
    for gen in range(6):
        ids = np.arange(gen * n, (gen + 1) * n)
        kb, ko = W.string_keys(ids)
        ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, now)
        assert_same(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now), ref, f"gen {gen}")
        # every second key of the generation is asked for again 50 s later: its entry lives 50 s longer than its neighbours'
        kb2, ko2 = W.string_keys(ids[::2])
        ref = orc.batch_keys(kb2, ko2, 5, 10, 60, 1, now + 50 * 10**9)
        assert_same(eng.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now + 50 * 10**9), ref, f"gen {gen} again")
        now += 70 * 10**9  # the untouched half has expired, the refreshed half has not
        before = len(orc)
        orc.force_cleanup(now)
        removed = eng.sweep_expired(now)
        assert removed == before - len(orc) and removed > 0, (gen, removed, before, len(orc))
        assert eng.counters()["live_slots"] == len(orc)
        assert eng.debug_check_keys() == 0, gen
    # the survivors of every generation still answer like the reference's
    kb, ko = W.string_keys(np.arange(0, 6 * n, 7))
    ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, now + 1)
    assert_same(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now + 1), ref, "survivors")
    assert eng.debug_check_keys() == 0
    eng.close()
