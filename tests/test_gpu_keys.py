"""GPU parity (string-key mode): on-device key table + GCRA engine vs the
string-keyed AdaptiveStore oracle -- bit-exact outputs on the same streams."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu

KAT = kat.load()
T0 = KAT["t0_ns"]
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")


def _engine(capacity, max_batch=1 << 16, **kw):
    import throttlecrab_amd as t
    e = t.Engine(capacity, max_batch, key_mode=True, **kw)
    e.check_on_close = True  # close() asserts tc_selfcheck() == 0
    return e


def _oracle(capacity=1000):
    """AdaptiveStore port whose cleanup heuristics never fire on their own.  With
    timestamps that go BACKWARDS the reference's result depends on when its store
    happened to clean (an entry that expired at t1 is live again for a request
    stamped t0 < t1 unless a cleanup ran in between -- see
    tests/test_oracle_golden.py::test_cleanup_is_not_neutral_when_time_goes_back);
    the engine matches the reference with cleanup deferred to tc_sweep_expired."""
    from oracle import oracle as O
    return O.AdaptiveOracle(capacity=max(capacity, 100000), created_ns=T0, auto_cleanup=False)


def assert_same(res, ref, ctx=""):
    for f in FIELDS:
        got = getattr(res, f)
        if not isinstance(got, np.ndarray):
            got = got.cpu().numpy()
        exp = getattr(ref, f)
        bad = np.nonzero(got.astype(np.int64) != exp.astype(np.int64))[0]
        assert bad.size == 0, f"{ctx}: field {f} differs at {bad[:8]} got {got[bad[:8]]} want {exp[bad[:8]]}"


@pytest.mark.parametrize("sc", KAT["scenarios"], ids=[s["name"] for s in KAT["scenarios"]])
def test_reference_known_answers_string_keys(sc):
    eng = _engine(256, 64)
    kat.replay_scenario(sc, eng)
    eng.close()


@pytest.mark.parametrize("case", KAT["store_contract"], ids=[c["name"] for c in KAT["store_contract"]])
def test_store_contract_string_keys(case):
    eng = _engine(1024, 64)
    kat.replay_store_contract(case, eng, T0)
    eng.close()


def _keyset(rng, n_keys):
    from oracle import oracle as O
    keys = []
    for i in range(n_keys):
        kind = i % 5
        if kind == 0:
            keys.append(b"key_%d" % i)
        elif kind == 1:
            keys.append(b"user:%d:api/v1/endpoint/%d" % (i, i * 7919))            # ~35 B (> one cell)
        elif kind == 2:
            keys.append(bytes(rng.integers(33, 127, int(rng.integers(1, 80))).astype(np.uint8)) + b"#%d" % i)
        elif kind == 3:
            keys.append(("kéy-\U0001F980-%d" % i).encode())
        else:
            keys.append(b"x" * 31 + b"%d" % (i % 10) if i < 50 else b"p%dq" % i)     # shared 31-byte prefixes
    for i in range(7, n_keys, 25):  # beyond the 112 bytes a slot's record holds inline: the overflow arena
        keys[i] = b"L%d|" % i + bytes([33 + (i + j) % 90 for j in range(113 + i % 200)])
    keys[0] = b""  # the empty key is a key (store_test_suite.rs:289-300)
    return keys


def _stream(rng, keys, n):
    from oracle import oracle as O
    idx = np.minimum(rng.zipf(1.4, n) - 1, len(keys) - 1) if rng.random() < 0.5 else rng.integers(0, len(keys), n)
    kb, ko = O.pack_keys([keys[i] for i in idx])
    return idx, kb, ko


PSETS = np.array([(5, 10, 60), (100, 1000, 3600), (1, 1, 1), (3, 7, 60), (10, 100, 60), (0, 1, 1), (2**63 - 1,) * 3],
                 dtype=np.int64)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_differential_host_pointers(seed):
    rng = np.random.default_rng(seed)
    keys = _keyset(rng, 700)
    eng, orc = _engine(2048), _oracle()
    for rnd in range(4):
        n = 20000
        idx, kb, ko = _stream(rng, keys[: 200 + 150 * rnd], n)  # new keys keep appearing
        ps = PSETS[idx % len(PSETS)]
        q = rng.choice(np.array([0, 1, 1, 2, -1], dtype=np.int64), n)
        now = T0 + rnd * 3 * 10**9 + rng.integers(0, 2 * 10**9, n)
        ref = orc.batch_keys(kb, ko, ps[:, 0].copy(), ps[:, 1].copy(), ps[:, 2].copy(), q, now)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=ps[:, 0].copy(), count_per_period=ps[:, 1].copy(),
                                        period=ps[:, 2].copy(), quantity=q, now_ns=now)
        assert_same(res, ref, f"seed {seed} round {rnd}")
    # every key the oracle holds live is bound on the device with the same value
    t_end = T0 + 20 * 10**9
    for k in keys[:200]:
        assert eng.get(k, t_end) == orc.get(k, t_end), k
    eng.close()


def test_uniform_batches_device_pointers():
    import torch
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    keys = [b"key_%d" % i for i in range(3000)]
    eng, orc = _engine(4096), _oracle(4096)
    eng.use_torch_stream()
    for rnd in range(4):
        n = 50000
        idx = np.minimum(rng.zipf(1.2, n) - 1, len(keys) - 1)
        kb, ko = O.pack_keys([keys[i] for i in idx])
        now = T0 + rnd * 500_000_000
        ref = orc.batch_keys(kb, ko, 10, 100, 60, 1, now)
        res = eng.rate_limit_batch_keys(torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda(),
                                        max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=now)
        torch.cuda.synchronize()
        assert_same(res, ref, f"round {rnd}")
    c = eng.counters()
    assert c["keys_inserted"] == len(set(idx.tolist()) | set()) or c["keys_inserted"] <= len(keys)
    eng.close()


def test_sweep_unbinds_keys_and_matches_cleanup():
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    keys = [b"k%d" % i for i in range(5000)]
    # an oracle store that never cleans on its own (no time / op-count trigger), so that
    # "removed" of the forced cleanup is comparable with the device sweep
    eng, orc = _engine(6000), O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    kb, ko = O.pack_keys(keys)
    now = T0 + rng.integers(0, 20 * 10**9, len(keys))
    ref = orc.batch_keys(kb, ko, 3, 30, 60, 1, now)      # ttl <= 8 s
    res = eng.rate_limit_batch_keys(kb, ko, max_burst=3, count_per_period=30, period=60, quantity=1, now_ns=now)
    assert_same(res, ref)
    for rnd, t_sweep in enumerate((T0 + 10 * 10**9, T0 + 15 * 10**9, T0 + 40 * 10**9)):
        before, forced = len(orc), orc.cleanups
        orc.force_cleanup(t_sweep)
        assert orc.cleanups == forced + 1 == rnd + 1, "oracle cleaned on its own"
        assert eng.sweep_expired(t_sweep) == before - len(orc)
        assert eng.counters()["live_slots"] == len(orc)
        # swept keys are unbound, live keys still resolve
        for k in keys[::97]:
            assert (eng.lookup_slot(k) >= 0) == (orc.get(k, t_sweep) is not None), k
        # and they can come back: same decisions as a fresh key in the reference
        now2 = t_sweep + 1000
        ref = orc.batch_keys(kb, ko, 3, 30, 60, 1, now2)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=3, count_per_period=30, period=60, quantity=1, now_ns=now2)
        assert_same(res, ref, f"after sweep {rnd}")
    eng.close()


def test_many_sweeps_trigger_table_rebuild():
    from oracle import oracle as O
    eng, orc = _engine(512), _oracle()
    for gen in range(12):  # 12 generations x 400 fresh keys through a 512-slot store (1024-entry table)
        keys = [b"gen%d-%d" % (gen, i) for i in range(400)]
        kb, ko = O.pack_keys(keys)
        now = T0 + gen * 100 * 10**9
        ref = orc.batch_keys(kb, ko, 2, 60, 60, 1, now)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=2, count_per_period=60, period=60, quantity=1, now_ns=now)
        assert_same(res, ref, f"gen {gen}")
        eng.sweep_expired(now + 50 * 10**9)
        assert eng.counters()["live_slots"] == 0
    eng.close()


def test_table_full_reports_and_keeps_working():
    import throttlecrab_amd as t
    from oracle import oracle as O
    eng = _engine(100, 1024)
    keys = [b"full-%d" % i for i in range(150)]
    kb, ko = O.pack_keys(keys)
    with pytest.raises(t.TcError) as ei:
        eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0)
    assert ei.value.code == -5
    # the first 100 distinct keys were served; after a sweep far in the future the rest fit
    assert eng.counters()["keys_inserted"] == 100
    eng.sweep_expired(T0 + 10**12)
    kb2, ko2 = O.pack_keys(keys[100:])
    res = eng.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 10**12 + 1)
    assert res.status.max() == 0 and res.allowed.min() == 1
    eng.close()


def test_config5_shape_mixed_insert_lookup_1m_keys():
    """BASELINE configs[4] shape at 1 M keys: hit / new / re-hit-after-expiry mix + sweeps."""
    import torch
    from oracle import oracle as O
    n_keys, B = 1_000_000, 200_000
    eng, orc = _engine(n_keys + 1000, B), O.AdaptiveOracle(capacity=4 * n_keys, created_ns=T0,
                                                           auto_cleanup=False)
    eng.use_torch_stream()
    rng = np.random.default_rng(12)
    seen = 0
    for step in range(6):
        new = rng.integers(seen, min(n_keys, seen + B // 4), B // 5) if seen < n_keys else np.zeros(0, np.int64)
        old = rng.integers(0, max(seen, 1), B - len(new))
        ids = np.concatenate([old, new])
        rng.shuffle(ids)
        seen = min(n_keys, seen + B // 4)
        kb, ko = O.format_keys(ids.astype(np.uint32))
        now = T0 + step * 4 * 10**9  # ttl of (10,100,60) <= 11 s: some keys expire between steps
        ref = orc.batch_keys(kb, ko, 10, 100, 60, 1, now)
        res = eng.rate_limit_batch_keys(torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda(),
                                        max_burst=10, count_per_period=100, period=60, quantity=1, now_ns=now)
        torch.cuda.synchronize()
        assert_same(res, ref, f"step {step}")
        if step % 2 == 1:
            orc.force_cleanup(now)  # (the oracle may also have cleaned on its own: compare what is left)
            eng.sweep_expired(now)
            assert eng.counters()["live_slots"] == len(orc)
    eng.close()


def test_pipelined_key_batches_inputs_ready():
    """TC_B_INPUTS_READY key batches: the key stage runs on the key stream, grouping on the
    auxiliary streams, evaluation in order; interleaved with host-pointer batches, single-key
    store operations and a sweep (which all run their key stage on the main stream).  New keys
    keep appearing and hot keys recur in every batch."""
    import torch
    from oracle import oracle as O
    rng = np.random.default_rng(91)
    keys = [b"pk_%d" % i for i in range(40000)] + [b"long-key-" + b"y" * 60 + b"%d" % i for i in range(500)]
    eng, orc = _engine(60000, 60000), _oracle(60000)
    eng.use_torch_stream()
    n, nb = 50000, 10
    staged = []
    for bidx in range(nb):
        hi = 4000 + 4000 * bidx
        idx = np.where(rng.random(n) < 0.3, np.minimum(rng.zipf(1.3, n) - 1, hi - 1), rng.integers(0, hi, n))
        idx[:50] = len(keys) - 1 - rng.integers(0, 500, 50)   # a few keys beyond the inline 112 bytes
        kb, ko = O.pack_keys([keys[i] for i in idx])
        staged.append((kb, ko, torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda()))
    torch.cuda.synchronize()
    outs, refs = [], []
    for bidx, (kb, ko, dkb, dko) in enumerate(staged):
        now = T0 + bidx * 2 * 10**9
        refs.append(orc.batch_keys(kb, ko, 5, 10, 60, 1, now))
        if bidx % 4 == 2:   # host-pointer batch in the middle of the pipeline
            outs.append(eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1,
                                                  now_ns=now))
        else:
            outs.append(eng.rate_limit_batch_keys(dkb, dko, max_burst=5, count_per_period=10, period=60, quantity=1,
                                                  now_ns=now, inputs_ready=True))
        if bidx == 5:       # single-key operations and a sweep between pipelined batches
            assert eng.get(b"pk_0", now) == orc.get(b"pk_0", now)
            orc.force_cleanup(now)
            eng.sweep_expired(now)
            assert eng.counters()["live_slots"] == len(orc)
    eng.synchronize()
    torch.cuda.synchronize()
    for bidx in range(nb):
        assert_same(outs[bidx], refs[bidx], f"piped key batch {bidx}")
    t_end = T0 + nb * 2 * 10**9
    for k in keys[:300] + keys[-50:]:
        assert eng.get(k, t_end) == orc.get(k, t_end), k
    eng.close()


@pytest.mark.parametrize("params", [(100, 1000, 3600), (5, 10, 60)], ids=["bench_params", "forces_denials"])
def test_baseline_config0_1k_keys_100k_requests(params):
    """BASELINE configs[0] / SURVEY section 8(d) cfg 1: 1 000 string keys key_<i>, 100 000 requests with
    a uniform key index (seed 1), q = 1, request i stamped t0 + i * 10 us -- through the AdaptiveStore
    port on the CPU and through the engine (per-request timestamps: the general path), bit-exact on
    all five result fields + status; also in one pass, in 7 uneven batches and one request at a time
    for the first 300."""
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    n_keys, n = 1000, 100_000
    idx = rng.integers(0, n_keys, n)
    keys = [b"key_%d" % i for i in range(n_keys)]
    now = T0 + np.arange(n, dtype=np.int64) * 10_000
    kb, ko = O.pack_keys([keys[i] for i in idx])
    orc = O.AdaptiveOracle(capacity=n_keys, created_ns=T0)        # the reference's own cleanup heuristics on
    ref = orc.batch_keys(kb, ko, *params, 1, now)
    assert 0 < int(ref.allowed.sum()) <= n
    # (a) one batch
    eng = _engine(2048, n)
    res = eng.rate_limit_batch_keys(kb, ko, max_burst=params[0], count_per_period=params[1], period=params[2], quantity=1, now_ns=now)
    assert_same(res, ref, "one batch")
    eng.close()
    # (b) uneven batches, (c) the first requests one at a time through tc_rate_limit
    eng = _engine(2048, n)
    cuts = [0, 300, 301, 5000, 5064, 40000, 99999, n]
    for i in range(300):
        got = eng.rate_limit(keys[idx[i]], *params, 1, int(now[i]))
        assert got == (int(ref.status[i]), bool(ref.allowed[i]), int(ref.limit[i]), int(ref.remaining[i]),
                       int(ref.reset_after_ns[i]), int(ref.retry_after_ns[i])), i
    for a, b in zip(cuts[1:-1], cuts[2:]):
        part_kb, part_ko = O.pack_keys([keys[i] for i in idx[a:b]])
        res = eng.rate_limit_batch_keys(part_kb, part_ko, max_burst=params[0], count_per_period=params[1], period=params[2],
                                        quantity=1, now_ns=now[a:b])
        for f in FIELDS:
            got = getattr(res, f)
            assert np.array_equal(got.astype(np.int64), getattr(ref, f)[a:b].astype(np.int64)), (f, a, b)
    assert eng.counters()["total"] == n and eng.counters()["allowed"] == int(ref.allowed.sum())
    eng.close()


@pytest.mark.parametrize("general", [False, True], ids=["one_now", "per_request_now"])
def test_async_host_key_batches(general):
    """TC_B_ASYNC key batches from a ring of 3 pinned buffer sets: key arena and offsets are staged on the key
    stream, resolved there, grouped and evaluated as usual, results copied back behind the evaluation.  New
    keys keep appearing, hot keys recur, a synchronous batch, a single-key operation and a sweep are mixed in."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    keys = [b"ak_%d" % i for i in range(30000)] + [b"long-key-" + b"z" * 60 + b"%d" % i for i in range(300)]
    eng, orc = _engine(50000, 40000), _oracle(50000)
    n, nb, K = 30000, 10, 3
    F = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")
    ring = [dict(kb=eng.host_alloc(n * 80, np.uint8), ko=eng.host_alloc(n + 1, np.uint32), now=eng.host_alloc(n, np.int64),
                 out=t.BatchResult(**{f: eng.host_alloc(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in F}))
            for _ in range(K)]
    refs, got = [], []
    for bidx in range(nb):
        r = ring[bidx % K]
        if bidx >= K:
            eng.wait_batches(K - 1)
            got.append({f: getattr(r["out"], f).copy() for f in F})
        hi = 3000 + 3000 * bidx
        idx = np.where(rng.random(n) < 0.3, np.minimum(rng.zipf(1.3, n) - 1, hi - 1), rng.integers(0, hi, n))
        idx[:40] = len(keys) - 1 - rng.integers(0, 300, 40)   # a few keys beyond the inline 112 bytes
        kb, ko = O.pack_keys([keys[i] for i in idx])
        r["kb"][:kb.size] = kb
        r["ko"][:] = ko
        base = T0 + bidx * 2 * 10**9
        sync_call = bidx == 4
        if general:
            r["now"][:] = base + rng.integers(0, 10**9, n)
            refs.append(orc.batch_keys(kb, ko, 5, 10, 60, 1, r["now"]))
            eng.rate_limit_batch_keys(r["kb"], r["ko"], max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=r["now"],
                                      want=F, out=r["out"], async_=not sync_call)
        else:
            refs.append(orc.batch_keys(kb, ko, 5, 10, 60, 1, base))
            eng.rate_limit_batch_keys(r["kb"], r["ko"], max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=base,
                                      want=F, out=r["out"], async_=not sync_call)
        if bidx == 6:       # single-key operation and a sweep between asynchronous batches
            now = base + 10**9
            assert eng.get(b"ak_0", now) == orc.get(b"ak_0", now)
            orc.force_cleanup(now)
            eng.sweep_expired(now)
            assert eng.counters()["live_slots"] == len(orc)
    eng.wait_batches(0)
    for bidx in range(nb - K, nb):
        got.append({f: getattr(ring[bidx % K]["out"], f).copy() for f in F})
    for bidx in range(nb):
        assert_same(t.BatchResult(**got[bidx]), refs[bidx], f"async key batch {bidx}")
    t_end = T0 + nb * 2 * 10**9
    for k in keys[:300] + keys[-50:]:
        assert eng.get(k, t_end) == orc.get(k, t_end), k
    eng.close()


def test_small_host_key_batches_single_launch():
    """String-key batches of at most 1024 requests in one launch: new keys (also twice in one batch), known
    keys, long keys, the empty key, per-request timestamps and rates; mixed with big batches, single calls and
    a sweep; finally a table that runs full."""
    from oracle import oracle as O
    rng = np.random.default_rng(29)
    keys = [b"sk_%d" % i for i in range(3000)] + [b"long-key-" + b"q" * 70 + b"%d" % i for i in range(40)] + [b""]
    eng, orc = _engine(6000, 4000), _oracle(6000)
    sizes = [1, 2, 5, 64, 65, 300, 1024, 1025, 2500, 17, 1000, 3, 700]
    for bidx, n in enumerate(sizes):
        hi = 50 + 230 * bidx
        idx = np.where(rng.random(n) < 0.4, rng.integers(0, 8, n), rng.integers(0, hi, n))
        idx[: max(1, n // 20)] = len(keys) - 1 - rng.integers(0, 41, max(1, n // 20))   # long keys and the empty key
        kb, ko = O.pack_keys([keys[i] for i in idx])
        now = T0 + bidx * 3 * 10**9 + rng.integers(0, 10**9, n)
        b, c, p_, q = rng.integers(0, 6, n), rng.integers(1, 30, n), rng.integers(1, 70, n), rng.integers(-1, 3, n)
        ref = orc.batch_keys(kb, ko, b, c, p_, q, now)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p_, quantity=q, now_ns=now)
        assert_same(res, ref, f"small key batch {bidx} (n={n})")
        if bidx == 6:
            t_sweep = T0 + bidx * 3 * 10**9 + 2 * 10**9
            orc.force_cleanup(t_sweep)
            eng.sweep_expired(t_sweep)
            assert eng.counters()["live_slots"] == len(orc)
            assert eng.rate_limit(b"sk_1", 5, 10, 60, 1, t_sweep)[:2] == orc.rate_limit(b"sk_1", 5, 10, 60, 1, t_sweep)[:2]
    t_end = T0 + len(sizes) * 3 * 10**9
    for k in keys[:200] + keys[-41:]:
        assert eng.get(k, t_end) == orc.get(k, t_end), k
    eng.close()
    # a table that runs full inside a small batch: the keys that fit are served, the others get Internal
    import throttlecrab_amd as t
    eng = _engine(16, 64)
    kb, ko = O.pack_keys([b"full_%d" % i for i in range(40)])
    with pytest.raises(t.TcError):
        eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0)
    kb, ko = O.pack_keys([b"full_%d" % i for i in range(8)])
    res = eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 1)
    assert (res.status == 0).all()      # bound by the first call (in request order: the first 16 keys got the slots)
    eng.check_on_close = False
    eng.close()


def test_async_key_batch_table_full_is_reported_later():
    """A TC_B_ASYNC batch cannot return TC_E_TABLE_FULL itself: the requests that found no slot carry status
    Internal, and the next synchronous key call delivers the return code once."""
    import throttlecrab_amd as t
    from oracle import oracle as O
    eng = _engine(16, 64)
    kb, ko = O.pack_keys([b"af_%d" % i for i in range(40)])
    pk, po = eng.host_alloc(kb.size, np.uint8), eng.host_alloc(ko.size, np.uint32)
    pk[:], po[:] = kb, ko
    out = t.BatchResult(status=eng.host_alloc(40, np.uint8), allowed=eng.host_alloc(40, np.uint8))
    eng.rate_limit_batch_keys(pk, po, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0, want=("status", "allowed"),
                              out=out, async_=True)
    eng.wait_batches(0)
    assert (out.status == 0).sum() == 16 and (out.status == 3).sum() == 24 and out.allowed[out.status == 0].all()
    served = [b"af_%d" % i for i in np.nonzero(out.status == 0)[0][:5]]
    kb2, ko2 = O.pack_keys(served)
    with pytest.raises(t.TcError):     # the earlier batch's TC_E_TABLE_FULL; this call's own requests were applied
        eng.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 1)
    res = eng.rate_limit_batch_keys(kb2, ko2, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 2)
    assert (res.status == 0).all() and (res.remaining == 2).all()      # third request on each of these keys
    eng.check_on_close = False
    eng.close()
