"""GPU parity (slot mode): the HIP engine (through the C ABI) vs the CPU oracle
on the same seeded streams -- bit-exact on all six outputs and on the resident
(tat, expiry) state."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu

KAT = kat.load()
T0 = KAT["t0_ns"]
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")


def _engine(capacity, max_batch=1 << 16):
    import throttlecrab_amd as t
    e = t.Engine(capacity, max_batch)
    e.check_on_close = True  # close() asserts tc_selfcheck() == 0
    return e


def _oracle(capacity):
    from oracle import oracle as O
    return O.DenseOracle(capacity)


def assert_same(res, ref, ctx=""):
    for f in FIELDS:
        got = getattr(res, f)
        if got is None:
            continue
        if not isinstance(got, np.ndarray):
            got = got.cpu().numpy()
        exp = getattr(ref, f)
        if f in ("status", "allowed"):
            got = got.astype(np.uint8)
        bad = np.nonzero(got.astype(np.int64) != exp.astype(np.int64))[0]
        assert bad.size == 0, f"{ctx}: field {f} differs at {bad[:8]} got {got[bad[:8]]} want {exp[bad[:8]]}"


def assert_record_same(res, ref, ctx=""):
    """result4[i] = (limit, remaining, reset_after_ns, retry_after_ns) of request i."""
    r4 = res.result4
    if not isinstance(r4, np.ndarray):
        r4 = r4.cpu().numpy()
    r4 = r4.reshape(-1, 4)
    for col, f in enumerate(("limit", "remaining", "reset_after_ns", "retry_after_ns")):
        bad = np.nonzero(r4[:, col] != getattr(ref, f).astype(np.int64))[0]
        assert bad.size == 0, f"{ctx}: result4.{f} differs at {bad[:8]}"


def assert_state_same(eng, orc, slots):
    tat, exp = eng.read_state(0, eng.capacity)
    for s in np.unique(slots):
        if s >= eng.capacity:
            continue
        ot, oe, occ = orc.peek(int(s))
        if not occ:
            assert exp[s] == 0, f"slot {s} should be vacant"
        else:
            assert (int(tat[s]), int(exp[s])) == (ot, oe), f"slot {s}: state ({tat[s]},{exp[s]}) want ({ot},{oe})"


class _SlotKeyed:
    """Replays string-keyed scenarios on a slot-mode engine (key -> slot by first use)."""

    def __init__(self, eng):
        self.eng = eng
        self.slots = {}

    def rate_limit(self, key, burst, count, period, q, now):
        s = self.slots.setdefault(key, len(self.slots))
        return self.eng.rate_limit(int(s).to_bytes(4, "little"), burst, count, period, q, now)


@pytest.mark.parametrize("sc", KAT["scenarios"], ids=[s["name"] for s in KAT["scenarios"]])
def test_reference_known_answers(sc):
    eng = _engine(64, 64)
    kat.replay_scenario(sc, _SlotKeyed(eng))
    eng.close()


@pytest.mark.parametrize("case", [c for c in KAT["store_contract"] if c["name"] not in ("special_keys", "many_keys")],
                         ids=lambda c: c["name"])
def test_store_contract(case):
    eng = _engine(2048, 64)

    class S:
        def __init__(self):
            self.m = {}

        def _k(self, key):
            return int(self.m.setdefault(key, len(self.m))).to_bytes(4, "little")

        def get(self, key, now):
            return eng.get(self._k(key), now)

        def set_if_not_exists_with_ttl(self, key, val, ttl, now):
            return eng.set_if_not_exists_with_ttl(self._k(key), val, ttl, now)

        def compare_and_swap_with_ttl(self, key, old, new, ttl, now):
            return eng.compare_and_swap_with_ttl(self._k(key), old, new, ttl, now)

        def sweep_expired(self, now):  # AdaptiveStore::cleanup (cleanup_test.rs cases)
            return eng.sweep_expired(now)

        def live_count(self):
            eng.sweep_expired(0)  # (refreshes the live-slot counter; nothing expires at time 0)
            return eng.counters()["live_slots"]

    kat.replay_store_contract(case, S(), T0)
    eng.close()


PARAM_SETS = np.array([
    (5, 10, 60), (100, 1000, 3600), (1, 1, 1), (2, 120, 60), (3, 7, 60), (10, 100, 60),
    (20, 600, 60), (10, 2**62, 60),                 # ei == 0
    (2**63 - 1, 2**63 - 1, 2**63 - 1),              # redis boundary test
    (0, 10, 60), (10, 0, 60), (10, 10, 0), (-3, 5, 5),  # invalid
    (2**32, 1, 2**62),                               # Duration*u32 overflow -> Internal
    (2**32 + 1, 10, 60),                             # (burst-1) as u32 wraps to 0
    ((2**63 - 1) // 1000, 100, 60),
], dtype=np.int64)


def _random_stream(rng, n, n_slots, capacity):
    slots = rng.integers(0, n_slots, n).astype(np.uint32)
    bad = rng.random(n) < 0.01
    slots[bad] = capacity + rng.integers(0, 5, bad.sum()).astype(np.uint32)  # out-of-range slots
    ps = PARAM_SETS[rng.integers(0, len(PARAM_SETS), n)]
    # most slots keep one param set so that state evolves meaningfully
    sticky = PARAM_SETS[(slots % 7).astype(np.int64)]
    use_sticky = rng.random(n) < 0.8
    ps = np.where(use_sticky[:, None], sticky, ps)
    q = rng.choice(np.array([0, 1, 1, 1, 2, 5, -1, 2**62], dtype=np.int64), n)
    now = T0 + rng.integers(-5 * 10**9, 30 * 10**9, n).astype(np.int64)
    now[rng.random(n) < 0.005] = -7  # pre-1970
    return slots, ps[:, 0].copy(), ps[:, 1].copy(), ps[:, 2].copy(), q, now


@pytest.mark.parametrize("seed,n,n_slots", [(1, 5000, 300), (2, 20000, 64), (3, 3000, 5000), (4, 40000, 7)])
def test_general_random_differential_host_pointers(seed, n, n_slots):
    cap = 6000
    rng = np.random.default_rng(seed)
    eng, orc = _engine(cap), _oracle(cap)
    from oracle import oracle as O
    for rnd in range(3):
        slots, b, c, p, q, now = _random_stream(rng, n, n_slots, cap)
        # out-of-range slots: Internal, outputs zero, nothing applied (our boundary
        # rule -- the reference has no slots); the oracle only sees in-range requests
        keep = slots < cap
        part = orc.batch_slots(slots[keep], b[keep], c[keep], p[keep], q[keep], now[keep])
        ref = O.BatchOut(n)
        ref.status[:] = 3
        for f in FIELDS:
            getattr(ref, f)[keep] = getattr(part, f)
        res = eng.rate_limit_batch_slots(slots, max_burst=b, count_per_period=c, period=p, quantity=q, now_ns=now)
        assert_same(res, ref, f"seed {seed} round {rnd}")
        assert_state_same(eng, orc, slots)
    cnt = eng.counters()
    assert cnt["total"] == 3 * n and cnt["allowed"] + cnt["denied"] + cnt["errors"] == cnt["total"]
    eng.close()


def test_general_device_pointers_match_host_pointers():
    import torch
    cap, n = 2000, 30000
    rng = np.random.default_rng(11)
    slots, b, c, p, q, now = _random_stream(rng, n, 150, cap)
    slots = np.minimum(slots, cap - 1)
    orc = _oracle(cap)
    ref = orc.batch_slots(slots, b, c, p, q, now)
    eng = _engine(cap)
    eng.use_torch_stream()
    dev = "cuda:0"
    tt = lambda a: torch.from_numpy(a.astype(np.int64)).to(dev)
    res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).to(dev), max_burst=tt(b),
                                     count_per_period=tt(c), period=tt(p), quantity=tt(q), now_ns=tt(now),
                                     want=FIELDS + ("allowed_bits", "result4"))
    torch.cuda.synchronize()
    assert_same(res, ref, "device ptrs")
    assert_record_same(res, ref, "device ptrs")
    bits = res.allowed_bits.cpu().numpy().view(np.uint64)
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).reshape(-1)[:n]
    assert np.array_equal(unpacked.astype(np.uint8), ref.allowed)
    assert_state_same(eng, orc, slots)
    eng.close()


UNIFORM_CASES = [
    # (burst, count, period, q, label)
    (100, 1000, 3600, 1, "bench params"),
    (5, 10, 60, 1, "small burst"),
    (1, 1, 1, 1, "burst 1: entries expire at once (irregular walk)"),
    (1, 1, 1, 0, "q=0 burst 1: never-expiring poison"),
    (10, 100, 60, 0, "q=0"),
    (10, 2**62, 60, 1, "ei=0"),
    (10, 100, 60, 3, "q=3"),
    (10, 10, 60, 2**62, "huge q saturates"),
    (2**63 - 1, 2**63 - 1, 2**63 - 1, 1, "i64::MAX triple"),
    (2**32 + 1, 10, 60, 1, "u32 wrap of burst-1"),
]


@pytest.mark.parametrize("burst,count,period,q,label", UNIFORM_CASES, ids=[c[4] for c in UNIFORM_CASES])
@pytest.mark.parametrize("registered", [False, True])
def test_uniform_batches_with_heavy_duplicates(burst, count, period, q, label, registered):
    """Scalar now/quantity + per-slot params: the closed-form path."""
    cap, n = 5000, 60000
    import zlib
    rng = np.random.default_rng(zlib.crc32(label.encode()))
    eng, orc = _engine(cap), _oracle(cap)
    if registered:
        eng.register_params_uniform(burst, count, period)
    z = rng.zipf(1.3, n).astype(np.int64)
    for rnd in range(4):
        slots = ((z * 2654435761 + rnd * 17) % 700).astype(np.uint32) if rnd % 2 == 0 else \
            rng.integers(0, cap, n).astype(np.uint32)
        now = T0 + rnd * 700_000_000  # 0.7 s apart; also goes back once
        if rnd == 3:
            now = T0 - 10**9
        ref = orc.batch_slots(slots, burst, count, period, q, now)
        kw = dict(registered=True) if registered else dict(max_burst=burst, count_per_period=count, period=period)
        res = eng.rate_limit_batch_slots(slots, quantity=q, now_ns=now, **kw)
        assert_same(res, ref, f"{label} round {rnd}")
        assert_state_same(eng, orc, slots)
    eng.close()


def test_registered_params_per_slot_and_unregistered():
    cap, n = 1000, 20000
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 8, cap)
    valid = PARAM_SETS[:8]
    eng, orc = _engine(cap), _oracle(cap)
    # slots 900.. stay unregistered -> InvalidRateLimit
    eng.register_params(valid[idx[:900], 0], valid[idx[:900], 1], valid[idx[:900], 2])
    slots = rng.integers(0, cap, n).astype(np.uint32)
    b = np.where(slots < 900, valid[idx[slots], 0], 0)
    c = np.where(slots < 900, valid[idx[slots], 1], 0)
    p = np.where(slots < 900, valid[idx[slots], 2], 0)
    for rnd in range(3):
        now = T0 + rnd * 10**9
        ref = orc.batch_slots(slots, b, c, p, 1, now)
        res = eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=now)
        assert_same(res, ref, f"registered round {rnd}")
    # per-request now with registered params -> general path
    now = T0 + 5 * 10**9 + rng.integers(0, 10**9, n)
    ref = orc.batch_slots(slots, b, c, p, 1, now)
    res = eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=now)
    assert_same(res, ref, "registered + per-request now")
    assert_state_same(eng, orc, slots)
    eng.close()


def test_register_rejects_invalid_triples():
    import throttlecrab_amd as t
    eng = _engine(16, 16)
    with pytest.raises(t.TcError):
        eng.register_params_uniform(0, 10, 60)
    with pytest.raises(t.TcError):
        eng.register_params(np.array([5, 5]), np.array([10, -1]), np.array([60, 60]))
    eng.close()


def test_unique_slots_fast_path():
    cap, n = 50000, 20000
    rng = np.random.default_rng(9)
    eng, orc = _engine(cap), _oracle(cap)
    for rnd in range(3):
        slots = rng.permutation(cap)[:n].astype(np.uint32)
        now = T0 + rng.integers(0, 10**9, n)
        q = rng.integers(0, 4, n)
        ref = orc.batch_slots(slots, 3, 30, 60, q, now)
        res = eng.rate_limit_batch_slots(slots, max_burst=3, count_per_period=30, period=60, quantity=q, now_ns=now,
                                         unique=True)
        assert_same(res, ref, f"unique round {rnd}")
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


def test_sweep_matches_adaptive_cleanup():
    cap, n = 4000, 8000
    rng = np.random.default_rng(21)
    eng, orc = _engine(cap), _oracle(cap)
    slots = rng.integers(0, cap, n).astype(np.uint32)
    now = T0 + rng.integers(0, 20 * 10**9, n)
    b, c, p = 3, 30, 60  # ttl <= ~8 s
    ref = orc.batch_slots(slots, b, c, p, 1, now)
    res = eng.rate_limit_batch_slots(slots, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now)
    assert_same(res, ref)
    for t_sweep in (T0 + 10 * 10**9, T0 + 15 * 10**9, T0 + 100 * 10**9):
        assert eng.sweep_expired(t_sweep) == orc.sweep(t_sweep)
        assert eng.counters()["live_slots"] == orc.live()
        assert_state_same(eng, orc, np.arange(cap))
    # decisions after a sweep are unchanged (cleanup is decision-neutral)
    now2 = T0 + 101 * 10**9
    ref = orc.batch_slots(slots, b, c, p, 1, now2)
    res = eng.rate_limit_batch_slots(slots, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now2)
    assert_same(res, ref, "after sweep")
    eng.close()


def test_empty_and_single_and_errors():
    import throttlecrab_amd as t
    eng = _engine(8, 8)
    res = eng.rate_limit_batch_slots(np.zeros(0, np.uint32), max_burst=1, count_per_period=1, period=1, now_ns=T0)
    assert res.allowed.size == 0
    with pytest.raises(t.TcError) as ei:
        eng.rate_limit_batch_slots(np.zeros(9, np.uint32), max_burst=1, count_per_period=1, period=1, now_ns=T0)
    assert ei.value.code == -4
    st = eng.rate_limit((3).to_bytes(4, "little"), 5, 10, 60, 1, T0)
    assert st == (0, True, 5, 4, 24_000_000_000, 0)
    eng.close()


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_full_size_10m_keys_1m_batch(kind):
    """BASELINE configs 2 and 3 at full size: 10 M slots, 1 M-request batches."""
    import torch
    from throttlecrab_amd import workload as W
    cap, B = 10_000_000, 1 << 20
    eng, orc = _engine(cap, B), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    z = W.Zipf(cap) if kind == "zipf" else None
    tot_allowed = 0
    for bidx in range(4):
        slots = z.slots(B, start=bidx * B) if z else W.uniform_slots(cap, B, start=bidx * B)
        now = T0 + bidx * 1_000_000
        ref = orc.batch_slots(slots, *W.REF_PARAMS, 1, now)
        d_slots = torch.from_numpy(slots.astype(np.int32)).cuda()
        res = eng.rate_limit_batch_slots(d_slots, registered=True, quantity=1, now_ns=now)
        torch.cuda.synchronize()
        assert_same(res, ref, f"{kind} batch {bidx}")
        tot_allowed += int(ref.allowed.sum())
    cnt = eng.counters()
    assert cnt["allowed"] == tot_allowed and cnt["total"] == 4 * B
    tat, exp = eng.read_state(0, cap)
    sample = np.unique(slots)[:: max(1, len(np.unique(slots)) // 5000)]
    for s in sample:
        ot, oe, occ = orc.peek(int(s))
        assert occ and (int(tat[s]), int(exp[s])) == (ot, oe)
    eng.close()


@pytest.mark.parametrize("own_stream", [False, True], ids=["torch_stream", "engine_stream"])
@pytest.mark.parametrize("mode", ["uniform", "general", "mixed"])
def test_pipelined_batches_inputs_ready(own_stream, mode):
    """TC_B_INPUTS_READY: batches are grouped on the auxiliary streams while earlier ones are
    still being evaluated; results and state must equal the sequential oracle.  12 batches are
    issued back to back without a host sync (ring of 3 scratch sets is reused 4 times); hot
    slots recur in every batch so a mis-ordered evaluation or a clobbered scratch set shows."""
    import torch
    cap, n, nb = 3000, 50000, 12
    rng = np.random.default_rng(77)
    eng, orc = _engine(cap, n), _oracle(cap)
    if not own_stream:
        eng.use_torch_stream()
    d_slots, refs, outs = [], [], []
    for bidx in range(nb):
        slots = ((rng.zipf(1.2, n) * 2654435761) % cap).astype(np.uint32)
        d_slots.append((slots, torch.from_numpy(slots.astype(np.int32)).cuda()))
    torch.cuda.synchronize()
    for bidx, (slots, ds) in enumerate(d_slots):
        piped = mode != "mixed" or bidx % 3 != 1   # mixed: every third batch runs in order on the stream
        if mode == "general" or (mode == "mixed" and bidx % 2):
            now = T0 + bidx * 10**8 + rng.integers(0, 10**8, n)
            q = rng.integers(0, 3, n)
            refs.append(orc.batch_slots(slots, 5, 10, 60, q, now))
            tt = lambda a: torch.from_numpy(a.astype(np.int64)).cuda()
            if not own_stream:
                kq, kn = tt(q), tt(now)
            else:
                kq, kn = tt(q), tt(now)
                torch.cuda.synchronize()  # engine-owned stream: the columns must be complete at call time
            outs.append(eng.rate_limit_batch_slots(ds, max_burst=5, count_per_period=10, period=60, quantity=kq,
                                                   now_ns=kn, inputs_ready=piped))
        else:
            now = T0 + bidx * 10**8
            refs.append(orc.batch_slots(slots, 5, 10, 60, 1, now))
            outs.append(eng.rate_limit_batch_slots(ds, max_burst=5, count_per_period=10, period=60, quantity=1,
                                                   now_ns=now, inputs_ready=piped))
    eng.synchronize()
    torch.cuda.synchronize()
    for bidx in range(nb):
        assert_same(outs[bidx], refs[bidx], f"{mode} piped batch {bidx}")
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


@pytest.mark.parametrize("uniform", [True, False])
@pytest.mark.parametrize("dev", [False, True], ids=["host", "device"])
def test_decision_records(uniform, dev):
    """tc_decision: remaining / reset_after / retry_after / allowed / status in one 32-byte record."""
    import torch
    import throttlecrab_amd as t
    cap, n = 800, 30000
    rng = np.random.default_rng(32)
    eng, orc = _engine(cap), _oracle(cap)
    eng.use_torch_stream()
    for rnd in range(3):
        slots = rng.integers(0, cap + 3, n).astype(np.uint32)          # a few out-of-range slots: status Internal
        if uniform:
            q, now = 1, T0 + rnd * 10**9
        else:
            q, now = rng.choice(np.array([0, 1, 2, -1], dtype=np.int64), n), T0 + rnd * 10**9 + rng.integers(0, 10**9, n)
        keep = slots < cap
        from oracle import oracle as O
        part = orc.batch_slots(slots[keep], 5, 10, 60, q if uniform else q[keep], now if uniform else now[keep])
        ref = O.BatchOut(n)
        ref.status[:] = 3
        for f in FIELDS:
            getattr(ref, f)[keep] = getattr(part, f)
        if dev:
            tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
            kw = dict(quantity=q if uniform else tt(q), now_ns=now if uniform else tt(now))
            res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), max_burst=5, count_per_period=10,
                                             period=60, want=t.Engine.DECISION_FIELDS, **kw)
            torch.cuda.synchronize()
        else:
            res = eng.rate_limit_batch_slots(slots, max_burst=5, count_per_period=10, period=60, quantity=q, now_ns=now,
                                             want=t.Engine.DECISION_FIELDS)
        got = t.Engine.unpack_decisions(res.decisions)
        for f in ("allowed", "status", "remaining", "reset_after_ns", "retry_after_ns"):
            bad = np.nonzero(got[f].astype(np.int64) != getattr(ref, f).astype(np.int64))[0]
            assert bad.size == 0, f"round {rnd}: decisions.{f} differs at {bad[:8]}"
    eng.close()


@pytest.mark.parametrize("uniform", [True, False])
def test_result_records_host_pointers(uniform):
    """RateLimitResult as one 32-byte record per request == the four columns."""
    import throttlecrab_amd as t
    cap, n = 800, 30000
    rng = np.random.default_rng(31)
    eng, orc = _engine(cap), _oracle(cap)
    for rnd in range(3):
        slots = rng.integers(0, cap, n).astype(np.uint32)
        if uniform:
            q, now = 1, T0 + rnd * 10**9
        else:
            q, now = rng.integers(0, 3, n), T0 + rnd * 10**9 + rng.integers(0, 10**9, n)
        ref = orc.batch_slots(slots, 5, 10, 60, q, now)
        res = eng.rate_limit_batch_slots(slots, max_burst=5, count_per_period=10, period=60, quantity=q, now_ns=now,
                                         want=t.Engine.RECORD_FIELDS)
        assert_same(res, ref, f"records round {rnd}")
        assert_record_same(res, ref, f"records round {rnd}")
    eng.close()


@pytest.mark.parametrize("burst,count,period,label", [
    (5, 10, 60, "mostly denied"),
    (100000, 1000, 1, "long allowed runs (burst 100000)"),
    (1, 1, 1, "burst 1: every entry expires as it is written"),
    (50, 2**62, 60, "ei = 0: everything allowed"),
])
def test_general_batches_hot_keys_span_many_waves(burst, count, period, label):
    """Per-request timestamps / quantities with a few very hot keys: segments of tens of
    thousands of requests cross hundreds of waves and blocks (k_eval_general's hand-over chain)."""
    cap, n = 4096, 200_000
    import zlib
    rng = np.random.default_rng(zlib.crc32(label.encode()))
    eng, orc = _engine(cap, n), _oracle(cap)
    for rnd in range(3):
        hot = rng.random(n)
        slots = np.where(hot < 0.45, 7, np.where(hot < 0.7, 4095, np.where(hot < 0.8, 63, rng.integers(0, cap, n)))).astype(np.uint32)
        now = T0 + rnd * 5 * 10**9 + np.sort(rng.integers(0, 4 * 10**9, n))     # queue order ~ time order ...
        jitter = rng.random(n) < 0.05
        now[jitter] -= rng.integers(0, 10**9, jitter.sum())                      # ... but not monotone
        q = rng.choice(np.array([1, 1, 1, 2, 0, -1], dtype=np.int64), n)
        ref = orc.batch_slots(slots, burst, count, period, q, now)
        res = eng.rate_limit_batch_slots(slots, max_burst=burst, count_per_period=count, period=period, quantity=q,
                                         now_ns=now)
        assert_same(res, ref, f"{label} round {rnd}")
        assert_state_same(eng, orc, slots)
    eng.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257, 2049])
def test_tiny_and_odd_batch_sizes_all_paths(n):
    """Wave / block / sort-tile edges: every evaluation path on batches around 64, 256 and 2048."""
    import torch
    cap = 300
    rng = np.random.default_rng(n)
    eng, orc = _engine(cap, 4096), _oracle(cap)
    eng.use_torch_stream()
    for rnd in range(3):
        slots = rng.integers(0, 40, n).astype(np.uint32)
        now = T0 + rnd * 10**9
        ref = orc.batch_slots(slots, 3, 10, 60, 1, now)                                   # uniform, host pointers
        assert_same(eng.rate_limit_batch_slots(slots, max_burst=3, count_per_period=10, period=60, quantity=1, now_ns=now), ref, "uniform")
        nows = now + 10**8 + rng.integers(0, 10**8, n)
        q = rng.integers(0, 3, n)
        ref = orc.batch_slots(slots, 3, 10, 60, q, nows)                                  # general, piped device pointers
        ds = torch.from_numpy(slots.astype(np.int32)).cuda()
        dq, dn = torch.from_numpy(q.astype(np.int64)).cuda(), torch.from_numpy(nows.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        res = eng.rate_limit_batch_slots(ds, max_burst=3, count_per_period=10, period=60, quantity=dq, now_ns=dn, inputs_ready=True)
        torch.cuda.synchronize()
        assert_same(res, ref, "general piped")
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


@pytest.mark.parametrize("general", [False, True], ids=["uniform", "per_request_now"])
def test_whole_batch_on_one_key(general):
    """500 000 requests for ONE key in one batch (16 k waves deep): closed form resp. the
    speculative hand-over chain; then the same key again in the next batch."""
    cap, n = 16, 500_000
    eng, orc = _engine(cap, n), _oracle(cap)
    slots = np.full(n, 5, np.uint32)
    for rnd in range(3):
        base = T0 + rnd * 2 * 10**9
        now = base + np.arange(n, dtype=np.int64) * 1000 if general else base      # 1 us apart: 0.5 s per batch
        ref = orc.batch_slots(slots, 100, 1000, 60, 1, now)
        res = eng.rate_limit_batch_slots(slots, max_burst=100, count_per_period=1000, period=60, quantity=1, now_ns=now)
        assert_same(res, ref, f"round {rnd}")
        assert 0 < int(ref.allowed.sum()) < n
    assert_state_same(eng, orc, slots[:1])
    eng.close()


@pytest.mark.parametrize("mode", ["uniform", "general", "unique"])
@pytest.mark.parametrize("dev", [False, True], ids=["host", "device"])
def test_grouped_output_rows(mode, dev):
    """TC_B_GROUPED_OUTPUT: rows come in evaluation order, order[k] names the request of row k;
    un-permuting them must give exactly the normal result (and `order` must be a permutation)."""
    import torch
    cap, n = 900, 40000
    rng = np.random.default_rng(41)
    eng, orc = _engine(cap, n), _oracle(cap)
    eng.use_torch_stream()
    for rnd in range(3):
        if mode == "unique":
            slots = rng.permutation(cap)[:700].astype(np.uint32)
        else:
            slots = ((rng.zipf(1.3, n) * 2654435761) % cap).astype(np.uint32)
        m = slots.size
        if mode == "uniform":
            q, now = 1, T0 + rnd * 10**9
        else:
            q, now = rng.integers(0, 3, m), T0 + rnd * 10**9 + rng.integers(0, 10**9, m)
        ref = orc.batch_slots(slots, 4, 10, 60, q, now)
        want = FIELDS + ("decisions", "allowed_bits")  # (uniform batches: the bits of the grouped rows come from the evaluation's own ballots)
        if dev:
            tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
            res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), max_burst=4, count_per_period=10,
                                             period=60, quantity=q if mode == "uniform" else tt(q),
                                             now_ns=now if mode == "uniform" else tt(now), unique=(mode == "unique"),
                                             want=want, grouped=True)
            torch.cuda.synchronize()
            order = res.order.cpu().numpy().astype(np.int64)
        else:
            res = eng.rate_limit_batch_slots(slots, max_burst=4, count_per_period=10, period=60, quantity=q, now_ns=now,
                                             unique=(mode == "unique"), want=want, grouped=True)
            order = res.order.astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(m)), "order is a permutation of the requests"
        if mode != "unique":   # rows are grouped by key, index order inside a key
            s = slots[order].astype(np.int64)
            assert np.all(np.diff(s) >= 0) and np.all((np.diff(s) > 0) | (np.diff(order) > 0))
        bits = res.allowed_bits.cpu().numpy() if not isinstance(res.allowed_bits, np.ndarray) else res.allowed_bits
        rows_allowed = res.allowed.cpu().numpy() if not isinstance(res.allowed, np.ndarray) else res.allowed
        assert np.array_equal(np.unpackbits(bits.view(np.uint8), bitorder="little")[:m], rows_allowed.astype(np.uint8)), (mode, rnd)
        for f in FIELDS:
            rows = getattr(res, f)
            rows = rows.cpu().numpy() if not isinstance(rows, np.ndarray) else rows
            back = np.empty(m, np.int64)
            back[order] = rows.astype(np.int64)
            assert np.array_equal(back, getattr(ref, f).astype(np.int64)), (mode, f, rnd)
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


@pytest.mark.parametrize("pinned", [True, False], ids=["pinned", "pageable"])
@pytest.mark.parametrize("mode", ["uniform", "general", "mixed", "unique"])
def test_async_host_batches(mode, pinned):
    """TC_B_ASYNC: host-array batches that only enqueue (inputs staged on the grouping stream, outputs
    copied back behind the evaluation).  A ring of 3 buffer sets is cycled 4 times with
    tc_wait_batches(2) before each refill; results and state must equal the sequential oracle,
    also when ordinary synchronous calls are mixed in."""
    import throttlecrab_amd as t
    cap, n, nb, K = (3000, 40000, 12, 3) if mode != "unique" else (50000, 40000, 12, 3)
    rng = np.random.default_rng(91)
    eng, orc = _engine(cap, n), _oracle(cap)

    def buf(count, dtype):
        return eng.host_alloc(count, dtype) if pinned else np.zeros(count, dtype)
    sets = [dict(slots=buf(n, np.uint32), q=buf(n, np.int64), now=buf(n, np.int64),
                 out=t.BatchResult(**{f: buf(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in FIELDS}))
            for _ in range(K)]
    refs, got = [], []
    for bidx in range(nb):
        s = sets[bidx % K]
        if bidx >= K:
            eng.wait_batches(K - 1)            # the batch that used this set is done: keep its results, refill
            got.append({f: getattr(s["out"], f).copy() for f in FIELDS})
        if mode == "unique":
            s["slots"][:] = rng.permutation(cap)[:n].astype(np.uint32)      # no slot twice (n <= cap here)
        else:
            s["slots"][:] = ((rng.zipf(1.2, n) * 2654435761) % cap).astype(np.uint32)
        general = mode == "general" or (mode == "mixed" and bidx % 2 == 1)
        sync_call = mode == "mixed" and bidx % 3 == 2
        if general:
            s["now"][:] = T0 + bidx * 10**8 + rng.integers(0, 10**8, n)
            s["q"][:] = rng.integers(0, 3, n)
            refs.append(orc.batch_slots(s["slots"], 5, 10, 60, s["q"], s["now"]))
            eng.rate_limit_batch_slots(s["slots"], max_burst=5, count_per_period=10, period=60, quantity=s["q"], now_ns=s["now"],
                                       want=FIELDS, out=s["out"], async_=not sync_call)
        else:
            now = T0 + bidx * 10**8
            refs.append(orc.batch_slots(s["slots"], 5, 10, 60, 1, now))
            eng.rate_limit_batch_slots(s["slots"], max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now,
                                       want=FIELDS, out=s["out"], async_=not sync_call, unique=(mode == "unique"))
        if sync_call:
            eng.wait_batches(0)                # (a synchronous call drains everything before it anyway)
    eng.wait_batches(0)
    for bidx in range(nb - K, nb):
        got.append({f: getattr(sets[bidx % K]["out"], f).copy() for f in FIELDS})
    assert len(got) == nb
    for bidx in range(nb):
        assert_same(t.BatchResult(**got[bidx]), refs[bidx], f"async {mode} batch {bidx}")
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


def test_async_flag_misuse():
    import torch
    import throttlecrab_amd as t
    eng = _engine(100, 1000)
    ds = torch.zeros(10, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        eng.rate_limit_batch_slots(ds, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0, async_=True)
    with pytest.raises(ValueError):   # a hidden dtype conversion would hand the engine a temporary
        eng.rate_limit_batch_slots(np.zeros(10, np.int64), max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0,
                                   async_=True)
    eng.wait_batches(0)
    eng.close()


def test_async_host_batches_record_and_bit_outputs():
    """TC_B_ASYNC with the other output forms: decision records, result records, packed bits, grouped rows."""
    import throttlecrab_amd as t
    cap, n = 2000, 10000
    rng = np.random.default_rng(5)
    eng, orc = _engine(cap, n), _oracle(cap)
    outs, refs, slots_all = [], [], []
    for bidx in range(4):
        slots = eng.host_alloc(n, np.uint32)
        slots[:] = ((rng.zipf(1.3, n) * 2654435761) % cap).astype(np.uint32)
        now = T0 + bidx * 10**8
        refs.append(orc.batch_slots(slots, 5, 10, 60, 1, now))
        out = t.BatchResult(decisions=eng.host_alloc(4 * n, np.int64), result4=eng.host_alloc(4 * n, np.int64),
                            allowed_bits=eng.host_alloc((n + 63) // 64, np.uint64), allowed=eng.host_alloc(n, np.uint8),
                            order=eng.host_alloc(n, np.uint32) if bidx % 2 else None)
        eng.rate_limit_batch_slots(slots, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=now,
                                   want=("decisions", "result4", "allowed_bits", "allowed"), out=out, async_=True, grouped=bool(bidx % 2))
        outs.append(out)
        slots_all.append(slots)
    eng.wait_batches(0)
    for bidx, (out, ref) in enumerate(zip(outs, refs)):
        order = out.order.astype(np.int64) if bidx % 2 else np.arange(n)
        dec = t.Engine.unpack_decisions(out.decisions)
        r4 = out.result4.reshape(-1, 4)
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)           # row of request i
        assert np.array_equal(out.allowed[inv], ref.allowed.astype(np.uint8)), f"batch {bidx} allowed"
        for col, f in enumerate(("limit", "remaining", "reset_after_ns", "retry_after_ns")):
            assert np.array_equal(r4[inv, col], getattr(ref, f).astype(np.int64)), f"batch {bidx} result4.{f}"
        bits = np.unpackbits(out.allowed_bits.view(np.uint8), bitorder="little")[:n]
        assert np.array_equal(bits, out.allowed), f"batch {bidx} bits vs bytes (both in row order)"
        assert np.array_equal(dec["remaining"][inv], ref.remaining.astype(np.int64))
        assert np.array_equal(dec["allowed"][inv].astype(np.uint8), ref.allowed.astype(np.uint8))
    assert_state_same(eng, orc, np.arange(cap))
    eng.close()


@pytest.mark.parametrize("registered", [False, True])
def test_small_host_batches_single_launch(registered):
    """Host-pointer batches of at most 1024 requests take k_small_batch (keys resolved, sorted and walked by one
    block, inputs and results in pinned host memory): every size class, heavy duplicates, per-request
    timestamps / quantities / rates, errors, all output columns and both record forms -- against the oracle,
    interleaved with big-pipeline batches on the same engine."""
    import throttlecrab_amd as t
    cap = 700
    rng = np.random.default_rng(23)
    eng, orc = _engine(cap, 5000), _oracle(cap)
    if registered:
        bursts, counts, periods = rng.integers(1, 8, cap), rng.integers(1, 50, cap), rng.integers(1, 90, cap)
        eng.register_params(bursts, counts, periods)
    sizes = [1, 2, 3, 5, 63, 64, 65, 100, 511, 512, 513, 1000, 1023, 1024, 1025, 3000, 7, 900]
    for bidx, n in enumerate(sizes):
        slots = np.where(rng.random(n) < 0.5, rng.integers(0, 12, n), rng.integers(0, cap, n)).astype(np.uint32)
        now = T0 + bidx * 10**9 + rng.integers(0, 10**9, n)
        q = rng.integers(-1, 4, n)
        if registered:
            b, c, p_ = bursts[slots], counts[slots], periods[slots]   # (the oracle has no registration: hand it the plans)
            ref = orc.batch_slots(slots, b, c, p_, q, now)
            res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now,
                                             want=FIELDS + ("result4", "decisions"))
        else:
            b = rng.integers(0, 8, n)           # burst 0: InvalidRateLimit
            c, p_ = rng.integers(1, 50, n), rng.integers(1, 90, n)
            ref = orc.batch_slots(slots, b, c, p_, q, now)
            res = eng.rate_limit_batch_slots(slots, max_burst=b, count_per_period=c, period=p_, quantity=q, now_ns=now,
                                             want=FIELDS + ("result4", "decisions"))
        assert_same(res, ref, f"small batch {bidx} (n={n})")
        ok = ref.status == 0
        r4 = res.result4.reshape(-1, 4)
        for col, f in enumerate(("limit", "remaining", "reset_after_ns", "retry_after_ns")):
            assert np.array_equal(r4[:, col], getattr(ref, f).astype(np.int64)), f"batch {bidx} result4.{f}"
        dec = t.Engine.unpack_decisions(res.decisions)
        assert np.array_equal(dec["status"], ref.status.astype(np.uint8)) and np.array_equal(dec["remaining"][ok], ref.remaining[ok])
    assert_state_same(eng, orc, np.arange(cap))
    c = eng.counters()
    assert c["total"] == sum(sizes)
    eng.close()
