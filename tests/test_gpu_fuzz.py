"""Model-based fuzzing: random interleavings of every operation the C ABI offers (uniform / general /
unique batches, host and device pointers, pipelined or not, registered plans, single-key store
operations with extreme values, sweeps, snapshot + restore) against the CPU oracle.  The point is
the INTERACTIONS: e.g. a closed-form batch right after a store operation planted i64::MAX in a cell,
a direct-store batch after a sweep, a pipelined batch next to a host-pointer one."""
import numpy as np
import pytest

from tests import kat

import os

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
# TC_FUZZ_SEEDS=N widens the sweep (soak runs); the default keeps the suite short
N_SLOT = int(os.environ.get("TC_FUZZ_SEEDS", "6"))
N_KEYS = int(os.environ.get("TC_FUZZ_SEEDS", "3"))
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")
PLANS = [(5, 10, 60), (100, 1000, 3600), (2, 120, 60), (1, 1, 1), (10, 2**62, 60), (3, 7, 60), (2**32 + 1, 10, 60),
         (2**63 - 1, 2**63 - 1, 2**63 - 1), (20, 600, 60)]
EXTREME = [0, 1, -1, 2**63 - 1, -2**63, T0, T0 + 10**12, -T0]


def _same(res, ref, ctx):
    for f in FIELDS:
        got = getattr(res, f)
        if got is None:
            continue
        if not isinstance(got, np.ndarray):
            got = got.cpu().numpy()
        bad = np.nonzero(got.astype(np.int64) != getattr(ref, f).astype(np.int64))[0]
        assert bad.size == 0, f"{ctx}: {f} differs at {bad[:6]}: got {got[bad[:6]]} want {getattr(ref, f)[bad[:6]]}"


@pytest.mark.parametrize("seed", range(N_SLOT))
def test_fuzz_slot_mode(seed, tmp_path):
    import torch
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    cap, nmax = 700, 6000
    eng, orc = t.Engine(cap, nmax), O.DenseOracle(cap)
    eng.check_on_close = True
    eng.use_torch_stream()
    now = T0
    registered = None  # per-slot plan index once registered
    log = []
    for step in range(60):
        now += int(rng.integers(-2 * 10**8, 3 * 10**9))          # mostly forward, sometimes back
        now = max(now, 1)
        op = rng.choice(["uniform", "uniform", "general", "unique", "store", "sweep", "register", "snapshot"],
                        p=[0.3, 0.15, 0.2, 0.08, 0.12, 0.07, 0.04, 0.04])
        log.append(op)
        n = int(rng.integers(1, nmax))
        hot = rng.random() < 0.5
        slots = (rng.zipf(1.3, n) % cap if hot else rng.integers(0, cap, n)).astype(np.uint32)
        if op == "uniform":
            b, c, p = PLANS[rng.integers(0, len(PLANS))]
            q = int(rng.choice([1, 1, 1, 2, 0, 7]))
            use_reg = registered is not None and rng.random() < 0.5
            if use_reg:
                pl = np.array(PLANS, dtype=object)[registered[slots]]
                ref = orc.batch_slots(slots, np.array([x[0] for x in pl]), np.array([x[1] for x in pl]),
                                      np.array([x[2] for x in pl]), q, now)
                kw = dict(registered=True)
            else:
                ref = orc.batch_slots(slots, b, c, p, q, now)
                kw = dict(max_burst=b, count_per_period=c, period=p)
            if rng.random() < 0.5:
                ds = torch.from_numpy(slots.astype(np.int32)).cuda()
                torch.cuda.synchronize()
                res = eng.rate_limit_batch_slots(ds, quantity=q, now_ns=now, inputs_ready=bool(rng.random() < 0.7), **kw)
                torch.cuda.synchronize()
            else:
                res = eng.rate_limit_batch_slots(slots, quantity=q, now_ns=now, **kw)
            _same(res, ref, f"seed {seed} step {step} uniform {log[-5:]}")
        elif op == "general":
            pi = rng.integers(0, len(PLANS), n)
            pl = np.array(PLANS, dtype=object)[pi]
            b, c, p = (np.array([x[k] for x in pl]) for k in range(3))
            q = rng.choice(np.array([0, 1, 1, 2, -1, 2**62], dtype=np.int64), n)
            nows = now + rng.integers(-10**9, 10**9, n)
            nows[rng.random(n) < 0.01] = -5
            ref = orc.batch_slots(slots, b, c, p, q, nows)
            res = eng.rate_limit_batch_slots(slots, max_burst=b, count_per_period=c, period=p, quantity=q, now_ns=nows)
            _same(res, ref, f"seed {seed} step {step} general")
        elif op == "unique":
            slots = rng.permutation(cap)[: min(n, cap)].astype(np.uint32)
            q = rng.integers(0, 4, slots.size)
            ref = orc.batch_slots(slots, 3, 30, 60, q, now)
            res = eng.rate_limit_batch_slots(slots, max_burst=3, count_per_period=30, period=60, quantity=q, now_ns=now, unique=True)
            _same(res, ref, f"seed {seed} step {step} unique")
        elif op == "store":
            for _ in range(20):
                s = int(rng.integers(0, cap))
                key = s.to_bytes(4, "little")
                kind = rng.integers(0, 3)
                v, w = int(rng.choice(EXTREME)), int(rng.choice(EXTREME))
                ttl = int(rng.choice([0, 1, 10**9, 60 * 10**9, 2**63, 2**64 - 1]))
                if kind == 0:
                    assert eng.get(key, now) == orc.get(key, now)
                elif kind == 1:
                    cur = orc.get(key, now)
                    old = cur if (cur is not None and rng.random() < 0.7) else v
                    assert eng.compare_and_swap_with_ttl(key, old, w, ttl, now) == orc.compare_and_swap_with_ttl(key, old, w, ttl, now)
                else:
                    assert eng.set_if_not_exists_with_ttl(key, v, ttl, now) == orc.set_if_not_exists_with_ttl(key, v, ttl, now)
        elif op == "sweep":
            assert eng.sweep_expired(now) == orc.sweep(now), f"seed {seed} step {step} sweep"
        elif op == "register":
            registered = rng.integers(0, 4, cap)     # only the first four plans: all valid
            pl = np.array(PLANS, dtype=object)[registered]
            eng.register_params(np.array([x[0] for x in pl]), np.array([x[1] for x in pl]), np.array([x[2] for x in pl]))
        elif op == "snapshot":
            path = str(tmp_path / f"f{seed}_{step}.snap")
            eng.snapshot_save(path)
            eng.check_on_close = False
            eng.close()
            eng = t.Engine(cap, nmax)
            eng.check_on_close = True
            eng.use_torch_stream()
            eng.snapshot_load(path)
        if step % 10 == 9:
            tat, exp = eng.read_state(0, cap)
            for s in range(cap):
                ot, oe, occ = orc.peek(s)
                assert (int(exp[s]) == 0) if not occ else ((int(tat[s]), int(exp[s])) == (ot, oe)), (seed, step, s, log[-10:])
    eng.close()


@pytest.mark.parametrize("seed", range(N_KEYS))
def test_fuzz_string_mode(seed):
    import torch
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(2000 + seed)
    keys = [b"k%d" % i for i in range(1500)] + [b"", b"\xf0\x9f\xa6\x80"] + [b"L" * 100 + b"%d" % i for i in range(30)]
    eng = t.Engine(4096, 5000, key_mode=True, track_denied=True)
    eng.check_on_close = True
    eng.use_torch_stream()
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    now = T0
    for step in range(40):
        now += int(rng.integers(0, 4 * 10**9))
        op = rng.choice(["uniform", "general", "store", "sweep"], p=[0.45, 0.3, 0.15, 0.1])
        n = int(rng.integers(1, 5000))
        idx = np.minimum(rng.zipf(1.25, n) - 1, len(keys) - 1) if rng.random() < 0.6 else rng.integers(0, len(keys), n)
        kb, ko = O.pack_keys([keys[i] for i in idx])
        if op == "uniform":
            b, c, p = PLANS[rng.integers(0, len(PLANS))]
            ref = orc.batch_keys(kb, ko, b, c, p, 1, now)
            if rng.random() < 0.5:
                dkb, dko = torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda()
                torch.cuda.synchronize()
                res = eng.rate_limit_batch_keys(dkb, dko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now,
                                                inputs_ready=bool(rng.random() < 0.7))
                torch.cuda.synchronize()
            else:
                res = eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now)
            _same(res, ref, f"seed {seed} step {step} uniform keys")
        elif op == "general":
            pl = np.array(PLANS, dtype=object)[rng.integers(0, len(PLANS), n)]
            b, c, p = (np.array([x[k] for x in pl]) for k in range(3))
            q = rng.choice(np.array([0, 1, 1, 3, -1], dtype=np.int64), n)
            nows = now + rng.integers(0, 10**9, n)        # forward only: cleanup stays decision-neutral
            ref = orc.batch_keys(kb, ko, b, c, p, q, nows)
            res = eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=q, now_ns=nows)
            _same(res, ref, f"seed {seed} step {step} general keys")
            now = int(nows.max())
        elif op == "store":
            for _ in range(15):
                key = keys[int(rng.integers(0, len(keys)))]
                v = int(rng.choice(EXTREME))
                ttl = int(rng.choice([0, 10**9, 60 * 10**9]))
                if rng.random() < 0.5:
                    assert eng.get(key, now) == orc.get(key, now)
                else:
                    assert eng.set_if_not_exists_with_ttl(key, v, ttl, now) == orc.set_if_not_exists_with_ttl(key, v, ttl, now)
        else:
            before = len(orc)
            orc.force_cleanup(now)
            assert eng.sweep_expired(now) == before - len(orc)
            assert eng.counters()["live_slots"] == len(orc)
    for k in keys[::11]:
        assert eng.get(k, now) == orc.get(k, now)
    eng.close()


SHORT_PLANS = [(5, 10, 1), (2, 60, 60), (10, 100, 60), (3, 7, 60), (1, 1, 1)]   # entries live 0.6 .. 34 s


@pytest.mark.parametrize("seed", range(N_KEYS))
def test_fuzz_string_mode_while_the_engine_cleans_itself(seed):
    """Round 5: the same interleavings with a cleanup policy set (tc_set_sweep_policy) and NO explicit sweep -- the engine decides
    when to clean (time, operation count, size), in front of synchronous, pipelined and TC_B_ASYNC batches, single requests and
    store operations.  Time only moves forward (under which cleanup is decision-neutral, adaptive_cleanup.rs:248): every decision
    must be the oracle's, which never cleans; at the end one explicit sweep on both sides must leave the same number of entries."""
    import torch
    import throttlecrab_amd as t
    from oracle import oracle as O
    rng = np.random.default_rng(3000 + seed)
    keys = [b"k%d" % i for i in range(1500)] + [b"", b"\xf0\x9f\xa6\x80"] + [b"L" * 100 + b"%d" % i for i in range(30)]
    eng = t.Engine(2048, 5000, key_mode=True, track_denied=True)
    eng.check_on_close = True
    eng.use_torch_stream()
    kind = ("adaptive", "adaptive", "periodic", "probabilistic")[seed % 4]
    if kind == "adaptive":
        eng.set_sweep_policy("adaptive", created_ns=T0, min_interval_ns=10**9, max_interval_ns=8 * 10**9,
                             max_operations=int(rng.integers(200, 3000)), map_capacity=int(rng.choice([0, 512])))
    elif kind == "periodic":
        eng.set_sweep_policy("periodic", created_ns=T0, interval_ns=2 * 10**9)
    else:
        eng.set_sweep_policy("probabilistic", cleanup_probability=7)
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    now = T0
    pending = []   # TC_B_ASYNC batches not yet waited for: (result, reference, context)

    def drain():
        eng.wait_batches()
        for res, ref, ctx, _keep in pending:   # (_keep: the input arrays of a TC_B_ASYNC call live until it was waited for)
            _same(res, ref, ctx)
        pending.clear()

    for step in range(50):
        now += int(rng.integers(0, 4 * 10**9))
        op = rng.choice(["uniform", "general", "async", "single", "store"], p=[0.35, 0.25, 0.15, 0.1, 0.15])
        n = int(rng.integers(1, 5000))
        idx = np.minimum(rng.zipf(1.25, n) - 1, len(keys) - 1) if rng.random() < 0.6 else rng.integers(0, len(keys), n)
        kb, ko = O.pack_keys([keys[i] for i in idx])
        ko = ko.astype(np.uint32)
        if op != "async":
            drain()
        if op == "uniform":
            b, c, p = SHORT_PLANS[rng.integers(0, len(SHORT_PLANS))]
            ref = orc.batch_keys(kb, ko, b, c, p, 1, now)
            if rng.random() < 0.5:
                dkb, dko = torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda()
                torch.cuda.synchronize()
                res = eng.rate_limit_batch_keys(dkb, dko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now,
                                                inputs_ready=bool(rng.random() < 0.7))
                torch.cuda.synchronize()
            else:
                res = eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now)
            _same(res, ref, f"seed {seed} step {step} uniform keys ({kind})")
        elif op in ("general", "async"):
            pl = np.array(SHORT_PLANS, dtype=object)[rng.integers(0, len(SHORT_PLANS), n)]
            b, c, p = (np.array([x[k] for x in pl], dtype=np.int64) for k in range(3))
            q = rng.choice(np.array([0, 1, 1, 3, -1], dtype=np.int64), n)
            nows = now + np.sort(rng.integers(0, 10**9, n)).astype(np.int64)   # ascending in index order: time never runs back
            ref = orc.batch_keys(kb, ko, b, c, p, q, nows)
            if op == "async":
                out = t.BatchResult(**{f: eng.host_alloc(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in FIELDS})
                cols = dict(max_burst=b, count_per_period=c, period=p, quantity=q, now_ns=nows)
                keep = (kb, ko, cols)
                res = eng.rate_limit_batch_keys(kb, ko, want=FIELDS, out=out, async_=True, **cols)
                pending.append((res, ref, f"seed {seed} step {step} async keys ({kind})", keep))
                if len(pending) >= 3:
                    drain()
            else:
                res = eng.rate_limit_batch_keys(kb, ko, max_burst=b, count_per_period=c, period=p, quantity=q, now_ns=nows)
                _same(res, ref, f"seed {seed} step {step} general keys ({kind})")
            now = int(nows.max())
        elif op == "single":
            for _ in range(10):
                key = keys[int(rng.integers(0, len(keys)))]
                b, c, p = SHORT_PLANS[rng.integers(0, len(SHORT_PLANS))]
                now += int(rng.integers(0, 10**8))
                k1, o1 = O.pack_keys([key])
                ref = orc.batch_keys(k1, o1, b, c, p, 1, now)
                st, allowed, limit, remaining, reset, retry = eng.rate_limit(key, b, c, p, 1, now)
                assert (st, int(allowed), limit, remaining, reset, retry) == (int(ref.status[0]), int(ref.allowed[0]), int(ref.limit[0]),
                        int(ref.remaining[0]), int(ref.reset_after_ns[0]), int(ref.retry_after_ns[0])), (seed, step, key)
        else:
            for _ in range(15):
                key = keys[int(rng.integers(0, len(keys)))]
                v = int(rng.choice(EXTREME))
                ttl = int(rng.choice([0, 10**9, 5 * 10**9]))
                kindop = rng.integers(0, 3)
                if kindop == 0:
                    assert eng.get(key, now) == orc.get(key, now)
                elif kindop == 1:
                    assert eng.set_if_not_exists_with_ttl(key, v, ttl, now) == orc.set_if_not_exists_with_ttl(key, v, ttl, now)
                else:
                    cur = orc.get(key, now)
                    old = cur if (cur is not None and rng.random() < 0.7) else v
                    assert eng.compare_and_swap_with_ttl(key, old, v, ttl, now) == orc.compare_and_swap_with_ttl(key, old, v, ttl, now)
    drain()
    st = eng.sweep_stats()
    assert st["kind"] == kind and st["sweeps"] > 0, st
    for k in keys[::11]:
        assert eng.get(k, now) == orc.get(k, now)
    orc.force_cleanup(now)
    eng.sweep_expired(now)
    assert eng.counters()["live_slots"] == len(orc) and eng.debug_check_keys() == 0, (st, eng.counters(), len(orc))
    eng.close()
