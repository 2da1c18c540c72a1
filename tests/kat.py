"""Shared known-answer replay: runs tests/golden/reference_kat.json through any
limiter exposing rate_limit(key: bytes, burst, count, period, q, now_ns) ->
(status, allowed, limit, remaining, reset_after_ns, retry_after_ns)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kat.json")


def load():
    with open(GOLDEN) as f:
        return json.load(f)


def check_step(name, i, st, got):
    status, allowed, limit, remaining, reset_ns, retry_ns = got
    e = st["expect"]
    where = f"{name} step {i}: {st['key']!r} ({st['burst']},{st['count']},{st['period']}) q={st['q']}"
    if "status" in e:
        if e["status"] == "err":
            assert status != 0, where
        else:
            assert status == e["status"], where
    if e.get("status") == "err":
        return
    for k in ("allowed", "limit", "remaining"):
        if k in e:
            assert status == 0, where
            assert {"allowed": allowed, "limit": limit, "remaining": remaining}[k] == e[k], f"{where}: {k}"
    if "remaining_gt" in e:
        assert remaining > e["remaining_gt"], where
    if "remaining_lt" in e:
        assert remaining < e["remaining_lt"], where
    if "remaining_ge" in e:
        assert remaining >= e["remaining_ge"], where
    if "remaining_le" in e:
        assert remaining <= e["remaining_le"], where
    if "retry_after_s_gt" in e:
        assert retry_ns // 10**9 > e["retry_after_s_gt"], where
    if "reset_after_s" in e:
        assert reset_ns // 10**9 == e["reset_after_s"], where
    if "retry_after_s" in e:
        assert retry_ns // 10**9 == e["retry_after_s"], where
    if "reset_after_ns" in e:
        assert reset_ns == e["reset_after_ns"], f"{where}: reset_after {reset_ns} ns, want {e['reset_after_ns']}"
    if "retry_after_ns" in e:
        assert retry_ns == e["retry_after_ns"], f"{where}: retry_after {retry_ns} ns, want {e['retry_after_ns']}"


def replay_scenario(sc, limiter):
    n_allowed = 0
    results = []
    for i, st in enumerate(sc["steps"]):
        got = limiter.rate_limit(st["key"].encode("utf-8"), st["burst"], st["count"], st["period"], st["q"], st["now"])
        check_step(sc["name"], i, st, got)
        n_allowed += int(bool(got[1]) and got[0] == 0)
        results.append(got)
    if "allowed_total" in sc:
        assert n_allowed == sc["allowed_total"], sc["name"]
    return results


def _sweep(store, now):
    """AdaptiveStore::cleanup at `now`, whatever the store under test calls it"""
    for name in ("sweep_expired", "force_cleanup", "cleanup", "sweep"):
        if hasattr(store, name):
            return getattr(store, name)(now)
    raise AssertionError("store has no cleanup entry point")


def _live(store):
    if hasattr(store, "live_count"):
        return store.live_count()
    if hasattr(store, "counters"):  # the engine: its live-slot counter is refreshed by a sweep (at time 0 nothing expires)
        store.sweep_expired(0)
        return store.counters()["live_slots"]
    return len(store)


def replay_store_contract(case, store, t0, explicit_sweeps=True):
    """explicit_sweeps=False: the ["sweep", t] steps are left out -- the store under test must clean itself where the
    reference's does (inside its own set_if_not_exists / compare_and_swap, adaptive_cleanup.rs:205-211,229,262)."""
    for op in case["ops"]:
        kind = op[0]
        key = op[1].encode("utf-8") if isinstance(op[1], str) else None
        if kind == "set_nx":
            _, _, val, ttl, t, exp = op
            assert store.set_if_not_exists_with_ttl(key, val, ttl, t0 + t) == exp, (case["name"], op[:2])
        elif kind == "get":
            _, _, t, exp = op
            assert store.get(key, t0 + t) == exp, (case["name"], op[:2])
        elif kind == "cas":
            _, _, old, new, ttl, t, exp = op
            assert store.compare_and_swap_with_ttl(key, old, new, ttl, t0 + t) == exp, (case["name"], op[:2])
        elif kind == "sweep":
            if explicit_sweeps:
                _sweep(store, t0 + op[1])
        elif kind == "len":
            n = _live(store)
            assert op[1] <= n <= op[2], (case["name"], op, n)
        else:
            raise AssertionError(kind)
