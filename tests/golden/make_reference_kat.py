#!/usr/bin/env python3
"""Transcribes the reference's known-answer tests for the GCRA hot path into
tests/golden/reference_kat.json.

The reference is Rust and cannot be executed in this image, so these vectors
are *transcriptions of the reference's own assertions* (every scenario cites
the test it comes from, paths relative to /root/reference).  Only what the
reference asserts is recorded; nothing here is computed by our code.

All reference tests use a synthetic clock: one SystemTime::now() (t0) plus
Duration offsets, so t0 is arbitrary; T0 below is used everywhere.

Expectation keys per step (all optional):
  status            0 Ok | 1 NegativeQuantity | 2 InvalidRateLimit | "err" (any Err)
  allowed           bool
  limit, remaining  exact i64
  remaining_gt / remaining_lt / remaining_ge / remaining_le   strict/loose bounds
  retry_after_s_gt  retry_after.as_secs() > value
  reset_after_s / retry_after_s      exact seconds after Duration::as_secs()
                    truncation (throttlecrab-server/src/types.rs:87-96)
  reset_after_ns / retry_after_ns    exact nanoseconds.  The reference asserts whole seconds at most; these
                    values are derived BY HAND from rate_limiter.rs:151-238 (each carries its derivation in a
                    comment below) so that the nanosecond outputs are pinned by something other than our own code.
Scenario-level:  "allowed_total": N  -> exactly N of the steps are allowed.

Store-contract ops (store_test_suite.rs, cleanup_test.rs):
  ["set_nx", key, val, ttl_ns, t, expect_bool] | ["get", key, t, expect_or_null] | ["cas", key, old, new, ttl_ns, t, expect_bool]
  ["sweep", t]            run the store's cleanup at t0 + t (AdaptiveStore::cleanup; the reference's stores trigger it
                          from their own heuristics, the engine's caller calls tc_sweep_expired)
  ["len", lo, hi]         lo <= live entries <= hi (store.len() in the reference's tests)
"""
import json
import os

T0 = 1_700_000_000 * 10**9
S = 10**9
MS = 10**6
I64_MAX = 2**63 - 1

CORE = "throttlecrab/src/core/tests.rs"
SUITE = "throttlecrab/src/core/store/store_test_suite.rs"
REDIS = "throttlecrab-server/src/transport/redis_test.rs"

scenarios = []


def step(key, burst, count, period, q, t=0, **expect):
    return {"key": key, "burst": burst, "count": count, "period": period, "q": q,
            "now": T0 + t, "expect": expect}


def scenario(name, source, steps, **extra):
    d = {"name": name, "source": source, "steps": steps}
    d.update(extra)
    scenarios.append(d)


# ---- throttlecrab/src/core/tests.rs ---------------------------------------
scenario("basic_rate_limiting", f"{CORE}:5-14",
         [step("test", 5, 10, 60, 1, allowed=True, limit=5, remaining=4, status=0)])

# (5, 10, 60): ei = 60e9 / 10 = 6e9 ns, dvt = ei * (5 - 1) = 24e9 ns.  Request i (0-based) of the burst at t0:
#   fresh key: tat = t0 - ei (:162-166), new_tat = t0 + i * ei, allowed while new_tat - dvt <= t0 (i <= 4),
#   reset_after = (new_tat - t0) + dvt = (i + 4) * 6e9 (:227-232), retry_after = 0.
# Sixth: tat = t0 + 24e9, new_tat = t0 + 30e9, allow_at = t0 + 6e9 > t0: denied; cur = tat:
#   reset_after = 24e9 + 24e9, retry_after = allow_at - t0 = 6e9 (:234-238).
scenario("burst_capacity", f"{CORE}:17-33",
         [step("burst_test", 5, 10, 60, 1, allowed=True, remaining=5 - (i + 1), reset_after_ns=(i + 4) * 6 * S, retry_after_ns=0)
          for i in range(5)]
         + [step("burst_test", 5, 10, 60, 1, allowed=False, remaining=0, retry_after_s_gt=0, reset_after_ns=48 * S,
                 retry_after_ns=6 * S)])

# (2, 60, 60): ei = 1e9, dvt = 1e9.  #1 new_tat = t0: reset 1e9.  #2 new_tat = t0 + 1e9, allow_at = t0: allowed, reset 2e9.
# #3 new_tat = t0 + 2e9, allow_at = t0 + 1e9 > t0: denied, cur = t0 + 1e9: reset 2e9, retry 1e9, remaining 0.
# #4 at t0 + 1 s: entry live (expiry t0 + 2e9), tat = t0 + 1e9 = now, new_tat = now + 1e9, allow_at = now: allowed,
#    remaining = (now + dvt - new_tat) / ei = 0, reset = 1e9 + 1e9.
scenario("rate_replenishment", f"{CORE}:36-62",
         [step("replenish_test", 2, 60, 60, 1, allowed=True, reset_after_ns=1 * S, retry_after_ns=0),
          step("replenish_test", 2, 60, 60, 1, allowed=True, reset_after_ns=2 * S, retry_after_ns=0),
          step("replenish_test", 2, 60, 60, 1, allowed=False, reset_after_ns=2 * S, retry_after_ns=1 * S),
          step("replenish_test", 2, 60, 60, 1, t=1 * S, allowed=True, reset_after_ns=2 * S, retry_after_ns=0)])

scenario("different_keys", f"{CORE}:65-91",
         [step("key1", 2, 2, 60, 1, allowed=True),
          step("key2", 2, 2, 60, 1, allowed=True),
          step("key1", 2, 2, 60, 1, allowed=True),
          step("key1", 2, 2, 60, 1, allowed=False),
          step("key2", 2, 2, 60, 1, allowed=True),
          step("key2", 2, 2, 60, 1, allowed=False)])

# (10, 10, 60): ei = 6e9, dvt = 54e9.  q5 fresh: new_tat = t0 - 6e9 + 30e9 = t0 + 24e9: reset 24e9 + 54e9 = 78e9.
# q6: new_tat = t0 + 60e9, allow_at = t0 + 6e9: denied; cur = t0 + 24e9: reset 78e9, retry 6e9.
# q5: new_tat = t0 + 54e9, allow_at = t0: allowed, reset 54e9 + 54e9 = 108e9.
scenario("quantity_parameter", f"{CORE}:94-118",
         [step("quantity_test", 10, 10, 60, 5, allowed=True, remaining=5, reset_after_ns=78 * S, retry_after_ns=0),
          step("quantity_test", 10, 10, 60, 6, allowed=False, remaining=5, reset_after_ns=78 * S, retry_after_ns=6 * S),
          step("quantity_test", 10, 10, 60, 5, allowed=True, remaining=0, reset_after_ns=108 * S, retry_after_ns=0)])

scenario("negative_quantity_error", f"{CORE}:121-127",
         [step("negative_test", 10, 10, 60, -1, status="err")])

scenario("invalid_parameters", f"{CORE}:130-145",
         [step("test", 0, 10, 60, 1, status="err"),
          step("test", 10, 0, 60, 1, status="err"),
          step("test", 10, 10, 0, 1, status="err")])

scenario("large_quantity_overflow_protection", f"{CORE}:148-160",
         [step("overflow_test", 10, 10, 60, I64_MAX // 2, status=0, allowed=False)])

scenario("saturating_arithmetic", f"{CORE}:163-176",
         [step("saturate_test", I64_MAX // 1000, 100, 60, 1, status=0),
          step("saturate_test2", 10, I64_MAX // 1000, 60, 1, status=0)])

scenario("remaining_count_accuracy", f"{CORE}:179-296",
         [step("remaining_test", 5, 10, 60, 1, allowed=True, remaining=4)]
         + [step("remaining_test", 5, 10, 60, 1, allowed=True, remaining=5 - i) for i in range(2, 6)]
         + [step("remaining_test", 5, 10, 60, 1, allowed=False, remaining=0, retry_after_s_gt=0),
            step("remaining_test", 5, 10, 60, 1, t=6 * S, allowed=True, remaining=0),
            step("remaining_test", 5, 10, 60, 1, t=6 * S, allowed=False, remaining=0),
            step("quantity_remaining", 5, 10, 60, 3, allowed=True, remaining=2),
            step("quantity_remaining", 5, 10, 60, 3, allowed=False, remaining=2),
            step("quantity_remaining", 5, 10, 60, 2, allowed=True, remaining=0),
            step("high_rate", 10, 600, 60, 1, allowed=True, remaining=9)]
         + [step("high_rate", 10, 600, 60, 1) for _ in range(9)]
         + [step("high_rate", 10, 600, 60, 1, t=1 * S, allowed=True, remaining_lt=10)])

scenario("remaining_count_all_stores(AdaptiveStore)", f"{CORE}:299-347",
         [step("test_key", 3, 6, 60, 1, allowed=True, remaining=3 - i) for i in range(1, 4)]
         + [step("test_key", 3, 6, 60, 1, allowed=False, remaining=0),
            step("test_key", 3, 6, 60, 1, t=10 * S, allowed=True, remaining=0)])

# the same sequence is asserted for every store type (test_all_stores!): the engine claims decision parity with all three
# (cleanup policies differ, decisions do not: adaptive_cleanup.rs:246-278 / periodic.rs:175-209 / probabilistic.rs:157-183)
for _store in ("PeriodicStore", "ProbabilisticStore"):
    scenario(f"remaining_count_all_stores({_store})", f"{CORE}:299-347",
             [step("test_key", 3, 6, 60, 1, allowed=True, remaining=3 - i) for i in range(1, 4)]
             + [step("test_key", 3, 6, 60, 1, allowed=False, remaining=0),
                step("test_key", 3, 6, 60, 1, t=10 * S, allowed=True, remaining=0)])

scenario("edge_cases_zero_remaining", f"{CORE}:350-412",
         [step("exact_timing", 2, 120, 60, 1, allowed=True, remaining=1),
          step("exact_timing", 2, 120, 60, 1, allowed=True, remaining=0),
          step("exact_timing", 2, 120, 60, 1, t=500 * MS, allowed=True, remaining=0),
          step("zero_period", 10, 10, 0, 1, status="err"),
          # (3, 7, 60): ei = trunc(60e9 / 7) = 8 571 428 571, dvt = 2 ei = 17 142 857 142.  new_tat = t0, t0 + ei, t0 + 2 ei:
          # reset = dvt, ei + dvt, 2 ei + dvt.  At t0 + 8 s: new_tat = t0 + 3 ei, allow_at = t0 + ei > now: denied,
          # retry = ei - 8e9 = 571 428 571, reset = (2 ei - 8e9) + dvt = 26 285 714 284.  At t0 + 9 s: allowed,
          # cur = t0 + 3 ei: reset = (3 ei - 9e9) + dvt = 33 857 142 855.
          step("fractional", 3, 7, 60, 1, allowed=True, remaining=2, reset_after_ns=17142857142, retry_after_ns=0),
          step("fractional", 3, 7, 60, 1, reset_after_ns=25714285713, retry_after_ns=0),
          step("fractional", 3, 7, 60, 1, reset_after_ns=34285714284, retry_after_ns=0),
          step("fractional", 3, 7, 60, 1, t=8 * S, allowed=False, reset_after_ns=26285714284, retry_after_ns=571428571),
          step("fractional", 3, 7, 60, 1, t=9 * S, allowed=True, remaining=0, reset_after_ns=33857142855, retry_after_ns=0),
          step("max_burst", I64_MAX // 1000, 100, 60, 1, allowed=True, remaining_gt=0)])

_grad = [step("multi_quantity", 10, 60, 60, 5, allowed=True, remaining=5),
         step("multi_quantity", 10, 60, 60, 6, allowed=False, remaining=5),
         step("multi_quantity", 10, 60, 60, 5, allowed=True, remaining=0),
         step("multi_quantity", 10, 60, 60, 2, t=3 * S, allowed=True, remaining=1)]
_grad += [step("gradual_replenish", 5, 120, 60, 1) for _ in range(5)]
for millis, _avail, rem in [(500, 1, 0), (1000, 2, 1), (1500, 3, 2), (2000, 4, 3), (2500, 5, 4)]:
    k = f"gradual_replenish_{millis}"
    _grad += [step(k, 5, 120, 60, 1) for _ in range(5)]
    _grad += [step(k, 5, 120, 60, 1, t=millis * MS, allowed=True, remaining=rem)]
scenario("quantity_variations_and_replenishment", f"{CORE}:415-500", _grad)

# (8, 240, 60): ei = 0.25e9, dvt = 1.75e9.  q6 fresh: new_tat = t0 - ei + 6 ei = t0 + 1.25e9: reset 1.25e9 + 1.75e9 = 3e9.
# +0.5 s q1: new_tat = t0 + 1.5e9: reset (1.5e9 - 0.5e9) + 1.75e9 = 2.75e9.  +1.5 s q1: new_tat = t0 + 1.75e9: reset 0.25e9 + 1.75e9.
_cx = [step("partial_burst", 8, 240, 60, 6, allowed=True, remaining=2, reset_after_ns=3000 * MS, retry_after_ns=0),
       step("partial_burst", 8, 240, 60, 1, t=500 * MS, allowed=True, remaining=3, reset_after_ns=2750 * MS, retry_after_ns=0),
       step("partial_burst", 8, 240, 60, 1, t=1500 * MS, allowed=True, remaining=6, reset_after_ns=2000 * MS, retry_after_ns=0)]
# (3, 6, 60): ei = 10e9, dvt = 20e9.  Three at t0: new_tat = t0, t0 + 10e9, t0 + 20e9 (reset 20e9, 30e9, 40e9).
# +5 s: new_tat = t0 + 30e9, allow_at = t0 + 10e9 > now: denied, retry 5e9, cur = t0 + 20e9: reset 15e9 + 20e9.
# +10 s: allow_at = now: allowed, cur = t0 + 30e9: reset 20e9 + 20e9.  +20 s: new_tat = t0 + 40e9: reset 20e9 + 20e9.
_cx += [step("slow_replenish", 3, 6, 60, 1, reset_after_ns=(20 + 10 * i) * S, retry_after_ns=0) for i in range(3)]
_cx += [step("slow_replenish", 3, 6, 60, 1, t=5 * S, allowed=False, reset_after_ns=35 * S, retry_after_ns=5 * S),
        step("slow_replenish", 3, 6, 60, 1, t=10 * S, allowed=True, remaining=0, reset_after_ns=40 * S, retry_after_ns=0),
        step("slow_replenish", 3, 6, 60, 1, t=20 * S, allowed=True, remaining=0, reset_after_ns=40 * S, retry_after_ns=0)]
_cx += [step("fractional_accumulation", 5, 100, 60, 1) for _ in range(5)]
for millis, rem in [(600, 0), (1200, 1), (1800, 2), (2400, 3), (3000, 4)]:
    k = f"fractional_accumulation_{millis}"
    _cx += [step(k, 5, 100, 60, 1) for _ in range(5)]
    _cx += [step(k, 5, 100, 60, 1, t=millis * MS, allowed=True, remaining=rem)]
scenario("complex_replenishment_scenarios", f"{CORE}:503-601", _cx)

scenario("quantity_edge_cases", f"{CORE}:604-655",
         [step("zero_quantity", 10, 100, 60, 0, allowed=True, remaining=10),
          step("neg_quantity", 10, 100, 60, -5, status="err"),
          step("large_quantity", 5, 100, 60, 10, allowed=False, remaining=5),
          step("exact_burst", 10, 100, 60, 10, allowed=True, remaining=0),
          # (20, 600, 60): ei = 0.1e9, dvt = 1.9e9.  q15 fresh: new_tat = t0 + 1.4e9: reset 1.4e9 + 1.9e9.
          # +1 s q12: new_tat = t0 + 2.6e9, allow_at = t0 + 0.7e9 <= now: allowed, reset 1.6e9 + 1.9e9.
          # q5: new_tat = t0 + 3.1e9, allow_at = t0 + 1.2e9 > now: denied, retry 0.2e9, cur = t0 + 2.6e9: reset 3.5e9.
          step("large_quantity_replenish", 20, 600, 60, 15, allowed=True, remaining=5, reset_after_ns=3300 * MS, retry_after_ns=0),
          step("large_quantity_replenish", 20, 600, 60, 12, t=1 * S, allowed=True, remaining=3, reset_after_ns=3500 * MS, retry_after_ns=0),
          step("large_quantity_replenish", 20, 600, 60, 5, t=1 * S, allowed=False, remaining=3, reset_after_ns=3500 * MS,
               retry_after_ns=200 * MS)])

scenario("rapid_time_changes", f"{CORE}:658-694",
         [step("time_jump", 3, 10, 60, 1, allowed=True),
          step("time_jump", 3, 10, 60, 1, t=-5 * S, status=0),
          step("time_jump", 3, 10, 60, 1, t=10 * S, allowed=True)]
         + [step("time_jitter", 10, 10, 60, 1, t=(i if i % 2 == 0 else -i) * S, status=0) for i in range(5)])

# ---- store_test_suite.rs: GCRA through every store --------------------------
scenario("rate_limiting_all_stores(AdaptiveStore)", f"{SUITE}:542-598",
         [step("test_key", 5, 10, 3600, 1, allowed=True, remaining=5 - i - 1) for i in range(5)]
         + [step("test_key", 5, 10, 3600, 1, allowed=False),
            step("test_key", 5, 10, 3600, 1, t=360 * S, allowed=True, remaining=0)])

for _store in ("PeriodicStore", "ProbabilisticStore"):
    scenario(f"rate_limiting_all_stores({_store})", f"{SUITE}:542-598",
             [step("test_key", 5, 10, 3600, 1, allowed=True, remaining=5 - i - 1) for i in range(5)]
             + [step("test_key", 5, 10, 3600, 1, allowed=False),
                step("test_key", 5, 10, 3600, 1, t=360 * S, allowed=True, remaining=0)])

# ---- redis_test.rs (process_command -> actor -> rate_limit; secs-truncated) ---
# (10, 100, 60): ei = 0.6e9, dvt = 5.4e9.  q1 fresh: new_tat = t0: reset = dvt = 5.4e9 (5 s on the wire).
# q5 fresh: new_tat = t0 - ei + 5 ei = t0 + 2.4e9: reset = 2.4e9 + 5.4e9 = 7.8e9 (7 s on the wire).
scenario("redis_throttle_allowed", f"{REDIS}:117-129",
         [step("test_key", 10, 100, 60, 1, allowed=True, limit=10, remaining=9, reset_after_s=5, retry_after_s=0,
               reset_after_ns=5_400_000_000, retry_after_ns=0)])
scenario("redis_throttle_with_quantity", f"{REDIS}:132-144",
         [step("test_key2", 10, 100, 60, 5, allowed=True, limit=10, remaining=5, reset_after_s=7, retry_after_s=0,
               reset_after_ns=7_800_000_000, retry_after_ns=0)])
scenario("redis_throttle_exhaustion", f"{REDIS}:272-304",
         [step("exhaustion_test", 3, 100, 60, 1, allowed=True, limit=3, remaining=2),
          step("exhaustion_test", 3, 100, 60, 1, allowed=True, remaining=1),
          step("exhaustion_test", 3, 100, 60, 1, allowed=True, remaining=0),
          step("exhaustion_test", 3, 100, 60, 1, allowed=False, remaining=0)])
scenario("redis_multiple_keys", f"{REDIS}:307-330",
         [step(k, 5, 100, 60, 1, allowed=True, limit=5, remaining=4) for k in ("user:123", "user:456", "api:endpoint")]
         + [step(k, 5, 100, 60, 1, allowed=True, remaining=3) for k in ("user:123", "user:456", "api:endpoint")])
scenario("redis_different_limits_same_key", f"{REDIS}:333-381",
         [step("dynamic_limit_key", 10, 100, 60, 1, allowed=True, limit=10, remaining=9),
          step("dynamic_limit_key", 5, 100, 60, 1, allowed=True, limit=5, remaining_ge=0, remaining_le=5)])
scenario("redis_large_quantity", f"{REDIS}:384-395",
         [step("large_quantity_key", 10, 100, 60, 15, allowed=False, limit=10, remaining=10)])
scenario("redis_zero_quantity", f"{REDIS}:492-502",
         [step("zero_quantity_key", 10, 100, 60, 0, allowed=True, remaining=10)])
scenario("redis_boundary_values", f"{REDIS}:678-717",
         [step("boundary_key", I64_MAX, I64_MAX, I64_MAX, 1, allowed=True, limit=I64_MAX),
          step("tiny_key", 1, 1, 1, 1, allowed=True, limit=1, remaining=0)])

scenario("redis_special_characters_in_key", f"{REDIS}:398-420",
         [step(k, 5, 100, 60, 1, allowed=True, limit=5, remaining=4)
          for k in ("user:email@example.com", "api:v2/users/{id}", "rate:limit:user-123", "key with spaces",
                    "key:with:colons:everywhere", "UTF8:\u6d4b\u8bd5\u952e")])
scenario("redis_concurrent_same_key", f"{REDIS}:556-604",
         [step("concurrent_key", 10, 100, 60, 1, allowed=True, limit=10) for _ in range(5)], allowed_total=5)
scenario("redis_rapid_succession", f"{REDIS}:607-630",
         [step("rapid_key", 5, 100, 60, 1) for _ in range(10)], allowed_total=5)
scenario("redis_empty_key", f"{REDIS}:633-655",
         [step("", 10, 100, 60, 1, allowed=True, limit=10, remaining=9)])
scenario("redis_command_case_insensitive", f"{REDIS}:720-761",
         [step("case_test_key", 10, 100, 60, 1, allowed=True) for _ in range(4)])
scenario("redis_very_long_key", f"{REDIS}:764-812",
         [step("x" * 1000, 10, 100, 60, 1, allowed=True, limit=10, remaining=9),
          step("x" * 1000, 10, 100, 60, 1, allowed=True, remaining=8)])

# ---- more of core/tests.rs, hand-derived to the nanosecond -------------------------------------------------
# (5, 120, 60): ei = 0.5e9, dvt = 2e9.  Five requests at t0 leave tat = t0 + 2e9.  At t0 + m ms the request is allowed
# when new_tat - dvt = t0 + 0.5e9 <= now, i.e. from 500 ms on; cur = t0 + 2.5e9:
#   remaining = (now + dvt - cur) / ei = (m * 1e6 - 0.5e9) / 0.5e9, reset = (cur - now) + dvt = 4.5e9 - m * 1e6.
_rep = []
for millis, rem in [(500, 0), (1000, 1), (1500, 2), (2000, 3), (2500, 4)]:
    k = f"ns_replenish_{millis}"
    _rep += [step(k, 5, 120, 60, 1, reset_after_ns=(i + 4) * 500 * MS, retry_after_ns=0) for i in range(5)]
    _rep += [step(k, 5, 120, 60, 1, t=millis * MS, allowed=True, remaining=rem, reset_after_ns=4500 * MS - millis * MS,
                  retry_after_ns=0)]
# ... and 100 ms too early (t0 + 400 ms): denied, retry = 0.5e9 - 0.4e9, cur = t0 + 2e9: reset = 1.6e9 + 2e9, remaining 0
_rep += [step("ns_replenish_early", 5, 120, 60, 1) for _ in range(5)]
_rep += [step("ns_replenish_early", 5, 120, 60, 1, t=400 * MS, allowed=False, remaining=0, reset_after_ns=3600 * MS,
              retry_after_ns=100 * MS)]
scenario("gradual_replenishment_ns_exact", f"{CORE}:444-500 (values by hand from rate_limiter.rs:151-238)", _rep)

scenario("redis_mixed_commands", f"{REDIS}:422-473",
         [step("mixed_key", 10, 100, 60, 1, allowed=True), step("mixed_key", 10, 100, 60, 1, allowed=True, remaining=8)])

# ---- actor_tests.rs / grpc.rs -------------------------------------------------
scenario("actor_concurrent_requests", "throttlecrab-server/src/actor_tests.rs:34-70",
         [step("concurrent_test", 10, 10, 60, 1) for _ in range(20)], allowed_total=10)
scenario("actor_basic_rate_limiting", "throttlecrab-server/src/actor_tests.rs:8-31",
         [step("test", 5, 10, 60, 1, allowed=True, limit=5, remaining=4)])
# the client loop stops at the first denial: five allowed, the sixth denied with retry_after > 0 (seconds on the wire: 6)
scenario("grpc_rate_limiting", "throttlecrab-server/src/transport/grpc.rs:244-295",
         [step("rate_limit_test", 5, 10, 60, 1, allowed=True) for _ in range(5)]
         + [step("rate_limit_test", 5, 10, 60, 1, allowed=False, retry_after_s_gt=0)], allowed_total=5)
scenario("grpc_server_basic", "throttlecrab-server/src/transport/grpc.rs:203-242",
         [step("test_key", 10, 20, 60, 1, allowed=True, limit=10, remaining=9)])

# ---- Rate::from_count_and_period (rate/tests.rs:41-56) ------------------------
rates = {"source": "throttlecrab/src/core/rate/tests.rs:41-56",
         "cases": [{"count": 10, "period": 60, "period_ns": 6 * S},
                   {"count": 30, "period": 60, "period_ns": 2 * S}]}

# ---- Store contract (store_test_suite.rs) -------------------------------------
# ops: ["set_nx", key, val, ttl_ns, t, expect_bool] | ["get", key, t, expect_or_null]
#      | ["cas", key, old, new, ttl_ns, t, expect_bool]
TTL60 = 60 * S
I64_MIN = -2**63
store_contract = [
    {"name": "basic_operations", "source": f"{SUITE}:21-59", "ops": [
        ["set_nx", "key1", 100, TTL60, 0, True], ["get", "key1", 0, 100],
        ["set_nx", "key1", 200, TTL60, 0, False], ["get", "key1", 0, 100]]},
    {"name": "compare_and_swap", "source": f"{SUITE}:62-110", "ops": [
        ["set_nx", "key1", 100, TTL60, 0, True],
        ["cas", "key1", 100, 200, TTL60, 0, True], ["get", "key1", 0, 200],
        ["cas", "key1", 100, 300, TTL60, 0, False], ["get", "key1", 0, 200],
        ["cas", "key2", 0, 100, TTL60, 0, False]]},
    {"name": "ttl_expiration", "source": f"{SUITE}:113-170", "ops": [
        ["set_nx", "key1", 100, TTL60, 0, True], ["get", "key1", 0, 100],
        ["get", "key1", 59 * S, 100], ["get", "key1", 61 * S, None],
        ["cas", "key1", 100, 200, TTL60, 61 * S, False],
        ["set_nx", "key1", 300, TTL60, 61 * S, True], ["get", "key1", 61 * S, 300]]},
    {"name": "negative_tat", "source": f"{SUITE}:173-209", "ops": [
        ["set_nx", "key1", -1000, TTL60, 0, True], ["get", "key1", 0, -1000],
        ["cas", "key1", -1000, -500, TTL60, 0, True], ["get", "key1", 0, -500]]},
    {"name": "short_ttl", "source": f"{SUITE}:212-247", "ops": [
        ["set_nx", "key1", 100, 1 * MS, 0, True], ["get", "key1", 0, 100],
        ["get", "key1", 2 * MS, None]]},
    {"name": "extreme_values", "source": f"{SUITE}:250-286", "ops": [
        ["set_nx", "max", I64_MAX, TTL60, 0, True], ["get", "max", 0, I64_MAX],
        ["set_nx", "min", I64_MIN, TTL60, 0, True], ["get", "min", 0, I64_MIN],
        ["cas", "max", I64_MAX, I64_MAX - 1, TTL60, 0, True]]},
    {"name": "special_keys", "source": f"{SUITE}:289-338", "ops": [
        ["set_nx", "", 100, TTL60, 0, True], ["get", "", 0, 100],
        ["set_nx", "a" * 1000, 200, TTL60, 0, True], ["get", "a" * 1000, 0, 200],
        ["set_nx", "\U0001F980\U0001F525\U0001F4BB", 300, TTL60, 0, True],
        ["get", "\U0001F980\U0001F525\U0001F4BB", 0, 300],
        ["set_nx", "key:with:colons/and/slashes\\and\\backslashes", 400, TTL60, 0, True],
        ["get", "key:with:colons/and/slashes\\and\\backslashes", 0, 400]]},
    {"name": "concurrent_operations", "source": f"{SUITE}:342-375", "ops":
        [["set_nx", "counter", 0, TTL60, 0, True]]
        + [op for i in range(10) for op in (["get", "counter", 0, i], ["cas", "counter", i, i + 1, TTL60, 0, True])]
        + [["get", "counter", 0, 10]]},
    {"name": "cleanup_behavior", "source": f"{SUITE}:378-420", "ops":
        [["set_nx", f"key{i}", i, 1 * S, 0, True] for i in range(100)]
        + [["get", f"key{i}", 0, i] for i in range(100)]
        + [["get", f"key{i}", 2 * S, None] for i in range(10)]
        + [["get", f"key{i}", 2 * S, None] for i in range(100)]},
    {"name": "ttl_update_on_cas", "source": f"{SUITE}:423-461", "ops": [
        ["set_nx", "key1", 100, 10 * S, 0, True], ["cas", "key1", 100, 200, 100 * S, 0, True],
        ["get", "key1", 11 * S, 200], ["get", "key1", 101 * S, None]]},
    # cleanup_test.rs drives a PeriodicStore, whose cleanup fires from the first operation after its interval; the
    # assertions are about what a cleanup at that time leaves behind, which is the same for every store
    # (adaptive_cleanup.rs:173-203 == periodic.rs cleanup: retain(expiry > now))
    {"name": "cleanup_actually_happens", "source": "throttlecrab/src/core/store/cleanup_test.rs:8-41", "ops":
        [["set_nx", f"key_{i}", i, 1 * S, 0, True] for i in range(1000)]
        + [["len", 1000, 1000], ["set_nx", "trigger", 999, 60 * S, 61 * S, True], ["sweep", 61 * S], ["len", 1, 49],
           ["get", "trigger", 61 * S, 999]]},
    {"name": "cleanup_with_memory_pressure", "source": "throttlecrab/src/core/store/cleanup_test.rs:44-84", "ops":
        [["set_nx", f"key_{i}", i, (1 if i % 2 == 0 else 3600) * S, 0, True] for i in range(500)]
        + [["set_nx", "trigger", 999, 60 * S, 61 * S, True], ["sweep", 61 * S], ["len", 201, 299]]
        + [["get", f"key_{i}", 61 * S, i] for i in range(1, 100, 2)]},
    {"name": "no_cleanup_without_triggers", "source": "throttlecrab/src/core/store/cleanup_test.rs:87-107", "ops":
        [["set_nx", f"key_{i}", i, 3600 * S, 0, True] for i in range(100)]
        + [["get", f"key_{i}", 0, i] for i in range(10)]
        + [["len", 100, 100], ["sweep", 0], ["len", 100, 100]]},
    {"name": "memory_store_set_and_get", "source": "throttlecrab/src/core/store/tests.rs:5-28", "ops": [
        ["set_nx", "key1", 42, TTL60, 0, True], ["get", "key1", 0, 42], ["set_nx", "key1", 100, TTL60, 0, False], ["get", "key1", 0, 42]]},
    {"name": "memory_store_compare_and_swap", "source": "throttlecrab/src/core/store/tests.rs:31-57", "ops": [
        ["set_nx", "key1", 10, TTL60, 0, True], ["cas", "key1", 10, 20, TTL60, 0, True], ["get", "key1", 0, 20],
        ["cas", "key1", 10, 30, TTL60, 0, False], ["get", "key1", 0, 20]]},
    {"name": "memory_store_ttl", "source": "throttlecrab/src/core/store/tests.rs:60-84", "ops": [
        ["set_nx", "key1", 42, 100 * MS, 0, True], ["get", "key1", 0, 42], ["set_nx", "key2", 100, TTL60, 200 * MS, True],
        ["get", "key1", 200 * MS, None]]},
    {"name": "memory_store_get_nonexistent", "source": "throttlecrab/src/core/store/tests.rs:87-93", "ops": [
        ["get", "nonexistent", 0, None]]},
    {"name": "memory_store_multiple_keys", "source": "throttlecrab/src/core/store/tests.rs:96-114", "ops":
        [["set_nx", f"key{i}", i * 10, TTL60, 0, True] for i in range(10)] + [["get", f"key{i}", 0, i * 10] for i in range(10)]},
    {"name": "zero_ttl", "source": f"{SUITE}:464-487", "ops": [
        ["set_nx", "key1", 100, 0, 0, True], ["get", "key1", 1, None]]},
    {"name": "many_keys", "source": f"{SUITE}:490-539", "ops":
        [["set_nx", f"key_{i}", i, 3600 * S, 0, True] for i in range(500)]
        + [["get", f"key_{i}", 0, i] for i in range(500)]
        + [["cas", f"key_{i}", i, i + 1000, 3600 * S, 0, True] for i in range(0, 500, 7)]
        + [["get", f"key_{i}", 0, i + 1000] for i in range(0, 500, 7)]},
]

out = {"t0_ns": T0, "scenarios": scenarios, "rates": rates, "store_contract": store_contract}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kat.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(f"wrote {path}: {len(scenarios)} scenarios, {sum(len(s['steps']) for s in scenarios)} steps")
