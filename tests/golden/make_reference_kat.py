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
Scenario-level:  "allowed_total": N  -> exactly N of the steps are allowed.
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

scenario("burst_capacity", f"{CORE}:17-33",
         [step("burst_test", 5, 10, 60, 1, allowed=True, remaining=5 - (i + 1)) for i in range(5)]
         + [step("burst_test", 5, 10, 60, 1, allowed=False, remaining=0, retry_after_s_gt=0)])

scenario("rate_replenishment", f"{CORE}:36-62",
         [step("replenish_test", 2, 60, 60, 1, allowed=True),
          step("replenish_test", 2, 60, 60, 1, allowed=True),
          step("replenish_test", 2, 60, 60, 1, allowed=False),
          step("replenish_test", 2, 60, 60, 1, t=1 * S, allowed=True)])

scenario("different_keys", f"{CORE}:65-91",
         [step("key1", 2, 2, 60, 1, allowed=True),
          step("key2", 2, 2, 60, 1, allowed=True),
          step("key1", 2, 2, 60, 1, allowed=True),
          step("key1", 2, 2, 60, 1, allowed=False),
          step("key2", 2, 2, 60, 1, allowed=True),
          step("key2", 2, 2, 60, 1, allowed=False)])

scenario("quantity_parameter", f"{CORE}:94-118",
         [step("quantity_test", 10, 10, 60, 5, allowed=True, remaining=5),
          step("quantity_test", 10, 10, 60, 6, allowed=False, remaining=5),
          step("quantity_test", 10, 10, 60, 5, allowed=True, remaining=0)])

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

scenario("edge_cases_zero_remaining", f"{CORE}:350-412",
         [step("exact_timing", 2, 120, 60, 1, allowed=True, remaining=1),
          step("exact_timing", 2, 120, 60, 1, allowed=True, remaining=0),
          step("exact_timing", 2, 120, 60, 1, t=500 * MS, allowed=True, remaining=0),
          step("zero_period", 10, 10, 0, 1, status="err"),
          step("fractional", 3, 7, 60, 1, allowed=True, remaining=2),
          step("fractional", 3, 7, 60, 1),
          step("fractional", 3, 7, 60, 1),
          step("fractional", 3, 7, 60, 1, t=8 * S, allowed=False),
          step("fractional", 3, 7, 60, 1, t=9 * S, allowed=True, remaining=0),
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

_cx = [step("partial_burst", 8, 240, 60, 6, allowed=True, remaining=2),
       step("partial_burst", 8, 240, 60, 1, t=500 * MS, allowed=True, remaining=3),
       step("partial_burst", 8, 240, 60, 1, t=1500 * MS, allowed=True, remaining=6)]
_cx += [step("slow_replenish", 3, 6, 60, 1) for _ in range(3)]
_cx += [step("slow_replenish", 3, 6, 60, 1, t=5 * S, allowed=False),
        step("slow_replenish", 3, 6, 60, 1, t=10 * S, allowed=True, remaining=0),
        step("slow_replenish", 3, 6, 60, 1, t=20 * S, allowed=True, remaining=0)]
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
          step("large_quantity_replenish", 20, 600, 60, 15, allowed=True, remaining=5),
          step("large_quantity_replenish", 20, 600, 60, 12, t=1 * S, allowed=True, remaining=3),
          step("large_quantity_replenish", 20, 600, 60, 5, t=1 * S, allowed=False, remaining=3)])

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

# ---- redis_test.rs (process_command -> actor -> rate_limit; secs-truncated) ---
scenario("redis_throttle_allowed", f"{REDIS}:117-129",
         [step("test_key", 10, 100, 60, 1, allowed=True, limit=10, remaining=9, reset_after_s=5, retry_after_s=0)])
scenario("redis_throttle_with_quantity", f"{REDIS}:132-144",
         [step("test_key2", 10, 100, 60, 5, allowed=True, limit=10, remaining=5, reset_after_s=7, retry_after_s=0)])
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

# ---- actor_tests.rs / grpc.rs -------------------------------------------------
scenario("actor_concurrent_requests", "throttlecrab-server/src/actor_tests.rs:34-70",
         [step("concurrent_test", 10, 10, 60, 1) for _ in range(20)], allowed_total=10)
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
