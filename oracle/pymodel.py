"""A second, independent restatement of the reference path in pure Python (test infrastructure,
like everything under oracle/): written from throttlecrab/src/core/rate_limiter.rs:102-250,
core/rate/mod.rs:164-176 and core/store/adaptive_cleanup.rs:220-279 with Python's unbounded
integers and the Rust operations spelled out (saturating_*, `as` casts, Duration arithmetic),
so that the C oracle (gcra_oracle.c) can be cross-checked against something that shares none of
its code.  Small cases only: it is a plain per-request loop over a dict."""
from __future__ import annotations

I64_MIN, I64_MAX, U64_MAX = -(1 << 63), (1 << 63) - 1, (1 << 64) - 1
OK, NEGATIVE_QUANTITY, INVALID_RATE_LIMIT, INTERNAL = 0, 1, 2, 3
NS = 1_000_000_000


def sat(x: int) -> int:
    return I64_MIN if x < I64_MIN else I64_MAX if x > I64_MAX else x


def as_i64(x: int) -> int:          # `as i64` of a u64 / u128: keep the low 64 bits, reinterpret
    x &= U64_MAX
    return x - (1 << 64) if x >= (1 << 63) else x


def as_u64(x: int) -> int:          # `as u64` of an i64
    return x & U64_MAX


def f64_as_u64(x: float) -> int:    # Rust float -> int casts saturate, NaN -> 0
    if x != x or x <= 0.0:
        return 0
    if x >= 18446744073709551616.0:
        return U64_MAX
    return int(x)


class PyModel:
    """RateLimiter<AdaptiveStore> with cleanup never firing on its own (decision-neutral)."""

    def __init__(self):
        self.data = {}  # key -> (tat i64, expiry in ns since the epoch as an unbounded int)

    # adaptive_cleanup.rs:246-252
    def get(self, key, now):
        e = self.data.get(key)
        return e[0] if e is not None and e[1] > now else None

    # adaptive_cleanup.rs:221-244
    def compare_and_swap_with_ttl(self, key, old, new, ttl, now):
        e = self.data.get(key)
        if e is None or e[1] <= now or e[0] != old:
            return False
        self.data[key] = (new, now + ttl)
        return True

    # adaptive_cleanup.rs:254-278
    def set_if_not_exists_with_ttl(self, key, value, ttl, now):
        e = self.data.get(key)
        if e is not None and e[1] > now:
            return False
        self.data[key] = (value, now + ttl)
        return True

    def cleanup(self, now):           # adaptive_cleanup.rs:173-203: retain(expiry > now)
        dead = [k for k, e in self.data.items() if not e[1] > now]
        for k in dead:
            del self.data[k]
        return len(dead)

    # rate_limiter.rs:102-250 -> (status, allowed, limit, remaining, reset_after_ns, retry_after_ns)
    def rate_limit(self, key, max_burst, count_per_period, period, quantity, now_ns):
        if quantity < 0:
            return (NEGATIVE_QUANTITY, False, 0, 0, 0, 0)
        if max_burst <= 0 or count_per_period <= 0 or period <= 0:
            return (INVALID_RATE_LIMIT, False, 0, 0, 0, 0)
        ei_u64 = f64_as_u64(float(period) * 1_000_000_000.0 / float(count_per_period))   # rate/mod.rs:172
        mult = (max_burst - 1) & 0xFFFFFFFF                                                # `as u32`
        # Duration * u32 (checked_mul, panics on overflow): secs*rhs + carry of the nanos must fit in u64
        secs, nanos = divmod(ei_u64, NS)
        if secs * mult + (nanos * mult) // NS > U64_MAX:
            return (INTERNAL, False, 0, 0, 0, 0)     # the reference panics here: outside the validated domain
        ei = as_i64(ei_u64)                          # as_nanos() as i64
        dvt = as_i64(ei_u64 * mult)
        if now_ns < 0:
            return (INTERNAL, False, 0, 0, 0, 0)     # the reference falls back to the wall clock (:126-144)
        if now_ns + dvt > I64_MAX:
            return (INTERNAL, False, 0, 0, 0, 0)     # plain `+` at :217 overflows: debug panic / release wrap
        stored = self.get(key, now_ns)
        if stored is not None:
            tat = max(stored, sat(now_ns - dvt))
        else:
            tat = sat(now_ns - ei)
        increment = sat(ei * quantity)
        new_tat = sat(tat + increment)
        allow_at = sat(new_tat - dvt)
        allowed = now_ns >= allow_at
        if allowed:
            ttl = as_u64(sat(sat(new_tat - now_ns) + dvt))
            if stored is not None:
                assert self.compare_and_swap_with_ttl(key, stored, new_tat, ttl, now_ns)
            else:
                assert self.set_if_not_exists_with_ttl(key, new_tat, ttl, now_ns)
        cur = new_tat if allowed else tat
        room = sat((now_ns + dvt) - cur)
        remaining = max(_trunc_div(room, ei), 0) if ei > 0 else 0
        reset_after = as_u64(max(sat(sat(cur - now_ns) + dvt), 0))
        retry_after = 0 if allowed else as_u64(max(sat(allow_at - now_ns), 0))
        return (OK, allowed, max_burst, remaining, reset_after, retry_after)


def _trunc_div(a: int, b: int) -> int:   # Rust `/` on i64 truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q
