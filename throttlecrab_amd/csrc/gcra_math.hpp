// gcra_math.hpp -- the GCRA arithmetic of throttlecrab's hot path as inlinable
// host/device functions.  Written for gfx950 (no other GPU target).
//
// Follows, line by line:
//   throttlecrab/src/core/rate_limiter.rs:111-123   validation, emission interval, dvt
//   throttlecrab/src/core/rate/mod.rs:164-176       Rate::from_count_and_period
//   throttlecrab/src/core/rate_limiter.rs:151-238   TAT select, allow test, TTL, results
//   throttlecrab/src/core/store/adaptive_cleanup.rs:220-279  liveness = expiry > now
// All i64 arithmetic is Rust `saturating_*` except `now + dvt` (:217), whose
// overflow is excluded up front (TC_INTERNAL, see include/tcgpu.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define TC_HD __host__ __device__ __forceinline__

namespace tc {

enum : int { ST_OK = 0, ST_NEGATIVE_QUANTITY = 1, ST_INVALID_RATE_LIMIT = 2, ST_INTERNAL = 3 };

TC_HD int64_t sat_add(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_add_overflow(a, b, &r)) return b > 0 ? INT64_MAX : INT64_MIN;
    return r;
}
TC_HD int64_t sat_sub(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_sub_overflow(a, b, &r)) return b < 0 ? INT64_MAX : INT64_MIN;
    return r;
}
TC_HD int64_t sat_mul(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_mul_overflow(a, b, &r)) return ((a < 0) != (b < 0)) ? INT64_MIN : INT64_MAX;
    return r;
}
TC_HD int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

// Rust `f64 as u64`: saturating, NaN -> 0.
TC_HD uint64_t f64_to_u64_sat(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}

// rate/mod.rs:164-176: (period as f64 * 1e9 / count as f64) as u64.  Two
// correctly rounded IEEE-754 operations (built with -ffp-contract=off; there
// is no multiply-add to contract here anyway).
TC_HD uint64_t emission_interval(int64_t count, int64_t period) {
    double p = (double)period * 1000000000.0;
    double pn = p / (double)count;
    return f64_to_u64_sat(pn);
}

// One slot of resident state: stored TAT + expiry (ns since epoch, saturated
// to u64).  expiry == 0 <=> vacant.  16 bytes so that one request touches one
// 16-byte granule (a single global_load_dwordx4 / global_store_dwordx4).
struct __attribute__((aligned(16))) Cell {
    int64_t tat;
    uint64_t expiry;
};

// Per-slot rate parameters (registered mode): emission interval and delay
// variation tolerance in ns, derived once at registration.
struct __attribute__((aligned(16))) Rate {
    int64_t ei;
    int64_t dvt;
};

// A cell is always read and written as ONE 16-byte access (global_load/store_dwordx4): the
// evaluation kernels let a key's cell be rewritten while later requests of the same key may still
// read it (k_eval_sorted, direct mode), which is only sound if nobody can see half a cell.
__device__ __forceinline__ Cell load_cell(const Cell* p) {
    const longlong2 v = *reinterpret_cast<const longlong2*>(p);
    Cell c;
    c.tat = v.x;
    c.expiry = (uint64_t)v.y;
    return c;
}
__device__ __forceinline__ void store_cell(Cell* p, const Cell& c) {
    *reinterpret_cast<longlong2*>(p) = make_longlong2(c.tat, (long long)c.expiry);
}

// TC_CFG_FIXED_PARAMS layout: 8 bytes per key, the stored TAT alone ("fixed-params
// shortcut").  While a key's (burst, count, period) never change, every write leaves
// expiry == tat + dvt (rate_limiter.rs:179-183 with adaptive_cleanup.rs:237: now + ((new_tat - now) + dvt)),
// so the expiry column is redundant; a key that holds nothing is TAT_VACANT.  The engine only accepts
// plans for this layout under which that identity is exact (fixed_plan_ok) and timestamps below 2^62.
constexpr int64_t TAT_VACANT = INT64_MIN;
TC_HD Cell fixed_cell(int64_t tat, int64_t dvt) {
    Cell c;
    c.tat = tat == TAT_VACANT ? 0 : tat;
    c.expiry = tat == TAT_VACANT ? 0ull : (uint64_t)(tat + dvt);
    return c;
}
// ei > 0 keeps TATs moving, dvt >= ei (burst >= 2 after the u32 truncation) keeps every ttl >= 0 (quantity 0
// on a fresh key included), the 2^60 bounds keep tat + dvt inside i64 for every timestamp below 2^62.
TC_HD bool fixed_plan_ok(int64_t ei, int64_t dvt) {
    const int64_t LIM = (int64_t)1 << 60;
    return ei > 0 && dvt >= ei && ei < LIM && dvt < LIM;
}

// A registered rate plan: everything RateLimiter::rate_limit derives from
// (max_burst, count_per_period, period) before it looks at the key
// (rate_limiter.rs:119-123): emission interval, delay variation tolerance, and
// the burst capacity that becomes `limit`.  Plans are dictionary-coded: a key
// carries a 16-bit class id (column rate_id[], 2 B/key), the classes sit in one
// small table that stays in L2.  The resident state proper is the 16-byte Cell:
// 10 M keys = 160 MB, inside the 256 MiB Infinity Cache, and one decision touches
// one 16-byte granule.  (Measured, tools/microbench.hip, 1 Mi sorted gathers with
// write-back over 10 M keys, fresh slots every launch: 32-byte records 48.5 us,
// 16-byte cells 35.2 us.)
struct __attribute__((aligned(32))) RateClass {
    int64_t ei;
    int64_t dvt;
    int64_t burst;
    int64_t pad;
};

// rate_limiter.rs:119-123,154-155: ei and dvt from (burst, count, period).
// Returns ST_OK / ST_INVALID_RATE_LIMIT / ST_INTERNAL (Duration*u32 overflow,
// where the reference panics).
TC_HD int derive_rate(int64_t burst, int64_t count, int64_t period, int64_t& ei, int64_t& dvt) {
    ei = 0;
    dvt = 0;
    if (burst <= 0 || count <= 0 || period <= 0) return ST_INVALID_RATE_LIMIT;
    const uint64_t ei_u = emission_interval(count, period);
    const uint64_t mult = (uint64_t)(uint32_t)(uint64_t)(burst - 1); // `(max_burst - 1) as u32`
    // Duration::checked_mul(u32): secs*rhs + (nanos*rhs)/1e9 must fit in u64
    const uint64_t secs = ei_u / 1000000000ull, nanos = ei_u % 1000000000ull;
    const uint64_t extra = (nanos * mult) / 1000000000ull;
    uint64_t s;
    if (__builtin_mul_overflow(secs, mult, &s) || __builtin_add_overflow(s, extra, &s)) return ST_INTERNAL;
    ei = (int64_t)ei_u;            // as_nanos() as i64 (truncating)
    dvt = (int64_t)(ei_u * mult);  // low 64 bits of the u128 product, as i64
    return ST_OK;
}

// Request-dependent part of the validated domain.
TC_HD int check_request(int64_t quantity, int64_t now, int64_t dvt) {
    if (quantity < 0) return ST_NEGATIVE_QUANTITY;
    int64_t t;
    if (now < 0 || __builtin_add_overflow(now, dvt, &t)) return ST_INTERNAL;
    return ST_OK;
}

// Full per-request status in the reference's precedence order
// (rate_limiter.rs:111-117: quantity first, then the rate triple).
TC_HD int derive_request(int64_t burst, int64_t count, int64_t period, int64_t quantity, int64_t now,
                         int64_t& ei, int64_t& dvt) {
    ei = 0;
    dvt = 0;
    if (quantity < 0) return ST_NEGATIVE_QUANTITY;
    int st = derive_rate(burst, count, period, ei, dvt);
    if (st != ST_OK) return st;
    return check_request(quantity, now, dvt);
}

struct Decision {
    int64_t remaining;
    int64_t reset_after;
    int64_t retry_after;
    bool allowed;
};

// One request against one cell (rate_limiter.rs:151-238 with the store calls
// of adaptive_cleanup.rs:221-279 inlined: get -> live iff expiry > now; the
// CAS / set-if-not-exists cannot fail for a single owner).  Advances `c` when
// the request is allowed.  WANT_RESULT=false skips the i64 division.
template <bool WANT_RESULT>
TC_HD Decision gcra_step(Cell& c, int64_t ei, int64_t dvt, int64_t q, int64_t now) {
    const bool live = c.expiry > (uint64_t)now;                       // adaptive_cleanup.rs:248
    const int64_t tat = live ? max64(c.tat, sat_sub(now, dvt))        // :158-161
                             : sat_sub(now, ei);                      // :162-166
    const int64_t increment = sat_mul(ei, q);                         // :170
    const int64_t new_tat = sat_add(tat, increment);                  // :171
    const int64_t allow_at = sat_sub(new_tat, dvt);                   // :174
    Decision d;
    d.allowed = now >= allow_at;                                      // :175
    if (d.allowed) {
        const uint64_t ttl = (uint64_t)sat_add(sat_sub(new_tat, now), dvt); // :179-183 `as u64`
        uint64_t e = (uint64_t)now + ttl;                             // now + ttl (adaptive_cleanup.rs:237)
        if (e < ttl) e = UINT64_MAX;                                  // beyond u64 ns == never (now < 2^63)
        c.tat = new_tat;
        c.expiry = e;
    }
    if (WANT_RESULT) {
        const int64_t cur = d.allowed ? new_tat : tat;                // :208
        const int64_t room = sat_sub(now + dvt, cur);                 // :217-218
        d.remaining = ei > 0 ? max64(room / ei, 0) : 0;               // :221-225
        d.reset_after = max64(sat_add(sat_sub(cur, now), dvt), 0);    // :227-232
        d.retry_after = d.allowed ? 0 : max64(sat_sub(allow_at, now), 0); // :234-238
    } else {
        d.remaining = d.reset_after = d.retry_after = 0;
    }
    return d;
}

// Closed form for a run of identical requests (same ei, dvt, q, now) against
// one cell: after the first request was allowed and left the cell at `new0`,
// request number r >= 1 of the run sees the stored TAT
//     new0 + (min(r, n_tot) - 1) * inc,     n_tot = 1 + (now + dvt - new0) / inc
// provided nothing saturates and the clamp of :158-161 is a no-op.  `regular`
// says whether those provisos hold; otherwise the run is walked one by one.
struct RunForm {
    int64_t new0;
    int64_t inc;
    int64_t n_tot; // requests of the run that are allowed (if the run is long enough)
    bool regular;
};
TC_HD RunForm run_form(const Cell& after0, int64_t ei, int64_t dvt, int64_t q, int64_t now) {
    RunForm f;
    f.new0 = after0.tat;
    f.regular = false;
    f.n_tot = 1;
    int64_t inc;
    const int64_t LIM = (int64_t)1 << 62;
    if (ei <= 0 || dvt < 0 || q <= 0 || __builtin_mul_overflow(ei, q, &inc)) {
        f.inc = 0;
        return f;
    }
    f.inc = inc;
    // now >= 0, dvt >= 0, now + dvt does not overflow (check_request)
    const int64_t lim = now + dvt;
    if (inc >= LIM || lim >= LIM || f.new0 <= -LIM || f.new0 > lim) return f;
    if (!(after0.expiry > (uint64_t)now)) return f; // entry must stay live (ttl != 0)
    if (f.new0 < now - dvt) return f;               // clamp :160 must be a no-op
    f.n_tot = 1 + (lim - f.new0) / inc;
    f.regular = true;
    return f;
}

// The same run without the 64-bit division, for decisions-only batches: in a regular run the request of
// rank r >= 1 is allowed  <=>  r < n_tot = 1 + (lim - new0) / inc  <=>  r * inc <= lim - new0.
struct RunLite {
    int64_t new0, inc;
    uint64_t room; // lim - new0 >= 0
    bool regular;
};
TC_HD RunLite run_lite(const Cell& after0, int64_t ei, int64_t dvt, int64_t q, int64_t now) {
    RunLite f;
    f.new0 = after0.tat;
    f.inc = 0;
    f.room = 0;
    f.regular = false;
    int64_t inc;
    const int64_t LIM = (int64_t)1 << 62;
    if (ei <= 0 || dvt < 0 || q <= 0 || __builtin_mul_overflow(ei, q, &inc)) return f;
    f.inc = inc;
    const int64_t lim = now + dvt;
    if (inc >= LIM || lim >= LIM || f.new0 <= -LIM || f.new0 > lim) return f;
    if (!(after0.expiry > (uint64_t)now)) return f;
    if (f.new0 < now - dvt) return f;
    f.room = (uint64_t)(lim - f.new0);
    f.regular = true;
    return f;
}
// r * inc <= room, exactly (r < 2^32, inc < 2^62: the product needs up to 94 bits)
TC_HD bool rank_allowed(const RunLite& f, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t hi = __umul64hi((uint64_t)r, (uint64_t)f.inc);
#else
    const uint64_t hi = (uint64_t)(((unsigned __int128)r * (unsigned __int128)(uint64_t)f.inc) >> 64);
#endif
    return hi == 0 && (uint64_t)r * (uint64_t)f.inc <= f.room;
}
// the cell an allowed request leaves: new_tat with the expiry of rate_limiter.rs:179-183 / adaptive_cleanup.rs:237
TC_HD Cell cell_after(int64_t new_tat, int64_t dvt, int64_t now) {
    Cell c;
    const uint64_t ttl = (uint64_t)sat_add(sat_sub(new_tat, now), dvt);
    uint64_t e = (uint64_t)now + ttl;
    if (e < ttl) e = UINT64_MAX;
    c.tat = new_tat;
    c.expiry = e;
    return c;
}

// Watchdog for the engine's wait loops (radix look-back, k_eval_sorted's direct stores, k_eval_general's chain).
// Each of them waits only for blocks that were dispatched earlier, which holds on today's in-order
// dispatcher but is not something HIP promises; should a wait ever outlast SPIN_LIMIT_TICKS of the 100 MHz wall
// clock (2 s: four orders of magnitude above the longest legitimate wait) the loop gives up and the engine's
// invariant counter (tc_selfcheck) is raised, so a broken assumption shows as a failed check instead of a hung GPU.
#if defined(__HIPCC__)
// A tripped watchdog (or any other broken invariant) POISONS the engine: besides the device-side counter that
// tc_selfcheck reads, the kernel raises a word in pinned host memory whose address sits POISON_PTR_WORDS words behind the
// counter.  Every ABI call looks at that word: from then on the engine answers TC_E_INVARIANT instead of handing out
// results computed from a state somebody stopped waiting for.
constexpr int POISON_PTR_WORDS = 4;
__device__ __forceinline__ void invariant_failed(unsigned long long* violations) {
    if (violations == nullptr) return;
    atomicAdd(violations, 1ull);
    uint32_t* host = reinterpret_cast<uint32_t*>(__hip_atomic_load(violations + POISON_PTR_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (host != nullptr) __hip_atomic_store(host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr long long SPIN_LIMIT_TICKS = 200000000ll;
struct SpinGuard {
    long long t0 = 0;
    uint32_t polls = 0;
};
__device__ __forceinline__ bool spin_expired(SpinGuard& g) {
    if ((++g.polls & 255u) != 0u) return false;
    const long long now = wall_clock64();
    if (g.t0 == 0) {
        g.t0 = now;
        return false;
    }
    return now - g.t0 > SPIN_LIMIT_TICKS;
}
#endif

} // namespace tc
