// throttlecrab_sweep.hpp -- WHEN to call tc_sweep_expired: the cleanup cadences of the reference's three
// stores as host-side schedulers (C++17, header only).  SURVEY.md section 8(f) row 4.
//
// The reference's stores differ only in when they run `retain(expiry > now)`; the decisions are the same
// (DESIGN.md section 2).  The engine never cleans implicitly, so the policy lives with the caller:
//   AdaptiveSweep       throttlecrab/src/core/store/adaptive_cleanup.rs:138-211 (should_clean / cleanup)
//   PeriodicSweep       throttlecrab/src/core/store/periodic.rs:128-142
//   ProbabilisticSweep  throttlecrab/src/core/store/probabilistic.rs:110-125
// All three work on batches: `on_batch(n, now)` says whether a sweep is due after n more operations
// (the reference counts one operation per mutating store call: one per allowed request), and
// `swept(removed, entries_before, now)` feeds the result of the sweep back (AdaptiveSweep adapts its
// interval with it).  Times are ns since the epoch, like everywhere at the C ABI.
#pragma once

#include <cstdint>
#include <numeric>

namespace throttlecrab {
namespace sweep {

constexpr int64_t NS = 1000000000LL;

// periodic.rs:128-142: `if now >= next_cleanup { retain; next_cleanup = now + interval }`
class PeriodicSweep {
  public:
    // PeriodicStore::with_capacity: interval 60 s (DEFAULT_CLEANUP_INTERVAL_SECS, periodic.rs:12), first cleanup one
    // interval after creation (periodic.rs:107)
    explicit PeriodicSweep(int64_t created_ns, int64_t interval_ns = 60 * NS) : interval_(interval_ns), next_(created_ns + interval_ns) {}
    bool on_batch(uint64_t /*n_ops*/, int64_t now_ns) const { return now_ns >= next_; }
    void swept(uint64_t /*removed*/, uint64_t /*entries_before*/, int64_t now_ns) { next_ = now_ns + interval_; }
    int64_t next_cleanup_ns() const { return next_; }

  private:
    int64_t interval_, next_;
};

// probabilistic.rs:110-125: the k-th operation cleans iff (k * 2654435761) % N == 0 (N = 1000 by default:
// the "probability" is deterministic).  While k * 2654435761 < 2^64 that is k % (N / gcd(N, 2654435761)) == 0.
class ProbabilisticSweep {
  public:
    explicit ProbabilisticSweep(uint64_t cleanup_probability = 1000)
        : n_(cleanup_probability ? cleanup_probability : 1), period_(n_ / std::gcd(n_, (uint64_t)2654435761ULL)) {}
    // does any of the next n_ops operations trigger a cleanup?
    bool on_batch(uint64_t n_ops, int64_t /*now_ns*/) {
        const uint64_t a = ops_;
        ops_ += n_ops;
        if (ops_ < (1ULL << 32)) return ops_ / period_ > a / period_;
        for (uint64_t k = a + 1; k <= ops_; ++k) // (beyond 2^32 operations the product wraps: test it as the reference does)
            if ((k * 2654435761ULL) % n_ == 0) return true;
        return false;
    }
    void swept(uint64_t, uint64_t, int64_t) {}
    uint64_t operations() const { return ops_; }

  private:
    uint64_t n_, period_, ops_ = 0;
};

// adaptive_cleanup.rs:138-211.  The expired-ratio trigger (:150-163) counts operations that ran into an
// expired entry; the device does not report those one by one, so `expired_seen` is whatever the caller
// knows (0 if nothing: the time, operation-count and occupancy triggers remain).
class AdaptiveSweep {
  public:
    // AdaptiveStore::with_capacity / builder defaults: min 1 s, max 300 s, start 5 s, 100 000 operations
    // (the server's default is 1 000 000: throttlecrab-server/src/config.rs:301)
    explicit AdaptiveSweep(int64_t created_ns, uint64_t capacity, int64_t min_interval_ns = 1 * NS, int64_t max_interval_ns = 300 * NS,
                           uint64_t max_operations = 100000)
        : min_(min_interval_ns), max_(max_interval_ns), cur_(5 * NS), next_(created_ns + 5 * NS), max_ops_(max_operations),
          // HashMap::with_capacity(capacity * 1.3) -- the memory-pressure trigger compares with 3/4 of the map's capacity
          map_capacity_((uint64_t)((double)capacity * 1.3)) {}

    // should_clean after n_ops more operations, with `entries` live entries in the store
    bool on_batch(uint64_t n_ops, int64_t now_ns, uint64_t entries = 0, uint64_t expired_seen = 0) {
        ops_ += n_ops;
        expired_ += expired_seen;
        if (now_ns >= next_) return true;                      // :140
        if (ops_ >= max_ops_) return true;                     // :145
        if (expired_ > 50) {                                   // :150-163
            const double ratio = (double)expired_ / (double)(entries ? entries : 1);
            const double threshold = last_removed_ > last_total_ / 4 ? 0.2 / 2.0 : 0.2 * 1.25;
            if (ratio > threshold) return true;
        }
        if (entries > map_capacity_ * 3 / 4) return true;      // :166
        return false;
    }
    // cleanup()'s bookkeeping (:186-202)
    void swept(uint64_t removed, uint64_t entries_before, int64_t now_ns) {
        if (removed == 0 && expired_ == 0) cur_ = cur_ * 2 < max_ ? cur_ * 2 : max_;
        else if ((double)removed > (double)entries_before * 0.5) cur_ = cur_ / 2 > min_ ? cur_ / 2 : min_;
        last_removed_ = removed;
        last_total_ = entries_before;
        next_ = now_ns + cur_;
        expired_ = 0;
        ops_ = 0;
    }
    int64_t current_interval_ns() const { return cur_; }
    int64_t next_cleanup_ns() const { return next_; }

  private:
    int64_t min_, max_, cur_, next_;
    uint64_t max_ops_, map_capacity_;
    uint64_t ops_ = 0, expired_ = 0, last_removed_ = 0, last_total_ = 0;
};

} // namespace sweep
} // namespace throttlecrab
