// throttlecrab_sweep.hpp -- WHEN to call tc_sweep_expired: the cleanup cadences of the reference's three
// stores as host-side schedulers (C++17, header only).  SURVEY.md section 8(f) row 4.
//
// The reference's stores differ only in when they run `retain(expiry > now)`; the decisions are the same
// (DESIGN.md section 2).  These classes are what libtcgpu.so itself runs once tc_set_sweep_policy has named one (round 5:
// csrc/autosweep.hip includes this header -- the engine then cleans in front of its own mutating calls, like the
// reference's stores do), and what a caller that keeps the engine policy-free can drive by hand:
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
    int64_t interval_ns() const { return interval_; }

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
          // HashMap::with_capacity(capacity * 1.3) -- the memory-pressure trigger compares with 3/4 of the map's capacity(),
          // which is what hashbrown rounds the request up to
          map_capacity_(hashbrown_capacity((uint64_t)((double)capacity * 1.3))) {}

    // std::collections::HashMap / ahash::AHashMap (hashbrown): buckets = the power of two that holds the request at a load
    // of 7/8, capacity() = 7/8 of the buckets (tiny tables: 4 buckets hold 3, 8 hold 7)
    static uint64_t hashbrown_capacity(uint64_t want) {
        if (want == 0) return 0;
        if (want < 4) return 3;
        if (want < 8) return 7;
        uint64_t buckets = 8;
        while (buckets < want * 8 / 7 + (want * 8 % 7 ? 1 : 0)) buckets *= 2;
        return buckets / 8 * 7;
    }
    // the map grows (doubles) when an insert finds it full; the engine's table does not, its owner tells the policy how many
    // entries there are and the virtual map follows -- otherwise a store that is legitimately more than 3/4 full of LIVE keys
    // would be swept in front of every call (the reference's would have grown out of the trigger)
    void grow_map(uint64_t entries) {
        if (map_capacity_ == 0) map_capacity_ = 3;
        while (entries > map_capacity_) map_capacity_ = hashbrown_capacity(map_capacity_ * 2);
    }
    uint64_t map_capacity() const { return map_capacity_; }
    // an owner whose table IS the map (the engine's default): the size trigger then fires at 3/4 of exactly this
    void set_map_capacity(uint64_t entries) { map_capacity_ = entries; }

    enum Why { NOT_DUE = 0, BY_TIME = 1, BY_OPERATIONS = 2, BY_EXPIRED_RATIO = 3, BY_SIZE = 4 };
    // should_clean after n_ops more operations, with `entries` live entries in the store
    bool on_batch(uint64_t n_ops, int64_t now_ns, uint64_t entries = 0, uint64_t expired_seen = 0) {
        ops_ += n_ops;
        expired_ += expired_seen;
        why_ = NOT_DUE;
        if (now_ns >= next_) why_ = BY_TIME;                   // :140
        else if (ops_ >= max_ops_) why_ = BY_OPERATIONS;       // :145
        else if (expired_ > 50) {                              // :150-163
            const double ratio = (double)expired_ / (double)(entries ? entries : 1);
            const double threshold = last_removed_ > last_total_ / 4 ? 0.2 / 2.0 : 0.2 * 1.25;
            if (ratio > threshold) why_ = BY_EXPIRED_RATIO;
        }
        if (why_ == NOT_DUE && entries > map_capacity_ * 3 / 4) why_ = BY_SIZE; // :166
        return why_ != NOT_DUE;
    }
    Why why() const { return why_; } // which trigger the last on_batch() saw
    // cleanup()'s bookkeeping (:186-202)
    void swept(uint64_t removed, uint64_t entries_before, int64_t now_ns) {
        swept_begin(now_ns);
        swept_result(removed, entries_before);
    }
    // The same in two steps, for an owner that only ENQUEUES the sweep and hears what it removed later (the engine): the
    // first step is what must not wait -- next_cleanup moves on and the counts restart, so the trigger does not fire again
    // for the batches in between -- the second adapts the interval (:186-196) and re-bases next_cleanup on it.
    void swept_begin(int64_t now_ns) {
        expired_at_sweep_ = expired_;
        sweep_now_ = now_ns;
        next_ = now_ns + cur_;
        expired_ = 0;
        ops_ = 0;
    }
    void swept_result(uint64_t removed, uint64_t entries_before) {
        if (removed == 0 && expired_at_sweep_ == 0) cur_ = cur_ * 2 < max_ ? cur_ * 2 : max_;
        else if ((double)removed > (double)entries_before * 0.5) cur_ = cur_ / 2 > min_ ? cur_ / 2 : min_;
        last_removed_ = removed;
        last_total_ = entries_before;
        next_ = sweep_now_ + cur_;
    }
    int64_t current_interval_ns() const { return cur_; }
    int64_t next_cleanup_ns() const { return next_; }
    uint64_t operations() const { return ops_; }
    uint64_t last_removed() const { return last_removed_; }

  private:
    int64_t min_, max_, cur_, next_;
    uint64_t max_ops_, map_capacity_;
    uint64_t ops_ = 0, expired_ = 0, last_removed_ = 0, last_total_ = 0;
    uint64_t expired_at_sweep_ = 0;
    int64_t sweep_now_ = 0;
    Why why_ = NOT_DUE;
};

} // namespace sweep
} // namespace throttlecrab
