// throttlecrab_gpu.hpp -- host-side mirror (C++17, header only) of the reference's
// interface for the accelerated path, over the C ABI of tcgpu.h.
//
// The reference is Rust and this image has no Rust toolchain, so the host side is
// written in C++ with the reference's names, argument meaning and error behaviour:
//   throttlecrab::RateLimiter<S>::{new, rate_limit}   throttlecrab/src/core/rate_limiter.rs:42-110
//   throttlecrab::RateLimitResult                      throttlecrab/src/core/rate_limiter.rs:13-22
//   throttlecrab::CellError                            throttlecrab/src/core/mod.rs:49-56
//   throttlecrab::Store                                throttlecrab/src/core/store/mod.rs:85-133
// plus the batched entry point north_star adds: RateLimiter::rate_limit_batch.
// INTEGRATION.md shows the equivalent Rust `extern "C"` binding.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstring>
#include <deque>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <variant>
#include <unordered_map>
#include <vector>

#include "tcgpu.h"

namespace throttlecrab {

using SystemTime = std::chrono::time_point<std::chrono::system_clock, std::chrono::nanoseconds>;
using Duration = std::chrono::nanoseconds;

inline int64_t to_ns(SystemTime t) { return t.time_since_epoch().count(); }

// CellError (core/mod.rs:49-56)
struct CellError {
    enum Kind { NegativeQuantity = TC_NEGATIVE_QUANTITY, InvalidRateLimit = TC_INVALID_RATE_LIMIT, Internal = TC_INTERNAL } kind;
    int64_t quantity = 0; // NegativeQuantity(n)
    std::string message;  // Internal(msg)
    std::string to_string() const {
        switch (kind) { // Display impl, core/mod.rs:58-66
            case NegativeQuantity: return "negative quantity: " + std::to_string(quantity);
            case InvalidRateLimit: return "invalid rate limit parameters";
            default: return "internal error: " + message;
        }
    }
};

// RateLimitResult (rate_limiter.rs:13-22)
struct RateLimitResult {
    int64_t limit;
    int64_t remaining;
    Duration reset_after;
    Duration retry_after;
};

// Result<(bool, RateLimitResult), CellError>
using RateLimitOutcome = std::variant<std::pair<bool, RateLimitResult>, CellError>;
inline bool is_ok(const RateLimitOutcome& o) { return o.index() == 0; }

// One request = the argument list of rate_limit (rate_limiter.rs:102-110)
struct Request {
    std::string_view key;
    int64_t max_burst;
    int64_t count_per_period;
    int64_t period;
    int64_t quantity;
    SystemTime now;
};

// WHEN the store cleans itself: the cadence of one of the reference's three stores (tcgpu.h: tc_set_sweep_policy).  The
// reference's stores run maybe_clean_expired inside every compare_and_swap_with_ttl / set_if_not_exists_with_ttl
// (adaptive_cleanup.rs:205-211,229,262); the engine does the same in front of its own mutating calls once a policy is set.
struct CleanupPolicy {
    uint32_t kind = TC_SWEEP_ADAPTIVE;
    SystemTime created = std::chrono::time_point_cast<Duration>(std::chrono::system_clock::now()); // `SystemTime::now()` in with_capacity
    Duration min_interval{0}, max_interval{0}; // adaptive (0: the reference's 1 s / 300 s)
    Duration interval{0};                      // periodic (0: 60 s)
    uint64_t max_operations = 0;               // adaptive (0: 100 000)
    uint64_t map_capacity = 0;                 // adaptive: `capacity` of with_capacity (0: the engine's own table)
    uint64_t cleanup_probability = 0;          // probabilistic (0: 1000)
    // AdaptiveStore::with_capacity / builder defaults (adaptive_cleanup.rs:12-16)
    static CleanupPolicy adaptive() { return CleanupPolicy{}; }
    // the server's adaptive store (throttlecrab-server/src/config.rs:285-304: 5 s .. 300 s, 1 000 000 operations)
    static CleanupPolicy server_defaults() {
        CleanupPolicy p;
        p.min_interval = std::chrono::seconds(5), p.max_interval = std::chrono::seconds(300), p.max_operations = 1000000;
        return p;
    }
    static CleanupPolicy periodic(Duration every = std::chrono::seconds(60)) {
        CleanupPolicy p;
        p.kind = TC_SWEEP_PERIODIC, p.interval = every;
        return p;
    }
    static CleanupPolicy probabilistic(uint64_t one_in = 1000) {
        CleanupPolicy p;
        p.kind = TC_SWEEP_PROBABILISTIC, p.cleanup_probability = one_in;
        return p;
    }
    static CleanupPolicy none() { // the owner calls cleanup() itself
        CleanupPolicy p;
        p.kind = TC_SWEEP_NONE;
        return p;
    }
    CleanupPolicy& created_at(SystemTime t) {
        created = t;
        return *this;
    }
};

// The GPU-resident store; plays the role of AdaptiveStore (adaptive_cleanup.rs) and
// implements the Store trait's three operations.
class GpuStore {
  public:
    // AdaptiveStore::with_capacity (adaptive_cleanup.rs:93-106); max_batch bounds rate_limit_batch
    // track_denied: keep a denial counter per key for Metrics' top denied keys (metrics.rs:24-76)
    // policy: the store cleans itself like AdaptiveStore by default (CleanupPolicy::none(): only when cleanup() is called)
    explicit GpuStore(uint64_t capacity = 1000, uint64_t max_batch = 1 << 16, int device = 0, bool track_denied = false,
                      const CleanupPolicy& policy = CleanupPolicy::adaptive()) {
        tc_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.flags = TC_CFG_KEY_MODE | (track_denied ? TC_CFG_TRACK_DENIED : 0u);
        cfg.device_id = device;
        cfg.capacity = capacity;
        cfg.max_batch = max_batch;
        int err = 0;
        e_ = tc_engine_create(&cfg, &err);
        if (!e_) throw std::runtime_error("tc_engine_create failed: " + std::to_string(err));
        max_batch_ = max_batch;
        try {
            set_cleanup_policy(policy);
        } catch (...) {
            tc_engine_destroy(e_);
            throw;
        }
    }
    void set_cleanup_policy(const CleanupPolicy& p) {
        tc_sweep_policy c{};
        c.struct_size = sizeof c;
        c.kind = p.kind;
        c.created_ns = to_ns(p.created);
        c.min_interval_ns = p.min_interval.count(), c.max_interval_ns = p.max_interval.count(), c.interval_ns = p.interval.count();
        c.max_operations = p.max_operations, c.map_capacity = p.map_capacity, c.cleanup_probability = p.cleanup_probability;
        check(tc_set_sweep_policy(e_, &c));
    }
    // what the store's own cleanups have done (sweeps by trigger, retries after a full table, the policy's state)
    tc_sweep_info cleanup_stats() {
        tc_sweep_info r{};
        r.struct_size = sizeof r;
        check(tc_sweep_stats(e_, &r));
        return r;
    }
    // AdaptiveStore::len(): entries in the store, expired ones that no cleanup has removed yet included
    uint64_t len() {
        uint64_t c[TC_CNT_COUNT];
        check(tc_counters(e_, c));
        return c[TC_CNT_LIVE_SLOTS]; // (string mode: the keys that hold a slot right now)
    }
    GpuStore(const GpuStore&) = delete;
    GpuStore& operator=(const GpuStore&) = delete;
    GpuStore(GpuStore&& o) noexcept : e_(o.e_), max_batch_(o.max_batch_) { o.e_ = nullptr; }
    ~GpuStore() { tc_engine_destroy(e_); }

    // trait Store (store/mod.rs:85-133); Err(String) -> exception
    bool compare_and_swap_with_ttl(std::string_view key, int64_t old_v, int64_t new_v, Duration ttl, SystemTime now) {
        int ok = 0;
        check(tc_store_compare_and_swap_with_ttl(e_, bytes(key), key.size(), old_v, new_v, (uint64_t)ttl.count(), to_ns(now), &ok));
        return ok != 0;
    }
    std::optional<int64_t> get(std::string_view key, SystemTime now) {
        int64_t v = 0;
        int found = 0;
        check(tc_store_get(e_, bytes(key), key.size(), to_ns(now), &v, &found));
        return found ? std::optional<int64_t>(v) : std::nullopt;
    }
    bool set_if_not_exists_with_ttl(std::string_view key, int64_t value, Duration ttl, SystemTime now) {
        int ok = 0;
        check(tc_store_set_if_not_exists_with_ttl(e_, bytes(key), key.size(), value, (uint64_t)ttl.count(), to_ns(now), &ok));
        return ok != 0;
    }
    // AdaptiveStore::cleanup (adaptive_cleanup.rs:173-203), called explicitly (the store also runs it by itself: CleanupPolicy)
    uint64_t cleanup(SystemTime now) {
        uint64_t removed = 0;
        check(tc_sweep_expired(e_, to_ns(now), &removed));
        return removed;
    }
    tc_engine* handle() { return e_; }
    uint64_t max_batch() const { return max_batch_; }

  private:
    static const uint8_t* bytes(std::string_view k) { return reinterpret_cast<const uint8_t*>(k.data()); }
    void check(int rc) {
        if (rc != TC_E_OK) throw std::runtime_error(std::string("tcgpu: ") + tc_last_error(e_));
    }
    tc_engine* e_;
    uint64_t max_batch_ = 0;
};

// RateLimiter<GpuStore> (rate_limiter.rs:42-58)
class RateLimiter {
  public:
    explicit RateLimiter(GpuStore store) : store_(std::move(store)) {}

    // rate_limiter.rs:102-250
    RateLimitOutcome rate_limit(std::string_view key, int64_t max_burst, int64_t count_per_period, int64_t period,
                                int64_t quantity, SystemTime now) {
        tc_result r{};
        int rc = tc_rate_limit(store_.handle(), reinterpret_cast<const uint8_t*>(key.data()), key.size(), max_burst,
                               count_per_period, period, quantity, to_ns(now), &r);
        if (rc != TC_E_OK) return CellError{CellError::Internal, 0, tc_last_error(store_.handle())};
        return outcome(r.status, r.allowed, r.limit, r.remaining, r.reset_after_ns, r.retry_after_ns, quantity);
    }

    // The batched entry point: exactly rate_limit applied to reqs[0], reqs[1], ... in order.
    std::vector<RateLimitOutcome> rate_limit_batch(const std::vector<Request>& reqs) {
        const size_t n = reqs.size();
        std::vector<RateLimitOutcome> out;
        out.reserve(n);
        if (!n) return out;
        if (n == 1) { // one launch either way; the single call skips the marshalling
            const Request& r = reqs[0];
            out.push_back(rate_limit(r.key, r.max_burst, r.count_per_period, r.period, r.quantity, r.now));
            return out;
        }
        // (up to 1024 requests the engine answers with one launch, ~35-45 us; beyond that with its pipeline)
        std::vector<uint8_t> arena;
        std::vector<uint32_t> off(n + 1, 0);
        std::vector<int64_t> burst(n), count(n), period(n), qty(n), now(n);
        std::vector<tc_decision> dec(n); // one 32-byte record per request; `limit` == the request's max_burst
        for (size_t i = 0; i < n; ++i) {
            arena.insert(arena.end(), reqs[i].key.begin(), reqs[i].key.end());
            off[i + 1] = (uint32_t)arena.size();
            burst[i] = reqs[i].max_burst;
            count[i] = reqs[i].count_per_period;
            period[i] = reqs[i].period;
            qty[i] = reqs[i].quantity;
            now[i] = to_ns(reqs[i].now);
        }
        if (arena.empty()) arena.push_back(0);
        tc_batch b{};
        b.struct_size = sizeof b;
        b.n = n;
        b.key_bytes = arena.data();
        b.key_off = off.data();
        b.max_burst = burst.data();
        b.count_per_period = count.data();
        b.period = period.data();
        b.quantity = qty.data();
        b.now_ns = now.data();
        b.decisions = dec.data();
        int rc = tc_rate_limit_batch_keys(store_.handle(), &b);
        for (size_t i = 0; i < n; ++i) {
            if (rc != TC_E_OK && rc != TC_E_TABLE_FULL)
                out.push_back(CellError{CellError::Internal, 0, tc_last_error(store_.handle())});
            else
                out.push_back(outcome(dec[i].status, dec[i].allowed, burst[i], dec[i].remaining, dec[i].reset_after_ns,
                                      dec[i].retry_after_ns, qty[i]));
        }
        return out;
    }

    // ---- the same, pipelined -------------------------------------------------------------------
    // submit_batch() marshals the requests into one of FLIGHTS sets of pinned buffers and only enqueues
    // them (TC_B_ASYNC: the PCIe transfers and the evaluation overlap with the caller marshalling the
    // next batch); collect_batch() waits for the OLDEST submitted batch and returns its outcomes.  The
    // outcomes are those of rate_limit_batch called once per submission, in submission order.  At most
    // FLIGHTS batches may be in flight (collect before submitting a fourth).
    static constexpr size_t FLIGHTS = 3;
    static constexpr size_t SMALL_BATCH = 1024; // what the engine serves with one launch (k_small_batch)
    size_t in_flight() const { return order_.size(); }
    void submit_batch(const std::vector<Request>& reqs) {
        if (order_.size() >= FLIGHTS) throw std::logic_error("submit_batch: collect_batch() first");
        const size_t n = reqs.size();
        if (n > store_.max_batch()) throw std::invalid_argument("submit_batch: more requests than the store's max_batch");
        size_t fi = 0;
        while (flights_[fi].busy) ++fi;
        Flight& f = flights_[fi];
        if (n > SMALL_BATCH) { // (may throw: nothing has changed yet)
            size_t bytes = 0;
            for (const Request& r : reqs) bytes += r.key.size();
            f.reserve(store_.max_batch(), bytes + 1);
        }
        f.n = n;
        f.ready.clear();
        f.rc = TC_E_OK;
        f.async = false;
        f.busy = true;
        order_.push_back(fi);
        if (n <= SMALL_BATCH) { // lightly loaded: the engine's one-launch path, answered at once (it runs behind
            f.ready = rate_limit_batch(reqs); // the batches in flight, in stream order)
            return;
        }
        uint32_t at = 0;
        for (size_t i = 0; i < n; ++i) {
            const Request& r = reqs[i];
            if (!r.key.empty()) std::memcpy(f.arena + at, r.key.data(), r.key.size());
            at += (uint32_t)r.key.size();
            f.off[i + 1] = at;
            f.col[0][i] = r.max_burst; // (kept either way: collect_batch reports a request's limit and quantity from them)
            f.col[1][i] = r.count_per_period;
            f.col[2][i] = r.period;
            f.col[3][i] = r.quantity;
            f.col[4][i] = to_ns(r.now);
        }
        // Round 6, TC_B_PLAN_DICT: the distinct (max_burst, count_per_period, period) triples of the batch once, a 16-bit index
        // per request, the quantities as u32 -- 6 bytes per request over PCIe instead of 32.  The caller's interface is the
        // reference's (rate_limiter.rs:102-110); more than 65 536 triples or a quantity outside u32: the wide columns.
        const bool compact = f.encode(reqs);
        tc_batch b{};
        b.struct_size = sizeof b;
        b.flags = TC_B_ASYNC | (compact ? TC_B_PLAN_DICT : 0u);
        b.n = n;
        b.key_bytes = f.arena;
        b.key_off = f.off;
        if (compact) {
            b.plan_dict = f.dict;
            b.n_plans = (uint32_t)(f.plans.size() / 3);
            b.plan_id = f.plan_id;
            b.quantity32 = f.qty32;
        } else {
            b.max_burst = f.col[0];
            b.count_per_period = f.col[1];
            b.period = f.col[2];
            b.quantity = f.col[3];
        }
        b.now_ns = f.col[4];
        b.decisions = f.dec;
        f.rc = tc_rate_limit_batch_keys(store_.handle(), &b);
        f.async = f.rc == TC_E_OK;
        if (!f.async) f.err = tc_last_error(store_.handle());
    }
    std::vector<RateLimitOutcome> collect_batch() {
        if (order_.empty()) throw std::logic_error("collect_batch: nothing in flight");
        Flight& f = flights_[order_.front()];
        order_.pop_front();
        std::vector<RateLimitOutcome> out;
        out.reserve(f.n);
        if (f.async) {
            // batches complete in submission order: everything but the younger asynchronous ones must be done
            uint32_t younger = 0;
            for (size_t fi : order_) younger += flights_[fi].async ? 1u : 0u;
            const int rc = tc_wait_batches(store_.handle(), younger);
            for (size_t i = 0; i < f.n; ++i) {
                if (rc != TC_E_OK) out.push_back(CellError{CellError::Internal, 0, tc_last_error(store_.handle())});
                else
                    out.push_back(outcome(f.dec[i].status, f.dec[i].allowed, f.col[0][i], f.dec[i].remaining, f.dec[i].reset_after_ns,
                                          f.dec[i].retry_after_ns, f.col[3][i]));
            }
        } else if (f.rc != TC_E_OK) {
            for (size_t i = 0; i < f.n; ++i) out.push_back(CellError{CellError::Internal, 0, f.err});
        } else {
            out = std::move(f.ready);
        }
        f.busy = false;
        return out;
    }

    GpuStore& store() { return store_; }

  private:
    // one set of pinned staging buffers (tc_host_alloc), grown on demand while the set is idle
    struct Flight {
        uint8_t* arena = nullptr;
        size_t arena_cap = 0;
        uint32_t* off = nullptr;
        int64_t* col[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        tc_decision* dec = nullptr;
        // TC_B_PLAN_DICT: the batch's dictionary (pinned, 65 536 triples), its plan ids and u32 quantities; the encoder's table
        int64_t* dict = nullptr;
        uint16_t* plan_id = nullptr;
        uint32_t* qty32 = nullptr;
        std::vector<int64_t> plans;
        std::unordered_map<std::string, uint32_t> plan_of;
        size_t rows = 0, n = 0;
        std::vector<RateLimitOutcome> ready; // small batches answered by single calls
        std::string err;
        int rc = TC_E_OK;
        bool async = false, busy = false;
        Flight() = default;
        Flight(const Flight&) = delete;
        Flight& operator=(const Flight&) = delete;
        Flight(Flight&& o) noexcept { *this = std::move(o); }
        Flight& operator=(Flight&& o) noexcept {
            std::swap(arena, o.arena), std::swap(arena_cap, o.arena_cap), std::swap(off, o.off), std::swap(dec, o.dec);
            std::swap(dict, o.dict), std::swap(plan_id, o.plan_id), std::swap(qty32, o.qty32), std::swap(plans, o.plans), std::swap(plan_of, o.plan_of);
            for (int j = 0; j < 5; ++j) std::swap(col[j], o.col[j]);
            std::swap(rows, o.rows), std::swap(n, o.n), std::swap(ready, o.ready), std::swap(err, o.err);
            std::swap(rc, o.rc), std::swap(async, o.async), std::swap(busy, o.busy);
            return *this;
        }
        ~Flight() {
            tc_host_free(arena), tc_host_free(off), tc_host_free(dec);
            tc_host_free(dict), tc_host_free(plan_id), tc_host_free(qty32);
            for (int64_t* c : col) tc_host_free(c);
        }
        // the batch's triples -> dict / plan_id / qty32; false: more than 65 536 of them, or a quantity that is no u32
        bool encode(const std::vector<Request>& reqs) {
            plans.clear();
            plan_of.clear();
            int64_t last[3] = {0, 0, 0};
            uint32_t last_id = UINT32_MAX;
            for (size_t i = 0; i < reqs.size(); ++i) {
                const Request& r = reqs[i];
                if (r.quantity < 0 || r.quantity > (int64_t)UINT32_MAX) return false;
                const int64_t t[3] = {r.max_burst, r.count_per_period, r.period};
                uint32_t id;
                if (last_id != UINT32_MAX && t[0] == last[0] && t[1] == last[1] && t[2] == last[2]) {
                    id = last_id; // (the same plan as the request before: what a connection's pipeline looks like)
                } else {
                    const auto ins = plan_of.emplace(std::string(reinterpret_cast<const char*>(t), sizeof t), (uint32_t)plan_of.size());
                    id = ins.first->second;
                    if (ins.second) {
                        if (id >= 65536u) return false;
                        plans.insert(plans.end(), t, t + 3);
                    }
                    last[0] = t[0], last[1] = t[1], last[2] = t[2], last_id = id;
                }
                plan_id[i] = (uint16_t)id;
                qty32[i] = (uint32_t)r.quantity;
            }
            std::memcpy(dict, plans.data(), plans.size() * sizeof(int64_t));
            return true;
        }
        template <class T>
        static T* pinned(size_t count) {
            T* p = static_cast<T*>(tc_host_alloc(count * sizeof(T)));
            if (!p) throw std::bad_alloc();
            return p;
        }
        void reserve(size_t max_rows, size_t arena_bytes) {
            if (rows < max_rows) {
                tc_host_free(off), tc_host_free(dec);
                for (int64_t*& c : col) tc_host_free(c), c = nullptr;
                tc_host_free(plan_id), tc_host_free(qty32);
                plan_id = nullptr, qty32 = nullptr;
                if (!dict) dict = pinned<int64_t>((size_t)65536 * 3);
                plan_id = pinned<uint16_t>(max_rows);
                qty32 = pinned<uint32_t>(max_rows);
                off = pinned<uint32_t>(max_rows + 1);
                dec = pinned<tc_decision>(max_rows);
                for (int64_t*& c : col) c = pinned<int64_t>(max_rows);
                rows = max_rows;
            }
            off[0] = 0;
            if (arena_cap < arena_bytes) {
                tc_host_free(arena);
                arena = nullptr;
                arena_cap = 0;
                arena = pinned<uint8_t>(arena_bytes * 2);
                arena_cap = arena_bytes * 2;
            }
        }
    };
    Flight flights_[FLIGHTS];
    std::deque<size_t> order_; // flights in submission order

    static RateLimitOutcome outcome(uint8_t status, uint8_t allowed, int64_t limit, int64_t remaining, int64_t reset_ns,
                                    int64_t retry_ns, int64_t quantity) {
        switch (status) {
            case TC_OK:
                return std::make_pair(allowed != 0, RateLimitResult{limit, remaining, Duration(reset_ns), Duration(retry_ns)});
            case TC_NEGATIVE_QUANTITY: return CellError{CellError::NegativeQuantity, quantity, {}};
            case TC_INVALID_RATE_LIMIT: return CellError{CellError::InvalidRateLimit, 0, {}};
            default: return CellError{CellError::Internal, 0, "outside the validated domain (see tcgpu.h)"};
        }
    }
    GpuStore store_;
};

} // namespace throttlecrab
