// autosweep.hip -- self-cleaning: AdaptiveStore::maybe_clean_expired (adaptive_cleanup.rs:205-211; periodic.rs:128-142;
// probabilistic.rs:110-125) in front of the engine's own mutating calls, once tc_set_sweep_policy has named a policy.
//
// The reference's stores run `maybe_clean_expired(now)` at the top of compare_and_swap_with_ttl / set_if_not_exists_with_ttl
// (adaptive_cleanup.rs:229,262): count the operation, ask should_clean, clean.  Here the WHEN is the reference's rule, run on
// the host by the classes of include/throttlecrab_sweep.hpp; the WHAT is tc_sweep_expired's kernels enqueued on the engine's
// stream in front of the call that tripped the rule.  What the rule needs to know about the device -- requests allowed so
// far (= operations), the store's size, what the last sweep removed, the newest timestamp of a device column -- comes back
// through mk::PolicyFeed in pinned memory, written by a one-block kernel behind every mutating call and read here without
// ever waiting; numbers that are a few batches old only move a cleanup by those few batches (the rule is a heuristic: cleanup
// does not change decisions).  Two places where stale numbers are not good enough, both string mode:
// a key batch that might not find free slots (room check: sweeps first, and waits once for fresh numbers if the ones it
// holds are older than that sweep) and a synchronous call that did run out of slots (auto_sweep_for_retry).
#include "engine.hpp"

using throttlecrab::sweep::AdaptiveSweep;
using throttlecrab::sweep::NS;
using throttlecrab::sweep::PeriodicSweep;
using throttlecrab::sweep::ProbabilisticSweep;

static int ensure_feed(tc_engine* e) {
    tc_engine::AutoSweep& a = e->as;
    if (a.feed_host) return TC_E_OK;
    void* h = nullptr;
    TC_HIP(e, hipHostMalloc(&h, sizeof(mk::PolicyFeed), hipHostMallocDefault));
    memset(h, 0, sizeof(mk::PolicyFeed));
    void* d = nullptr;
    TC_HIP(e, hipHostGetDevicePointer(&d, h, 0));
    a.feed_host = h;
    a.feed_dev = d;
    return TC_E_OK;
}

// the newest complete record, if there is one the host has not seen
static bool read_feed(tc_engine* e, mk::PolicyFeed* out) {
    const volatile mk::PolicyFeed* f = (const volatile mk::PolicyFeed*)e->as.feed_host;
    if (!f) return false;
    const unsigned long long end = f->seq_end;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    out->allowed = f->allowed;
    out->swept = f->swept;
    out->entries = f->entries;
    out->free_slots = f->free_slots;
    out->last_now = f->last_now;
    out->inserted = f->inserted;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const unsigned long long begin = f->seq_begin;
    out->seq_end = end;
    out->seq_begin = begin;
    return begin == end && end > e->as.seen; // (begin != end: the device is in the middle of writing the next one)
}

static void ingest(tc_engine* e) {
    tc_engine::AutoSweep& a = e->as;
    mk::PolicyFeed f;
    if (!read_feed(e, &f)) return;
    a.seen = f.seq_end;
    // (tc_snapshot_load replaces the device's counter block: a total that went BACKWARDS is a new baseline, not 2^64
    // operations -- which would have tripped the operation trigger, halved the adaptive interval and pinned new_share at 1)
    auto since = [](unsigned long long total, uint64_t accounted) -> uint64_t { return total >= accounted ? total - accounted : 0ull; };
    a.host_ops += since(f.allowed, a.allowed_accounted); // operations: one per allowed request (one CAS / set_if_not_exists each)
    a.allowed_accounted = f.allowed;
    a.entries = f.entries;
    a.free_slots = f.free_slots;
    if (f.last_now > a.last_now) a.last_now = f.last_now;
    uint64_t covered = 0; // requests of the key batches this feed is the first to cover
    while (!a.keys_after.empty() && a.keys_after.front().first <= a.seen) {
        covered += a.keys_after.front().second;
        a.keys_after.pop_front();
    }
    if (covered) {
        const uint64_t fresh = since(f.inserted, a.inserted_seen);
        a.new_share[a.new_share_at++ % 8] = std::min(1.0, (double)fresh / (double)covered);
    }
    a.inserted_seen = f.inserted;
    if (a.pending && a.seen >= a.pending_seq) {
        const uint64_t removed = since(f.swept, a.swept_accounted);
        if (a.kind == TC_SWEEP_ADAPTIVE) {
            a.adaptive.swept_result(removed, a.pending_entries);
            // A store that is still past the size trigger AFTER a cleanup is full of live keys.  The reference's map grows out
            // of the trigger then (len passes capacity(), the table doubles); the engine's table does not grow, and without
            // this a store between 3/4 full and full would be swept in front of every call: the virtual map grows instead
            // (what protects the real table is the room check, not this trigger)
            a.adaptive.grow_map(a.entries + a.entries / 3 + 1);
        }
        a.stats.last_removed = removed;
        a.swept_accounted = f.swept;
        a.pending = false;
    } else if (!a.pending) {
        a.swept_accounted = f.swept; // (the caller's own tc_sweep_expired calls)
    }
    if (a.kind == TC_SWEEP_ADAPTIVE) a.adaptive.grow_map(a.entries);
}

static int enqueue_feed(tc_engine* e, const int64_t* now_last_dev, int64_t now_scalar) {
    tc_engine::AutoSweep& a = e->as;
    TC_TRY(ensure_feed(e));
    const unsigned long long seq = ++a.issued;
    hipLaunchKernelGGL(mk::k_policy_feed, dim3(1), dim3(ev::NSHARD), 0, cur_stream(e), (const unsigned long long*)e->counters,
                       (mk::PolicyFeed*)a.feed_dev, seq, now_last_dev, now_scalar, e->key_mode ? (const int*)e->kt.free_top : (const int*)nullptr,
                       e->capacity);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

// fresh numbers, waited for: everything enqueued so far (key stages on the key stream included) is reflected in them
static int feed_sync(tc_engine* e) {
    tc_engine::AutoSweep& a = e->as;
    hipStream_t s = cur_stream(e);
    if (e->k_busy) {
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    TC_TRY(enqueue_feed(e, nullptr, a.last_now));
    TC_HIP(e, hipStreamSynchronize(s));
    ingest(e);
    a.stats.feed_waits++;
    return poisoned(e);
}

// the engine starts a cleanup at now_ns: the sweep's kernels on the engine's stream, the policy's bookkeeping (first half)
static int start_sweep(tc_engine* e, int64_t now_ns) {
    tc_engine::AutoSweep& a = e->as;
    if (a.pending) ingest(e);
    TC_TRY(sweep_enqueue(e, now_ns));
    if (a.kind == TC_SWEEP_ADAPTIVE) a.adaptive.swept_begin(now_ns);
    else if (a.kind == TC_SWEEP_PERIODIC) a.periodic.swept(0, 0, now_ns);
    if (!a.pending) a.pending_entries = a.entries; // (a result still outstanding: both sweeps are reported as one)
    a.pending = true;
    a.pending_seq = a.issued + 1; // the next feed is enqueued behind these kernels
    a.stats.sweeps++;
    return TC_E_OK;
}

int auto_sweep_before(tc_engine* e, uint64_t n, bool key_batch, bool now_known, int64_t now_ns, uint64_t ops_now) {
    tc_engine::AutoSweep& a = e->as;
    if (a.kind == TC_SWEEP_NONE) return TC_E_OK;
    ingest(e);
    if (now_known && now_ns >= 0) {
        if (now_ns > a.last_now) a.last_now = now_ns;
    } else {
        now_ns = a.last_now; // timestamps in a device column (or outside the domain): the newest one the engine has been shown
    }
    const uint64_t ops = a.host_ops + ops_now;
    a.host_ops = 0;
    bool due = false;
    switch (a.kind) {
    case TC_SWEEP_ADAPTIVE:
        due = a.adaptive.on_batch(ops, now_ns, a.entries);
        // (the size trigger while a cleanup is still in flight: that cleanup is the answer, its result is not in yet)
        if (due && a.pending && a.adaptive.why() >= AdaptiveSweep::BY_EXPIRED_RATIO) due = false;
        if (due) {
            const AdaptiveSweep::Why w = a.adaptive.why();
            if (w == AdaptiveSweep::BY_TIME) a.stats.sweeps_by_time++;
            else if (w == AdaptiveSweep::BY_OPERATIONS) a.stats.sweeps_by_operations++;
            else a.stats.sweeps_by_size++;
        }
        break;
    case TC_SWEEP_PERIODIC:
        due = a.periodic.on_batch(ops, now_ns);
        if (due) a.stats.sweeps_by_time++;
        break;
    default:
        due = a.probabilistic.on_batch(ops, now_ns);
        if (due) a.stats.sweeps_by_operations++;
        break;
    }
    if (due) TC_TRY(start_sweep(e, now_ns));
    if (!key_batch || !e->key_mode) return TC_E_OK;
    // room: every request of a key batch MAY be a key the table has not seen -- but a stream's batches mostly repeat keys, and a
    // check that assumes the worst stops the pipeline (one wait per call) as soon as fewer slots are free than a few batches
    // hold requests: 1.9 instead of 1.1 ms per 1 Mi-request call with 1 Mi slots free and not one new key.  So the batches the
    // host has no numbers for yet are expected to bring twice the share of new keys the worst of the last eight looks showed
    // (+ 1/32; everything, while there is no history).  `worst`: the bound that cannot be wrong, for the decisions that wait.
    double share = 0.0;
    for (double v : a.new_share) share = std::max(share, v);
    share = std::min(1.0, 2.0 * share + 1.0 / 32.0);
    auto free_lower = [&](bool worst = false) {
        uint64_t taken = 0;
        for (const auto& kn : a.keys_after) taken += worst ? kn.second : (uint64_t)((double)kn.second * share) + 1u;
        return a.free_slots > taken ? a.free_slots - taken : 0ull;
    };
    const uint64_t expect = std::min<uint64_t>(n, (uint64_t)((double)n * share) + 1u);
    if (free_lower() >= expect) return TC_E_OK;
    // (an unproductive room sweep is not repeated before the stream's clock has moved by the policy's shortest interval:
    // a table full of LIVE keys is full, and says so -- TC_E_TABLE_FULL -- instead of sweeping in front of every batch)
    if (a.room_quiet_until != INT64_MIN && now_ns < a.room_quiet_until) return TC_E_OK;
    if (a.pending || !a.keys_after.empty()) { // the numbers are older than work in flight: fresh ones first
        TC_TRY(feed_sync(e));
        if (free_lower() >= expect) return TC_E_OK;
    }
    if (!due) { // (else: a cleanup that was due anyway has just run, at this very timestamp, and did not make the room)
        TC_TRY(start_sweep(e, now_ns));
        a.stats.sweeps_for_room++;
        TC_TRY(feed_sync(e));
    }
    if (free_lower() < expect) {
        const int64_t quiet = a.kind == TC_SWEEP_ADAPTIVE ? a.min_interval_ns : (a.kind == TC_SWEEP_PERIODIC ? a.periodic.interval_ns() : NS);
        a.room_quiet_until = now_ns > INT64_MAX - quiet ? INT64_MAX : now_ns + quiet;
    }
    return TC_E_OK;
}

int auto_sweep_after(tc_engine* e, uint64_t n, bool key_batch, const int64_t* now_last_dev, int64_t now_scalar) {
    tc_engine::AutoSweep& a = e->as;
    if (a.kind == TC_SWEEP_NONE) return TC_E_OK;
    TC_TRY(enqueue_feed(e, now_last_dev, now_scalar));
    if (key_batch && e->key_mode && !a.in_retry) a.keys_after.emplace_back(a.issued, n); // (a retried sub-batch: its requests were entered with the batch they come from)
    return TC_E_OK;
}

int auto_sweep_for_retry(tc_engine* e, int64_t now_ns) {
    tc_engine::AutoSweep& a = e->as;
    if (now_ns < 0) now_ns = a.last_now;
    TC_TRY(start_sweep(e, now_ns));
    a.stats.retries++;
    return feed_sync(e);
}

extern "C" int tc_set_sweep_policy(tc_engine* e, const tc_sweep_policy* p) {
    if (!e) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    tc_engine::AutoSweep& a = e->as;
    if (!p || p->kind == TC_SWEEP_NONE) {
        a.kind = TC_SWEEP_NONE;
        return TC_E_OK;
    }
    if (p->struct_size < sizeof(tc_sweep_policy) || p->kind > TC_SWEEP_PROBABILISTIC) return fail(e, TC_E_INVALID_ARG, "tc_set_sweep_policy: struct_size / kind");
    if (p->min_interval_ns < 0 || p->max_interval_ns < 0 || p->interval_ns < 0 || p->created_ns < 0)
        return fail(e, TC_E_INVALID_ARG, "tc_set_sweep_policy: negative time");
    TC_HIP(e, hipSetDevice(e->device));
    const int64_t min_i = p->min_interval_ns ? p->min_interval_ns : 1 * NS;
    const int64_t max_i = p->max_interval_ns ? p->max_interval_ns : 300 * NS;
    if (max_i < min_i) return fail(e, TC_E_INVALID_ARG, "tc_set_sweep_policy: max_interval < min_interval");
    // with_capacity(c) sizes the map for 1.3 c entries (rounded up as hashbrown does); by default the map is the engine's table
    const uint64_t map_cap = p->map_capacity ? p->map_capacity : e->capacity;
    a.adaptive = AdaptiveSweep(p->created_ns, map_cap, min_i, max_i, p->max_operations ? p->max_operations : 100000);
    if (!p->map_capacity) a.adaptive.set_map_capacity(e->capacity); // (no rounding: 3/4 of the slots there are)
    a.periodic = PeriodicSweep(p->created_ns, p->interval_ns ? p->interval_ns : 60 * NS);
    a.probabilistic = ProbabilisticSweep(p->cleanup_probability ? p->cleanup_probability : 1000);
    a.min_interval_ns = min_i;
    a.kind = p->kind;
    a.host_ops = 0;
    a.pending = false;
    a.keys_after.clear();
    for (double& v : a.new_share) v = 1.0;
    a.room_quiet_until = INT64_MIN;
    a.last_now = p->created_ns;
    memset(&a.stats, 0, sizeof a.stats);
    // where the device's totals stand now: operations and removals are counted from here
    TC_TRY(feed_sync(e));
    a.inserted_seen = ((const volatile mk::PolicyFeed*)a.feed_host)->inserted;
    a.stats.feed_waits = 0;
    a.host_ops = 0;
    return TC_E_OK;
}

extern "C" int tc_sweep_stats(tc_engine* e, tc_sweep_info* out) {
    if (!e || !out || out->struct_size < sizeof(tc_sweep_info)) return TC_E_INVALID_ARG;
    tc_engine::AutoSweep& a = e->as;
    if (a.kind != TC_SWEEP_NONE) { // a diagnostics call: fresh numbers, waited for (drains the engine's stream)
        TC_CHECK_POISON(e);
        TC_HIP(e, hipSetDevice(e->device));
        const uint64_t waits = a.stats.feed_waits;
        TC_TRY(feed_sync(e));
        a.stats.feed_waits = waits;
    }
    tc_sweep_info r = a.stats;
    r.struct_size = sizeof r;
    r.kind = a.kind;
    r.entries = a.entries;
    r.operations = a.host_ops + (a.kind == TC_SWEEP_ADAPTIVE ? a.adaptive.operations() : (a.kind == TC_SWEEP_PROBABILISTIC ? a.probabilistic.operations() : 0));
    r.current_interval_ns = a.kind == TC_SWEEP_ADAPTIVE ? a.adaptive.current_interval_ns() : (a.kind == TC_SWEEP_PERIODIC ? a.periodic.interval_ns() : 0);
    r.next_cleanup_ns = a.kind == TC_SWEEP_ADAPTIVE ? a.adaptive.next_cleanup_ns() : (a.kind == TC_SWEEP_PERIODIC ? a.periodic.next_cleanup_ns() : 0);
    *out = r;
    return TC_E_OK;
}
