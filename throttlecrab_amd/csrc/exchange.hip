// exchange.hip -- multi-GPU: one rank's side of the exchange (every GPU routes ITS slice of the global stream straight into
// the owners' inboxes and evaluates what lands in its own), as library calls.  Round 3 drove these phases from Python
// (74 us of host time per step in `route` alone: flow control over the mailboxes, a ctypes array of inbox pointers and a
// tensor view per destination per call); here a step is one call.  The reference shards on the client
// (README.md:247-249: "use client-side sharding by key"); nothing on the decision path is a collective.
#include "engine.hpp"

#include <sched.h>
#include <time.h>

struct tc_exchange {
    tc_engine* e = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 0, ring = 0, seg_cap = 0, flags = 0;
    uint64_t keys_per_shard = 0;
    std::vector<uint32_t*> inbox;       // [world]: base of destination d's [ring][world][seg_cap] array, as this process maps it
    uint32_t* mail = nullptr;           // shared host memory [world][ring][world][2]: (count, step + 1)
    int64_t* done = nullptr;            // shared host memory [world]: steps whose inboxes the destination has evaluated
    static constexpr uint32_t ROUTES = 8; // routers in flight (their count blocks)
    uint32_t* counts_dev = nullptr;     // [ROUTES][world]
    uint32_t* counts_host[ROUTES] = {}; // pinned: [world + 1]: counts, then the router's tag
    std::deque<std::pair<uint64_t, hipEvent_t>> evaluated; // steps whose evaluation is enqueued, oldest first
    std::vector<hipEvent_t> pool;
    std::vector<uint32_t*> dst;         // scratch: the inboxes of one step
    std::vector<const uint32_t*> seg_ptr;
    std::vector<uint32_t> seg_n;
    uint64_t wait_ns[3] = {0, 0, 0};    // host time spent waiting: for inbox slots (route), for the router's tag (post), for the sources (collect)
    uint64_t next_route = 0, next_post = 0; // a phase repeated for a step it has already done (a TC_E_AGAIN retry of tc_exchange_step) does nothing
    uint64_t wait_limit_ns = 30ull * 1000000000ull; // a blocking wait on a peer / the device gives up after this long (TCGPU_EXCHANGE_WAIT_S)
};

static inline uint32_t* inbox_of(const tc_exchange* x, uint32_t d, uint32_t k, uint32_t s) {
    return x->inbox[d] + ((size_t)k * x->world + s) * x->seg_cap;
}
static inline uint32_t* mail_of(const tc_exchange* x, uint32_t d, uint32_t k, uint32_t s) {
    return x->mail + (((size_t)d * x->ring + k) * x->world + s) * 2;
}

// steps whose evaluation has finished free their inbox slots (in order: done counts steps)
static void publish_done(tc_exchange* x) {
    while (!x->evaluated.empty()) {
        const hipError_t rc = hipEventQuery(x->evaluated.front().second);
        if (rc == hipErrorNotReady) {
            (void)hipGetLastError();
            break;
        }
        __atomic_store_n(&x->done[x->rank], (int64_t)x->evaluated.front().first + 1, __ATOMIC_RELEASE);
        x->pool.push_back(x->evaluated.front().second);
        x->evaluated.pop_front();
    }
}

extern "C" int tc_exchange_create(tc_engine* e, const tc_exchange_config* c, tc_exchange** out) {
    if (!e || !c || !out || c->struct_size < sizeof(tc_exchange_config)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (c->world == 0 || c->world > TC_MAX_SEGMENTS || c->rank >= c->world || c->ring < 2 || c->seg_cap == 0 || !c->inbox || !c->mail || !c->done)
        return fail(e, TC_E_INVALID_ARG, "tc_exchange_create: world 1..64, rank < world, ring >= 2, seg_cap > 0, arrays given");
    rt::Map m;
    if (!rt::make_map(c->world, c->keys_per_shard, &m)) return fail(e, TC_E_INVALID_ARG, "tc_exchange_create: keys_per_shard 1..2^32");
    for (uint32_t d = 0; d < c->world; ++d)
        if (!c->inbox[d]) return fail(e, TC_E_INVALID_ARG, "tc_exchange_create: NULL inbox");
    TC_HIP(e, hipSetDevice(e->device));
    tc_exchange* x = new (std::nothrow) tc_exchange;
    if (!x) return TC_E_NOMEM;
    x->e = e;
    x->device = e->device;
    x->rank = c->rank, x->world = c->world, x->ring = c->ring, x->seg_cap = c->seg_cap, x->flags = c->flags;
    x->keys_per_shard = c->keys_per_shard;
    x->inbox.assign(c->inbox, c->inbox + c->world);
    x->mail = c->mail;
    x->done = c->done;
    x->dst.resize(c->world);
    x->seg_ptr.resize(c->world);
    x->seg_n.resize(c->world);
    if (hipMalloc(&x->counts_dev, (size_t)tc_exchange::ROUTES * c->world * sizeof(uint32_t)) != hipSuccess) {
        delete x;
        return fail(e, TC_E_NOMEM, "tc_exchange_create: device memory");
    }
    (void)hipMemset(x->counts_dev, 0, (size_t)tc_exchange::ROUTES * c->world * sizeof(uint32_t));
    for (uint32_t r = 0; r < tc_exchange::ROUTES; ++r) {
        x->counts_host[r] = (uint32_t*)tc_host_alloc((c->world + 1) * sizeof(uint32_t));
        if (!x->counts_host[r]) {
            tc_exchange_destroy(x);
            return fail(e, TC_E_NOMEM, "tc_exchange_create: pinned memory");
        }
        memset(x->counts_host[r], 0, (c->world + 1) * sizeof(uint32_t));
    }
    TC_HIP(e, hipDeviceSynchronize());
    if (const char* ws = getenv("TCGPU_EXCHANGE_WAIT_S")) {
        const double sec = atof(ws);
        if (sec > 0) x->wait_limit_ns = (uint64_t)(sec * 1e9);
    }
    *out = x;
    return TC_E_OK;
}

extern "C" int tc_exchange_destroy(tc_exchange* x) {
    if (!x) return TC_E_OK;
    (void)hipSetDevice(x->device); // (the engine may be gone already: nothing of it is touched here)
    for (auto& p : x->evaluated) {
        (void)hipEventSynchronize(p.second);
        (void)hipEventDestroy(p.second);
    }
    for (hipEvent_t ev : x->pool) (void)hipEventDestroy(ev);
    for (uint32_t r = 0; r < tc_exchange::ROUTES; ++r) tc_host_free(x->counts_host[r]);
    if (x->counts_dev) (void)hipFree(x->counts_dev);
    delete x;
    return TC_E_OK;
}

extern "C" int tc_exchange_wait_ns(tc_exchange* x, uint64_t out[3]) {
    if (!x || !out) return TC_E_INVALID_ARG;
    for (int i = 0; i < 3; ++i) {
        out[i] = x->wait_ns[i];
        x->wait_ns[i] = 0;
    }
    return TC_E_OK;
}

extern "C" int tc_exchange_poll(tc_exchange* x) {
    if (!x) return TC_E_INVALID_ARG;
    publish_done(x);
    return TC_E_OK;
}

// a wait of the host on other ranks / on the device: nothing in TC_X_NONBLOCKING mode (the caller comes back)
static inline uint64_t mono_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
#define TC_X_WAIT(x, which, cond)                               \
    do {                                                        \
        if (!(cond)) {                                          \
            const uint64_t _t0 = mono_ns();                     \
            uint32_t _spins = 0;                                \
            while (!(cond)) {                                   \
                publish_done(x);                                \
                if ((x)->flags & TC_X_NONBLOCKING) return TC_E_AGAIN; \
                if ((++_spins & 1023u) == 0u) {                 \
                    TC_CHECK_POISON((x)->e);                    \
                    /* (a dead peer or a mis-set pipeline must not hang the caller for good) */ \
                    if (mono_ns() - _t0 > (x)->wait_limit_ns)   \
                        return fail((x)->e, TC_E_AGAIN, "tc_exchange: gave up waiting for a peer rank / the device (TCGPU_EXCHANGE_WAIT_S); the step was not done"); \
                    sched_yield();                              \
                }                                               \
            }                                                   \
            (x)->wait_ns[which] += mono_ns() - _t0;             \
        }                                                       \
    } while (0)

extern "C" int tc_exchange_route(tc_exchange* x, uint64_t step, const uint32_t* d_global_id, uint32_t n) {
    if (!x || !d_global_id) return TC_E_INVALID_ARG;
    tc_engine* e = x->e;
    TC_CHECK_POISON(e);
    if (n == 0 || n > x->seg_cap) return fail(e, TC_E_INVALID_ARG, "tc_exchange_route: a slice holds 1..seg_cap requests");
    if (step < x->next_route) return TC_E_OK;
    // (a router's count block is reused every ROUTES steps: its counts must have been posted by then)
    if (step >= x->next_post + tc_exchange::ROUTES) return fail(e, TC_E_INVALID_ARG, "tc_exchange_route: more than 8 steps ahead of tc_exchange_post");
    const uint32_t r = (uint32_t)(step % tc_exchange::ROUTES), k = (uint32_t)(step % x->ring);
    // flow control: a destination's inbox slot is free once it has evaluated step - ring
    const int64_t need = (int64_t)step - (int64_t)x->ring + 1;
    for (uint32_t d = 0; d < x->world; ++d) TC_X_WAIT(x, 0, __atomic_load_n(&x->done[d], __ATOMIC_ACQUIRE) >= need);
    for (uint32_t d = 0; d < x->world; ++d) x->dst[d] = inbox_of(x, d, k, x->rank);
    tc_route rq;
    memset(&rq, 0, sizeof rq);
    rq.struct_size = sizeof rq;
    rq.world = x->world;
    rq.keys_per_shard = x->keys_per_shard;
    rq.n = n;
    rq.global_id = d_global_id;
    rq.only = -1;
    // (the router's output goes to the inboxes, never to a batch's slot column: it need not wait for any batch in flight)
    rq.flags = TC_ROUTE_AHEAD | TC_ROUTE_NO_READERS;
    rq.out_count = x->counts_dev + (size_t)r * x->world;
    rq.out_count_host = x->counts_host[r];
    rq.tag = (uint32_t)(step + 1);
    rq.out_dst = x->dst.data();
    const int rc = tc_route_batch(e, &rq);
    if (rc == TC_E_OK) x->next_route = step + 1;
    return rc;
}

extern "C" int tc_exchange_post(tc_exchange* x, uint64_t step) {
    if (!x) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(x->e);
    if (step < x->next_post) return TC_E_OK;
    const uint32_t r = (uint32_t)(step % tc_exchange::ROUTES), k = (uint32_t)(step % x->ring);
    volatile uint32_t* ch = x->counts_host[r];
    TC_X_WAIT(x, 1, ch[x->world] == (uint32_t)(step + 1)); // the router's tag: every segment has landed (routed a few steps ago: no wait in steady state)
    for (uint32_t d = 0; d < x->world; ++d) {
        uint32_t* m = mail_of(x, d, k, x->rank);
        __atomic_store_n(&m[0], ch[d], __ATOMIC_RELAXED);
        __atomic_store_n(&m[1], (uint32_t)(step + 1), __ATOMIC_RELEASE); // (count before tag: a reader that sees the tag sees the count)
    }
    x->next_post = step + 1;
    return TC_E_OK;
}

extern "C" int tc_exchange_collect(tc_exchange* x, uint64_t step, uint32_t* counts) {
    if (!x) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(x->e);
    const uint32_t k = (uint32_t)(step % x->ring);
    for (uint32_t s = 0; s < x->world; ++s) {
        uint32_t* m = mail_of(x, x->rank, k, s);
        TC_X_WAIT(x, 2, __atomic_load_n(&m[1], __ATOMIC_ACQUIRE) == (uint32_t)(step + 1));
        x->seg_n[s] = __atomic_load_n(&m[0], __ATOMIC_RELAXED);
        x->seg_ptr[s] = inbox_of(x, x->rank, k, s);
        if (counts) counts[s] = x->seg_n[s];
    }
    return TC_E_OK;
}

// the `world` inboxes of `step` as ONE batch whose slot column comes in pieces, sources in rank order (so that a key's
// requests keep the order of the global stream), in chunks of at most max_batch; `tmpl` gives the outputs, the rate /
// quantity / timestamp and the flags (device pointers; TC_B_INPUTS_READY is added)
extern "C" int tc_exchange_evaluate(tc_exchange* x, uint64_t step, const tc_batch* tmpl, uint64_t* decided) {
    if (!x || !tmpl || tmpl->struct_size < offsetof(tc_batch, n_segments)) return TC_E_INVALID_ARG;
    tc_engine* e = x->e;
    TC_CHECK_POISON(e);
    int rc = tc_exchange_collect(x, step, nullptr);
    if (rc != TC_E_OK) return rc;
    tc_batch b;
    memset(&b, 0, sizeof b);
    memcpy(&b, tmpl, std::min<size_t>(tmpl->struct_size, sizeof b));
    b.struct_size = sizeof b;
    b.flags |= TC_B_DEVICE_PTRS | TC_B_INPUTS_READY;
    b.slot = nullptr;
    uint64_t total = 0;
    for (uint32_t s = 0; s < x->world; ++s) total += x->seg_n[s];
    const uint64_t cap = e->max_batch;
    // (checked BEFORE any chunk is issued -- ADVICE r4: found while building the second chunk, the first had already changed
    // the resident state and the step's completion was never recorded)
    //  tmpl->n, when given, is the capacity of the caller's output arrays
    if (tmpl->n != 0 && total > tmpl->n) return fail(e, TC_E_INVALID_ARG, "tc_exchange_evaluate: the step is larger than the output arrays (tmpl->n)");
    if (total > cap && (b.flags & TC_B_GROUPED_OUTPUT)) return fail(e, TC_E_UNSUPPORTED, "tc_exchange_evaluate: grouped output of a step larger than max_batch");
    if (total > cap && b.allowed_bits) return fail(e, TC_E_UNSUPPORTED, "tc_exchange_evaluate: allowed_bits of a step larger than max_batch");
    // cut the concatenation into chunks of at most max_batch requests (the common case: one chunk, nothing copied)
    const uint32_t* piece_ptr[2 * TC_MAX_SEGMENTS];
    uint32_t piece_n[2 * TC_MAX_SEGMENTS];
    uint64_t at = 0; // requests already handed over: where this chunk's outputs start
    uint32_t s = 0, off = 0;
    while (at < total) {
        uint32_t np = 0;
        uint64_t size = 0;
        while (s < x->world && size < cap && np < TC_MAX_SEGMENTS) {
            const uint32_t left = x->seg_n[s] - off;
            const uint32_t take = (uint32_t)std::min<uint64_t>(left, cap - size);
            if (take) {
                piece_ptr[np] = x->seg_ptr[s] + off;
                piece_n[np] = take;
                ++np;
                size += take;
            }
            if (take == left) {
                ++s;
                off = 0;
            } else {
                off += take;
            }
        }
        tc_batch c = b;
        c.n = size;
        c.n_segments = np;
        c.seg_slot = piece_ptr;
        c.seg_n = piece_n;
        if (at) { // later chunks write behind the earlier ones
            if (c.allowed) c.allowed += at;
            if (c.limit) c.limit += at;
            if (c.remaining) c.remaining += at;
            if (c.reset_after_ns) c.reset_after_ns += at;
            if (c.retry_after_ns) c.retry_after_ns += at;
            if (c.status) c.status += at;
            if (c.result4) c.result4 += 4 * at;
            if (c.decisions) c.decisions += at;
            c.flags &= ~TC_B_OUTPUTS_IDLE; // (only the chunk that starts the arrays may preset them)
        }
        rc = tc_rate_limit_batch_slots(e, &c);
        if (rc != TC_E_OK) {
            // a chunk failed with earlier chunks of the step already applied: the step cannot be completed, and sources that
            // wait for its inbox slots must not spin on a `done` word that never moves -- the engine is poisoned (sticky
            // TC_E_INVARIANT on this rank; the peers' waits run into their limit)
            if (at) {
                *e->poison_host = 1u;
                e->err += " (tc_exchange_evaluate: a later chunk of the step failed after earlier ones were applied: engine poisoned)";
            }
            return rc;
        }
        at += size;
    }
    // completion of this step's evaluation frees its inbox slots: an event on the stream the ENGINE evaluates on
    // (the Python version recorded it on torch's current stream, which need not be that stream)
    hipEvent_t ev = nullptr;
    if (!x->pool.empty()) {
        ev = x->pool.back();
        x->pool.pop_back();
    } else {
        TC_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    TC_HIP(e, hipEventRecord(ev, cur_stream(e)));
    x->evaluated.emplace_back(step, ev);
    publish_done(x);
    if (decided) *decided = total;
    return TC_E_OK;
}

extern "C" int tc_exchange_step(tc_exchange* x, uint64_t step, const uint32_t* d_global_id_ahead, uint32_t n_ahead, uint32_t route_ahead,
                                uint32_t post_ahead, const tc_batch* tmpl, uint64_t* decided) {
    if (!x) return TC_E_INVALID_ARG;
    // a source may run ring - 1 steps ahead of a destination, and a router's count block comes round every ROUTES steps
    if (route_ahead >= x->ring || post_ahead > route_ahead || route_ahead - post_ahead >= tc_exchange::ROUTES)
        return fail(x->e, TC_E_INVALID_ARG, "tc_exchange_step: need post_ahead <= route_ahead < ring and route_ahead - post_ahead < 8");
    int rc;
    if (d_global_id_ahead && (rc = tc_exchange_route(x, step + route_ahead, d_global_id_ahead, n_ahead)) != TC_E_OK) return rc;
    if ((rc = tc_exchange_post(x, step + post_ahead)) != TC_E_OK) return rc;
    return tc_exchange_evaluate(x, step, tmpl, decided);
}
