// shard.hip -- multi-GPU, `replicate` mode as library calls: every rank is handed the whole global batch, keeps the requests whose
// keys it owns (tc_route_batch, one destination, on the engine's grouping streams a few steps AHEAD of the evaluation) and decides
// them on its own engine.  Round 4 drove this from Python: route_batch + a poll of the pinned count word + one batch call per
// chunk + bookkeeping = 26-38 us of host time per step for a 42-63 us step (one scheduling hiccup away from
// being the bound with eight ranks on one node's cores).  Here a step is ONE call, like tc_exchange_step for the exchange.
// The reference shards on the client (README.md:247-249); nothing on the decision path is a collective.
#include "engine.hpp"

#include <sched.h>
#include <time.h>

struct tc_shard {
    tc_engine* e = nullptr;
    uint32_t rank = 0, world = 0, ring = 0;
    uint64_t keys_per_shard = 0, max_global = 0;
    std::vector<uint32_t*> slots;      // [ring] routed slot columns (device, max_global entries each)
    uint32_t* counts_dev = nullptr;    // [ring][world]
    std::vector<uint32_t*> counts_host; // [ring] pinned: [world + 1]: counts, then the step's tag
    uint64_t next_route = 0;           // steps routed so far (a repeated call for a step already routed does nothing)
    uint64_t wait_ns = 0;              // host time spent waiting for a router's tag
    uint64_t wait_limit_ns = 30ull * 1000000000ull;
};

static inline uint64_t shard_mono_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

extern "C" int tc_shard_destroy(tc_shard* x) {
    if (!x) return TC_E_OK;
    (void)hipSetDevice(x->e->device);
    (void)hipDeviceSynchronize(); // (routers on the grouping streams may still write the columns)
    for (uint32_t* p : x->slots)
        if (p) (void)hipFree(p);
    if (x->counts_dev) (void)hipFree(x->counts_dev);
    for (uint32_t* p : x->counts_host) tc_host_free(p);
    delete x;
    return TC_E_OK;
}

extern "C" int tc_shard_create(tc_engine* e, const tc_shard_config* c, tc_shard** out) {
    if (!e || !c || !out || c->struct_size < sizeof(tc_shard_config)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    rt::Map m;
    if (!rt::make_map(c->world, c->keys_per_shard, &m) || c->rank >= c->world) return fail(e, TC_E_INVALID_ARG, "tc_shard_create: world 1..64, rank < world, keys_per_shard 1..2^32");
    if (c->ring < 2 || c->ring > 64 || c->max_global == 0 || c->max_global > 0x7FFFFFFFull)
        return fail(e, TC_E_INVALID_ARG, "tc_shard_create: ring 2..64 routed batches kept, max_global 1..2^31");
    if (e->key_mode) return fail(e, TC_E_UNSUPPORTED, "tc_shard_create: slot-mode engines (string keys are sharded by key: tc_route_keys_host)");
    TC_HIP(e, hipSetDevice(e->device));
    tc_shard* x = new (std::nothrow) tc_shard;
    if (!x) return TC_E_NOMEM;
    x->e = e;
    x->rank = c->rank, x->world = c->world, x->ring = c->ring;
    x->keys_per_shard = c->keys_per_shard, x->max_global = c->max_global;
    x->slots.assign(c->ring, nullptr);
    x->counts_host.assign(c->ring, nullptr);
    bool ok = hipMalloc(&x->counts_dev, (size_t)c->ring * c->world * sizeof(uint32_t)) == hipSuccess;
    for (uint32_t r = 0; ok && r < c->ring; ++r) {
        ok = hipMalloc(&x->slots[r], c->max_global * sizeof(uint32_t)) == hipSuccess;
        if (ok) {
            x->counts_host[r] = (uint32_t*)tc_host_alloc((c->world + 1) * sizeof(uint32_t));
            ok = x->counts_host[r] != nullptr;
            if (ok) memset(x->counts_host[r], 0, (c->world + 1) * sizeof(uint32_t));
        }
    }
    if (!ok) {
        tc_shard_destroy(x);
        return fail(e, TC_E_NOMEM, "tc_shard_create: memory for the routed columns");
    }
    if (const char* ws = getenv("TCGPU_EXCHANGE_WAIT_S")) {
        const double sec = atof(ws);
        if (sec > 0) x->wait_limit_ns = (uint64_t)(sec * 1e9);
    }
    *out = x;
    return TC_E_OK;
}

extern "C" int tc_shard_route(tc_shard* x, uint64_t step, const uint32_t* d_global_id, uint64_t n) {
    if (!x || !d_global_id) return TC_E_INVALID_ARG;
    tc_engine* e = x->e;
    TC_CHECK_POISON(e);
    if (n == 0 || n > x->max_global) return fail(e, TC_E_INVALID_ARG, "tc_shard_route: a global batch holds 1..max_global requests");
    if (step < x->next_route) return TC_E_OK;
    if (step != x->next_route) return fail(e, TC_E_INVALID_ARG, "tc_shard_route: steps are routed in order");
    const uint32_t r = (uint32_t)(step % x->ring);
    tc_route rq;
    memset(&rq, 0, sizeof rq);
    rq.struct_size = sizeof rq;
    rq.world = x->world;
    rq.keys_per_shard = x->keys_per_shard;
    rq.n = n;
    rq.global_id = d_global_id;
    rq.only = (int32_t)x->rank;
    // on a grouping stream, behind the grouping of the batch that last read this ring entry (the engine remembers it)
    rq.flags = TC_ROUTE_AHEAD;
    rq.out_slot = x->slots[r];
    rq.out_count = x->counts_dev + (size_t)r * x->world;
    rq.out_count_host = x->counts_host[r];
    rq.tag = (uint32_t)(step + 1);
    const int rc = tc_route_batch(e, &rq);
    if (rc == TC_E_OK) x->next_route = step + 1;
    return rc;
}

// the requests of `step` this rank owns, as batches of at most max_batch (each writing its outputs behind the previous one's);
// tmpl: outputs (device arrays that hold the rank's whole share of a global batch), the rate plan / quantity / timestamp, flags
extern "C" int tc_shard_evaluate(tc_shard* x, uint64_t step, const tc_batch* tmpl, uint64_t* decided) {
    if (!x || !tmpl || tmpl->struct_size < offsetof(tc_batch, n_segments)) return TC_E_INVALID_ARG;
    tc_engine* e = x->e;
    TC_CHECK_POISON(e);
    if (step >= x->next_route) return fail(e, TC_E_INVALID_ARG, "tc_shard_evaluate: the step has not been routed");
    if (step + x->ring < x->next_route + 1) return fail(e, TC_E_INVALID_ARG, "tc_shard_evaluate: the step's routed column has been overwritten (ring too short for this look-ahead)");
    const uint32_t r = (uint32_t)(step % x->ring);
    volatile uint32_t* ch = x->counts_host[r];
    if (ch[x->world] != (uint32_t)(step + 1)) { // the router's tag: routed a few steps ago -- no wait in steady state
        const uint64_t t0 = shard_mono_ns();
        uint32_t spins = 0;
        while (ch[x->world] != (uint32_t)(step + 1)) {
            if ((++spins & 1023u) == 0u) {
                TC_CHECK_POISON(e);
                if (shard_mono_ns() - t0 > x->wait_limit_ns) return fail(e, TC_E_AGAIN, "tc_shard_evaluate: gave up waiting for the router's counts");
                sched_yield();
            }
        }
        x->wait_ns += shard_mono_ns() - t0;
    }
    const uint64_t mine = ch[x->rank];
    tc_batch b;
    memset(&b, 0, sizeof b);
    memcpy(&b, tmpl, std::min<size_t>(tmpl->struct_size, sizeof b));
    b.struct_size = sizeof b;
    b.flags |= TC_B_DEVICE_PTRS | TC_B_INPUTS_READY;
    b.n_segments = 0;
    const uint64_t cap = e->max_batch;
    // tmpl->n, when given, is what the caller's output arrays hold: a skewed step whose share is larger must not write past them
    // (the share comes from the router's count, nothing compared it with the arrays) -- checked before any chunk is issued
    if (tmpl->n != 0 && mine > tmpl->n)
        return fail(e, TC_E_INVALID_ARG, "tc_shard_evaluate: this rank's share of the step is larger than the output arrays (tmpl->n)");
    if (mine > cap && ((b.flags & TC_B_GROUPED_OUTPUT) || b.allowed_bits))
        return fail(e, TC_E_UNSUPPORTED, "tc_shard_evaluate: grouped output / allowed_bits of a share larger than max_batch");
    for (uint64_t at = 0; at < mine; at += cap) {
        tc_batch c = b;
        c.n = std::min<uint64_t>(cap, mine - at);
        c.slot = x->slots[r] + at;
        if (at) {
            if (c.allowed) c.allowed += at;
            if (c.limit) c.limit += at;
            if (c.remaining) c.remaining += at;
            if (c.reset_after_ns) c.reset_after_ns += at;
            if (c.retry_after_ns) c.retry_after_ns += at;
            if (c.status) c.status += at;
            if (c.result4) c.result4 += 4 * at;
            if (c.decisions) c.decisions += at;
            c.flags &= ~TC_B_OUTPUTS_IDLE; // (only the batch that starts the arrays may preset them)
        }
        const int rc = tc_rate_limit_batch_slots(e, &c);
        if (rc != TC_E_OK) {
            if (at) e->err += " (tc_shard_evaluate: a later chunk of the step failed after " + std::to_string(at) + " requests were applied)";
            return rc;
        }
    }
    if (decided) *decided = mine;
    return TC_E_OK;
}

extern "C" int tc_shard_step(tc_shard* x, uint64_t step, const uint32_t* d_global_id_ahead, uint64_t n_ahead, uint32_t route_ahead, const tc_batch* tmpl,
                             uint64_t* decided) {
    if (!x) return TC_E_INVALID_ARG;
    if (route_ahead + 2u > x->ring) return fail(x->e, TC_E_INVALID_ARG, "tc_shard_step: route_ahead needs a ring of at least route_ahead + 2 routed batches");
    int rc;
    if (d_global_id_ahead && (rc = tc_shard_route(x, step + route_ahead, d_global_id_ahead, n_ahead)) != TC_E_OK) return rc;
    return tc_shard_evaluate(x, step, tmpl, decided);
}

extern "C" int tc_shard_wait_ns(tc_shard* x, uint64_t* out) {
    if (!x || !out) return TC_E_INVALID_ARG;
    *out = x->wait_ns;
    x->wait_ns = 0;
    return TC_E_OK;
}
