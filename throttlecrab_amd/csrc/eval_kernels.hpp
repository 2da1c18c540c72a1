// eval_kernels.hpp -- the evaluation kernels of the batched GCRA engine (gfx950): request
// decoding, output writing, decision counters and
//   k_eval_unique   one lane per request (caller promised unique slots)
//   k_eval_sorted   batches with one `now` / `quantity`: closed form per key run
//   k_eval_general  per-request now / quantity / rate: wave-cooperative, cross-wave hand-over chain
//   k_commit_list, k_fold_counters, k_pack_bits
// Kernels have internal linkage: every translation unit (engine.hpp) compiles the ones it launches.(the engine and the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tcgpu.h"
#include "gcra_math.hpp"
#include "key_table.hpp"

namespace ev {

using tc::Cell;
using tc::Decision;
using tc::RateClass;
constexpr int BLOCK = 256;
constexpr uint32_t F_REGISTERED = 1u;    // Params.flags: per-slot registered rate plan
constexpr uint32_t F_UNIFORM_CLASS = 2u; // every slot carries plan `uniform_class`: skip the rate_id[] read
constexpr uint32_t F_FIXED = 4u;         // TC_CFG_FIXED_PARAMS engine: 8-byte TAT column (tat8), timestamps < 2^62
constexpr uint32_t F_PREFILL0 = 16u;     // TC_B_OUTPUTS_IDLE batch whose `allowed` bytes were all set to 0 ahead of the evaluation
constexpr uint32_t F_PREFILL1 = 32u;     // ... to 1: the evaluation only stores the decisions that differ from the fill
constexpr uint32_t F_DEBUG_NO_ANNOUNCE = 64u; // tc_debug_break_wait: row 0 never announces (the watchdog's test)
constexpr uint32_t F_NO_EARLIER = 128u;  // TCGPU_GENERAL_EARLIER=0 (A/B): k_eval_general without the earlier-state rule
constexpr uint32_t F_GENERAL_RUNS = 256u; // k_eval_general settles a run of allowed requests in one round (TCGPU_GENERAL_RUNS=0: one per round)
#ifdef TCGPU_DEBUG_KNOBS
constexpr uint32_t F_DEBUG_NOSTORE = 8u; // measurement builds only (make DEBUG_KNOBS=1 + TCGPU_DEBUG_NO_DECISION_STORE=1): the lean kernel skips its decision bytes
#else
constexpr uint32_t F_DEBUG_NOSTORE = 0u; // (round 6: the shipped library has no switch that corrupts results -- the test compiles away)
#endif
constexpr uint32_t MAX_CLASSES = 65536;  // rate_id is u16; id 0 = "not registered"
constexpr uint32_t TOPK_MAX = 10000;     // tc_top_denied: MAX_DENIED_KEYS_LIMIT (throttlecrab-server/src/metrics.rs:17)

// ---------------------------------------------------------------------------
// kernel argument block
// ---------------------------------------------------------------------------
struct Params {
    uint32_t n;
    uint32_t flags;
    const uint32_t* slot;
    const int64_t* burst;
    const int64_t* count;
    const int64_t* period;
    const int64_t* q;
    const int64_t* now;
    int64_t burst_s, count_s, period_s, q_s, now_s;
    uint8_t* allowed;
    int64_t* limit;
    int64_t* remaining;
    int64_t* reset;
    int64_t* retry;
    uint8_t* status;
    int64_t* result4;
    tc_decision* decisions;
    uint32_t* order; // TC_B_GROUPED_OUTPUT: output row k belongs to request order[k] (rows are in grouped order)
    Cell* cells;
    int64_t* tat8; // TC_CFG_FIXED_PARAMS: the resident state as one TAT per key (cells == nullptr then)
    const uint16_t* rate_id;
    const RateClass* classes;
    uint32_t uniform_class;
    uint64_t capacity;
    unsigned long long* counters;
    uint32_t* denied; // per-slot denial counters (TC_CFG_TRACK_DENIED) or nullptr
    uint64_t* row_bits; // TC_B_GROUPED_OUTPUT + allowed_bits on a batch whose runs are all regular: the evaluation packs
                        // the decisions of its 64-row waves itself (one ballot, one 8-byte store), no byte column
    // round 6: runs of at least heavy_min requests are noted in heavy[] (heavy_note below); heavy_min == 0: nothing is
    unsigned long long* heavy = nullptr;
    uint32_t heavy_min = 0, heavy_tag = 0;
};

// Which slots are HOT?  The grouping wants to know (range_part.hpp: a slot that takes thousands of a batch's requests is peeled
// out of the range partition), the host decides, and the evaluation is where every run's length is in a register anyway: the
// lane holding the LAST request of a run of at least heavy_min requests notes min(length, 2^20 - 1) << 44 | tag << 32 | slot in a
// small table in device memory -- with atomicMax, at TWO places hashed from the slot (one in each half of the table): where two
// slots meet, the longer run stays, so the heaviest slots of a stream are never the ones that get lost (a first version stored
// plainly at one place hashed from slot and batch: every copy missed one of the eight heaviest keys of a Zipf stream now and
// then, and the batch grouped without it paid k_finish's pass through global memory).  Two atomics per heavy run, nothing
// waits.  mk::k_heavy_publish copies the table to pinned host memory now and then and clears it: a copy holds the longest
// runs since the one before.
constexpr uint32_t HEAVY_BITS = 12, HEAVY_HALF = 1u << HEAVY_BITS, HEAVY_SLOTS = 2u * HEAVY_HALF;
constexpr uint32_t HEAVY_TAG_BITS = 12, HEAVY_LEN_MAX = (1u << 20) - 1u;
__device__ __forceinline__ void heavy_note(const Params& p, uint32_t slot, uint32_t len) {
    const unsigned long long v = ((unsigned long long)(len < HEAVY_LEN_MAX ? len : HEAVY_LEN_MAX) << 44) |
                                 ((unsigned long long)(p.heavy_tag & ((1u << HEAVY_TAG_BITS) - 1u)) << 32) | slot;
    atomicMax(&p.heavy[(slot * 0x9E3779B1u) >> (32 - HEAVY_BITS)], v);
    atomicMax(&p.heavy[HEAVY_HALF + ((slot * 0x85EBCA6Bu + 0x27D4EB2Fu) >> (32 - HEAVY_BITS))], v);
}

// First position of the run of `slot` that position k belongs to, given that position k - 1 holds the same slot.  Gallops back
// over equal slots, then bisects: all it assumes is that a slot's requests are CONTIGUOUS in `sorted` -- true of a sorted batch
// and of a batch whose hot slots were gathered behind the sorted rest (range_part.hpp) -- and it costs 2 log2(run so far) loads
// instead of log2(k).
__device__ __forceinline__ uint32_t run_start_of(const uint64_t* __restrict__ sorted, uint32_t k, uint32_t slot) {
    uint32_t hi = k, lo, step = 1;
    for (;;) {
        if (hi < step) {
            if ((uint32_t)(sorted[0] >> 32) == slot) return 0u;
            lo = 0u;
            break;
        }
        const uint32_t q = hi - step;
        if ((uint32_t)(sorted[q] >> 32) != slot) {
            lo = q;
            break;
        }
        hi = q;
        step <<= 1;
    }
    while (hi - lo > 1u) { // sorted[lo] is another slot, sorted[hi] is mine
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint32_t)(sorted[mid] >> 32) == slot) hi = mid;
        else lo = mid;
    }
    return hi;
}

struct Req {
    int64_t ei, dvt, q, now, limit;
    int status;
};

// Request i against `slot`: arguments, derived rate and status (rate_limiter.rs:111-123).
__device__ __forceinline__ Req make_req(const Params& p, uint32_t i, uint32_t slot) {
    Req r;
    r.q = p.q ? p.q[i] : p.q_s;
    r.now = p.now ? p.now[i] : p.now_s;
    r.ei = r.dvt = r.limit = 0;
    if (slot >= p.capacity) {
        r.status = tc::ST_INTERNAL;
        return r;
    }
    if (p.flags & F_REGISTERED) {
        const uint32_t id = (p.flags & F_UNIFORM_CLASS) ? p.uniform_class : (uint32_t)p.rate_id[slot];
        const RateClass rc = p.classes[id]; // class 0 is all zero: burst 0 = never registered
        r.ei = rc.ei;
        r.dvt = rc.dvt;
        r.limit = rc.burst;
        if (r.q < 0) r.status = tc::ST_NEGATIVE_QUANTITY;            // rate_limiter.rs:111
        else if (r.limit <= 0) r.status = tc::ST_INVALID_RATE_LIMIT; // slot never registered
        else r.status = tc::check_request(r.q, r.now, r.dvt);
        if ((p.flags & F_FIXED) && r.status == tc::ST_OK && r.now >= ((int64_t)1 << 62)) r.status = tc::ST_INTERNAL;
    } else {
        const int64_t burst = p.burst ? p.burst[i] : p.burst_s;
        const int64_t count = p.count ? p.count[i] : p.count_s;
        const int64_t period = p.period ? p.period[i] : p.period_s;
        r.limit = burst;
        r.status = tc::derive_request(burst, count, period, r.q, r.now, r.ei, r.dvt);
    }
    return r;
}

// make_req for the batches k_eval_sorted* see (one `now`, one `quantity`; parameters registered per slot or
// scalar), with the slot's rate class already loaded by the caller.
__device__ __forceinline__ Req make_req_rc(const Params& p, uint32_t slot, const RateClass& rc) {
    Req r;
    r.q = p.q_s;
    r.now = p.now_s;
    r.ei = r.dvt = r.limit = 0;
    if (slot >= p.capacity) {
        r.status = tc::ST_INTERNAL;
        return r;
    }
    if (p.flags & F_REGISTERED) {
        r.ei = rc.ei;
        r.dvt = rc.dvt;
        r.limit = rc.burst;
        if (r.q < 0) r.status = tc::ST_NEGATIVE_QUANTITY;            // rate_limiter.rs:111
        else if (r.limit <= 0) r.status = tc::ST_INVALID_RATE_LIMIT; // slot never registered
        else r.status = tc::check_request(r.q, r.now, r.dvt);
        if ((p.flags & F_FIXED) && r.status == tc::ST_OK && r.now >= ((int64_t)1 << 62)) r.status = tc::ST_INTERNAL;
    } else {
        r.limit = p.burst_s;
        r.status = tc::derive_request(p.burst_s, p.count_s, p.period_s, r.q, r.now, r.ei, r.dvt);
    }
    return r;
}

// (returns the decision: 1 = allowed)
__device__ __forceinline__ bool write_out(const Params& p, uint32_t i, const Req& r, const Decision& d) {
    const bool ok = r.status == tc::ST_OK;
    if (p.allowed) p.allowed[i] = (ok && d.allowed) ? 1 : 0;
    if (p.status) p.status[i] = (uint8_t)r.status;
    if (p.limit) p.limit[i] = ok ? r.limit : 0;
    if (p.remaining) p.remaining[i] = ok ? d.remaining : 0;
    if (p.reset) p.reset[i] = ok ? d.reset_after : 0;
    if (p.retry) p.retry[i] = ok ? d.retry_after : 0;
    if (p.result4) {
        // one 32-byte record = two 16-byte stores into the same 64-byte granule
        longlong2* r4 = reinterpret_cast<longlong2*>(p.result4 + (size_t)i * 4);
        r4[0] = make_longlong2(ok ? r.limit : 0, ok ? d.remaining : 0);
        r4[1] = make_longlong2(ok ? d.reset_after : 0, ok ? d.retry_after : 0);
    }
    if (p.decisions) {
        longlong2* dr = reinterpret_cast<longlong2*>(p.decisions + i);
        const long long flags = (long long)((ok && d.allowed) ? 1u : 0u) | ((long long)(uint8_t)r.status << 8);
        dr[0] = make_longlong2(ok ? d.remaining : 0, ok ? d.reset_after : 0);
        dr[1] = make_longlong2(ok ? d.retry_after : 0, flags); // allowed | status << 8, pad = 0 (little endian)
    }
    return ok && d.allowed;
}

// LEAN batches ask for the decision bytes only (no status / limit / result columns, request order): the other
// output pointers are then not even looked at, which keeps the kernel's scalar registers under the 80 that eight
// 256-thread blocks per CU need (MI355X_MICROARCH.md, residency).
template <bool LEAN>
__device__ __forceinline__ bool put_out(const Params& p, uint32_t i, const Req& r, const Decision& d) {
    if (LEAN) {
        const bool a = r.status == tc::ST_OK && d.allowed;
        // (prefilled batches: the byte is already there unless this decision is the batch's minority one)
        if (!(p.flags & (a ? (F_PREFILL1 | F_DEBUG_NOSTORE) : (F_PREFILL0 | F_DEBUG_NOSTORE)))) p.allowed[i] = a ? 1 : 0;
        return a;
    }
    return write_out(p, i, r, d);
}

// The resident state of `slot` under either layout.  TC_CFG_FIXED_PARAMS engines keep one TAT per key
// (p.tat8, p.cells == nullptr; tc::fixed_cell rebuilds the expiry from the key's dvt); the raw load does
// not need the rate yet, so it can be issued before the plan is known.
template <bool FIXED>
__device__ __forceinline__ Cell load_raw(const Params& p, uint32_t slot) {
    if (FIXED) {
        Cell c;
        c.tat = p.tat8[slot];
        c.expiry = 0;
        return c;
    }
    return tc::load_cell(&p.cells[slot]);
}
template <bool FIXED>
__device__ __forceinline__ void store_state(const Params& p, uint32_t slot, const Cell& c) {
    if (FIXED) p.tat8[slot] = c.tat;
    else tc::store_cell(&p.cells[slot], c);
}
// the same with the layout looked up at run time (wave-uniform branch): the kernels that are not hot
__device__ __forceinline__ Cell load_state_rt(const Params& p, uint32_t slot, int64_t dvt) {
    if (p.tat8) return tc::fixed_cell(p.tat8[slot], dvt);
    return tc::load_cell(&p.cells[slot]);
}
__device__ __forceinline__ void store_state_rt(const Params& p, uint32_t slot, const Cell& c) {
    if (p.tat8) p.tat8[slot] = c.tat;
    else tc::store_cell(&p.cells[slot], c);
}

// Decision counters are accumulated in NSHARD shards (block b -> shard b % NSHARD):
// a device-scope atomic on ONE address costs ~12 ns and serialises, so 4096
// blocks hitting one counter would add ~50 us to a 1 Mi-request batch.
// k_fold_counters sums the shards into the canonical TC_CNT_* block on demand.
constexpr int NSHARD = 256;
constexpr int SHARD_WORDS = 4; // allowed, denied, errors, pad

// sum three per-thread counts over the block, one atomic each per block
// `hint` (one designated block per launch, else nullptr): a pinned host word that receives "most of this block's decisions
// were allowed" -- the host reads it, without ever waiting, to choose the fill value of later TC_B_OUTPUTS_IDLE batches
template <int NT = BLOCK>
__device__ __forceinline__ void block_count3(uint32_t a, uint32_t b, uint32_t c, unsigned long long* counters, uint32_t* hint = nullptr) {
    __shared__ uint32_t s_cnt[3][NT / 64];
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_cnt[0][wave] = a;
        s_cnt[1][wave] = b;
        s_cnt[2][wave] = c;
    }
    __syncthreads();
    // (opaque copies: inside k_eval_lean_hot's loop over stretches the compiler otherwise works these two addresses out ahead of
    // the loop and keeps them -- on its stack: the kernel's only scratch, which costs its FIRST dispatch in a process a scratch
    // allocation on the queue, hundreds of microseconds, in the middle of the stream)
    uint32_t tx = threadIdx.x, bx = blockIdx.x;
    asm volatile("" : "+v"(tx), "+s"(bx));
    if (tx < 3) {
        uint32_t t = 0;
        for (int w = 0; w < NT / 64; ++w) t += s_cnt[tx][w];
        if (t) {
            unsigned long long* shard = counters + (TC_CNT_COUNT + 1) + (bx % NSHARD) * SHARD_WORDS;
            atomicAdd(&shard[tx], (unsigned long long)t);
        }
    }
    if (hint != nullptr && threadIdx.x == 64) {
        uint32_t ta = 0, tb = 0;
        for (int w = 0; w < NT / 64; ++w) {
            ta += s_cnt[0][w];
            tb += s_cnt[1][w];
        }
        __hip_atomic_store(hint, ta >= tb ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Denial counters per key (the device-side analogue of Metrics::record_request_with_key's
// TopDeniedKeys update, throttlecrab-server/src/metrics.rs:24-50,162-173).  In the sorted kernels the
// lanes of one key are contiguous, so the first lane of each run inside the wave adds the run's
// denials with ONE atomic (a hot key would otherwise serialise ~12 ns per denied request).
__device__ __forceinline__ void wave_denied_add(const Params& p, uint32_t slot, bool denied_here) {
    if (!p.denied) return; // wave-uniform
    const int lane = threadIdx.x & 63;
    const uint32_t prev = __shfl_up(slot, 1, 64);
    const bool run_head = lane == 0 || prev != slot;
    const unsigned long long heads = __ballot(run_head), dm = __ballot(denied_here);
    if (run_head && slot < p.capacity) {
        const unsigned long long later = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const int end = later ? lane + __builtin_ctzll(later) : 63; // last lane of my run
        const unsigned long long run = (end == 63 ? ~0ull : ((2ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
        const uint32_t cnt = (uint32_t)__popcll(dm & run);
        if (cnt) atomicAdd(&p.denied[slot], cnt);
    }
}

// ---------------------------------------------------------------------------
// K1: one lane per request, slots unique within the batch
// ---------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(BLOCK) void k_eval_unique(Params p) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t na = 0, nd = 0, ne = 0;
    if (i < p.n) {
        const uint32_t slot = p.slot[i];
        const Req r = make_req(p, i, slot);
        Decision d;
        d.allowed = false;
        d.remaining = d.reset_after = d.retry_after = 0;
        if (r.status == tc::ST_OK) {
            Cell c = load_state_rt(p, slot, r.dvt);
            d = tc::gcra_step<FULL>(c, r.ei, r.dvt, r.q, r.now);
            if (d.allowed) store_state_rt(p, slot, c);
            na = d.allowed;
            nd = !d.allowed;
            if (p.denied && nd) atomicAdd(&p.denied[slot], 1u); // unique slots: no contention
        } else {
            ne = 1;
        }
        write_out(p, i, r, d);
        if (p.order) p.order[i] = i;
    }
    block_count3(na, nd, ne, p.counters);
}

// ---------------------------------------------------------------------------
// K0: ONE request (RateLimiter::rate_limit called singly, rate_limiter.rs:102-250): one thread resolves
// the key, evaluates and stores; one launch and one 48-byte copy back instead of the batch pipeline.
// ---------------------------------------------------------------------------
struct InlineKey {
    uint8_t bytes[64];
    uint32_t len;
    uint32_t is_inline;
};
struct OneResult {
    tc_decision d;
    int64_t limit;
    uint32_t table_full;
    uint32_t pad;
};
static __global__ void k_rate_limit_one(Params p, kt::Table t, int key_mode, InlineKey ik, const uint8_t* __restrict__ long_key,
                                 uint32_t slot_in, OneResult* __restrict__ out, unsigned long long* inserted) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t slot = slot_in;
    bool full = false;
    if (key_mode) slot = kt::find_or_bind_one(t, ik.is_inline ? ik.bytes : long_key, ik.len, true, &full, inserted);
    Req r = make_req(p, 0, slot < p.capacity ? slot : (uint32_t)p.capacity); // unresolved key: Internal
    Decision d;
    d.allowed = false;
    d.remaining = d.reset_after = d.retry_after = 0;
    unsigned long long* shard = p.counters + (TC_CNT_COUNT + 1); // shard 0: allowed, denied, errors
    if (r.status == tc::ST_OK) {
        Cell c = tc::load_cell(&p.cells[slot]);
        d = tc::gcra_step<true>(c, r.ei, r.dvt, r.q, r.now);
        if (d.allowed) tc::store_cell(&p.cells[slot], c);
        atomicAdd(&shard[d.allowed ? 0 : 1], 1ull);
        if (p.denied && !d.allowed) atomicAdd(&p.denied[slot], 1u);
    } else {
        atomicAdd(&shard[2], 1ull);
    }
    const bool ok = r.status == tc::ST_OK;
    OneResult o;
    o.d.remaining = ok ? d.remaining : 0;
    o.d.reset_after_ns = ok ? d.reset_after : 0;
    o.d.retry_after_ns = ok ? d.retry_after : 0;
    o.d.allowed = (ok && d.allowed) ? 1 : 0;
    o.d.status = (uint8_t)r.status;
    for (int i = 0; i < 6; ++i) o.d.pad[i] = 0;
    o.limit = ok ? r.limit : 0;
    o.table_full = full ? 1u : 0u;
    o.pad = 0;
    *out = o;
}

// ---------------------------------------------------------------------------
// Ks: a SMALL batch (<= SMALL_MAX requests) in ONE launch of one block: resolve the keys (string mode: the
// claim / bind / follow protocol of key_table.hpp with block barriers between the phases), sort
// (slot, index) in LDS (bitonic), and let the first lane of every key run walk its requests in index order
// with the ordinary step -- any mix of timestamps, quantities and rates.  Inputs are read from and results
// written to PINNED HOST memory directly (the host wrapper copies the caller's arrays in and out), so a
// small batch costs one launch and one synchronisation (~35 us) instead of the seven launches and eight
// copies of the big pipeline (70 us slot mode, 110 us string mode) -- what the actor's loop sees whenever
// its queue holds a handful of requests (actor.rs:217-236 answers them one by one).
// ---------------------------------------------------------------------------
constexpr int SMALL_MAX = 1024;
static __global__ __launch_bounds__(SMALL_MAX) void k_small_batch(Params p, kt::Table t, int key_mode, const uint8_t* __restrict__ key_bytes,
                                                           const uint32_t* __restrict__ key_off, uint32_t* __restrict__ table_full,
                                                           unsigned long long* inserted, uint32_t* __restrict__ slot_out) {
    __shared__ uint64_t s_key[SMALL_MAX]; // slot << 32 | request index; padding = ~0
    __shared__ uint32_t s_slot[SMALL_MAX];
    const uint32_t i = threadIdx.x, n = p.n;
    uint32_t slot = kt::NO_SLOT;
    if (key_mode) {
        // a TC_B_ASYNC batch before this one may have run into a full table: its return code is delivered here
        if (i == 0 && atomicExch(t.error_flag, 0u) != 0u) table_full[1] = 1u;
        uint32_t st = kt::ST_FOUND, ax = 0;
        uint64_t h = 0;
        bool recycled = false;
        if (i < n) st = kt::probe_request<true>(t, key_bytes, key_off, n, i, slot, ax, h, recycled);
        kt::tombs_sub<SMALL_MAX>(t, recycled);
        kt::reserve_overflow<SMALL_MAX>(t, i < n, i < n ? key_off[i + 1] - key_off[i] : 0u, st, slot);
        uint32_t total = 0;
        const uint32_t rank = kt::block_rank<SMALL_MAX>(i < n && st == kt::ST_CLAIMANT, total); // one barrier
        const int top = *t.free_top; // (thread 0 moves it after the next barrier)
        if (i < n && st == kt::ST_CLAIMANT) {
            slot = kt::bind_claimant(t, key_bytes, key_off, n, i, ax, h, slot, top - 1 - (int)rank);
        } else if (i < n && st == kt::ST_NOSPACE) {
            kt::release_claim(t, key_off, i, ax, h);
            slot = kt::NO_SLOT;
        }
        if (i < n) s_slot[i] = slot;
        __syncthreads();
        if (i == 0 && total) {
            const int got = top < 0 ? 0 : (top < (int)total ? top : (int)total);
            *t.free_top = top - got;
            if (got) atomicAdd(inserted, (unsigned long long)got);
        }
        if (i < n && st == kt::ST_FOLLOWER) slot = s_slot[ax];
        if (i < n && slot == kt::NO_SLOT) *table_full = 1u; // (same value from every lane that gets here)
        if (i < n && slot_out != nullptr) slot_out[i] = slot; // (which requests were turned away: the host may apply them again)
    } else if (i < n) {
        slot = p.slot[i];
    }
    // bitonic sort of the padded power of two that holds the batch
    uint32_t P = 64;
    while (P < n) P <<= 1;
    s_key[i] = i < n ? (((uint64_t)slot << 32) | i) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t partner = i ^ j;
            if (i < P && partner > i) {
                const uint64_t a = s_key[i], b = s_key[partner];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s_key[i] = b;
                    s_key[partner] = a;
                }
            }
            __syncthreads();
        }
    }
    // the first lane of each run applies the run's requests one after the other
    uint32_t na = 0, nd = 0, ne = 0;
    if (i < n) {
        const uint64_t me = s_key[i];
        const uint32_t my_slot = (uint32_t)(me >> 32);
        if (i == 0 || (uint32_t)(s_key[i - 1] >> 32) != my_slot) {
            Cell c;
            c.tat = 0;
            c.expiry = 0;
            const bool has_cell = my_slot < p.capacity;
            // (the 8-byte layout only takes registered plans: every request of the key has the first one's dvt)
            if (has_cell) c = load_state_rt(p, my_slot, make_req(p, (uint32_t)me, my_slot).dvt);
            bool dirty = false;
            uint32_t run_denied = 0;
            for (uint32_t k = i; k < n; ++k) {
                const uint64_t e = s_key[k];
                if ((uint32_t)(e >> 32) != my_slot) break;
                const uint32_t idx = (uint32_t)e;
                const Req rq = make_req(p, idx, my_slot);
                Decision d;
                d.allowed = false;
                d.remaining = d.reset_after = d.retry_after = 0;
                if (rq.status == tc::ST_OK) {
                    d = tc::gcra_step<true>(c, rq.ei, rq.dvt, rq.q, rq.now);
                    dirty |= d.allowed;
                    na += d.allowed;
                    nd += !d.allowed;
                    run_denied += !d.allowed;
                } else {
                    ne += 1;
                }
                write_out(p, idx, rq, d);
            }
            if (dirty) store_state_rt(p, my_slot, c);
            if (p.denied && run_denied) atomicAdd(&p.denied[my_slot], run_denied);
        }
    }
    block_count3<SMALL_MAX>(na, nd, ne, p.counters);
}

// Deferred cell store for a segment that spans waves (see k_eval_sorted).
struct __attribute__((aligned(16))) PendEntry {
    Cell cell;
    uint32_t slot;
    uint32_t pad[3];
};

// ---------------------------------------------------------------------------
// K2: evaluate over the (slot, index)-sorted batch, ITEMS sorted positions per lane.
//   UNIFORM (one `now`, one `quantity`, per-slot or scalar params): every
//     request of a slot's segment is identical, so lane r of the segment derives
//     the cell left by its r predecessors in closed form (tc::run_form) and
//     applies the ordinary step to it.  Exactly one lane per segment -- the one
//     performing the last allowed step -- produces the new cell.  If the whole
//     segment sits inside one row (64 consecutive positions = one wave's j-th items) the lane stores
//     it directly (every lane of the segment loaded the old cell earlier in program order); otherwise the
//     store is parked in pend[] and applied by k_commit_list after this kernel (folding it into
//     the kernel's last block was measured slower: the hand-off needs every block to drain its stores),
//     so no lane of another wave can read a half-updated cell.
//   Batches with per-request now / quantity / rate go through k_eval_general.
//   DIRECT (the host proved every run of this batch regular, see all_runs_regular()): no store is
//     parked and no commit launch follows.  The lane owning a segment's new cell stores it itself
//     once every row holding EARLIER requests of the segment has announced (loaded[row] = seq)
//     that it read the old cell -- earlier blocks were dispatched earlier, and inside a block rows are
//     announced and waited for in position order with no block barrier between announce and wait, so a
//     row only ever waits for rows whose waves cannot be waiting for it: the wait cannot deadlock.
//     Requests AFTER the owner (rank >= n_tot: denied) may read either cell: against the
//     old one the closed form gives "denied against new0 + (n_tot-1) inc", against the new one the
//     plain step gives the same, because the new cell IS that state.
//   Loads: every load a lane will need is issued before anything waits -- the batch-wide rate class
//     (scalar), ITEMS sorted elements + the row-edge neighbours, then plan ids (per-slot plans only) and
//     ITEMS cells.  Block b covers positions [b*BLOCK*ITEMS, (b+1)*BLOCK*ITEMS); lane t's j-th position is
//     b*BLOCK*ITEMS + j*BLOCK + t.  Measured (1 Mi requests, 10 M keys): the kernel's time does not depend
//     on ITEMS for the decisions-only output (43 us: it is bound by the ~1.5 M fabric requests its random
//     16-byte cells cost, not by latency), record outputs gain 15 % at ITEMS = 4 in order, the Zipf stream
//     7 % at ITEMS = 2 when overlapped with the next batch's sort.
// ---------------------------------------------------------------------------
#ifndef TC_EVAL_LEAN_WAVES
#define TC_EVAL_LEAN_WAVES 8 // waves per SIMD the LEAN variants are compiled for (= 256-thread blocks per CU)
#endif
#ifndef TC_LEAN_HOT_WAVES
#define TC_LEAN_HOT_WAVES 6  // ... and k_eval_lean_hot (2 positions per lane; TC_LEAN_HOT_WAVES4: 4 positions)
#endif
#ifndef TC_LEAN_HOT_WAVES4
#define TC_LEAN_HOT_WAVES4 4
#endif
template <bool FULL, bool DIRECT, int ITEMS, bool FIXED, bool LEAN, int BS = BLOCK>
__device__ __forceinline__ void eval_sorted_body(const Params& p, const uint64_t* __restrict__ sorted,
                                                 PendEntry* __restrict__ pend, uint32_t* __restrict__ pend_count,
                                                 uint32_t* __restrict__ loaded, uint32_t seq,
                                                 const uint32_t* __restrict__ gate, uint32_t gate_min, uint32_t* hint, uint32_t bid, uint32_t n,
                                                 uint32_t grid, uint32_t opaque_zero = 0u) {
    // (bid, n, grid: this block's number among the `grid` blocks of the sorted part and the part's length -- blockIdx.x, p.n and
    // gridDim.x unless the kernel also runs the hot role, k_eval_sorted_lean)
    if (gate != nullptr && __builtin_nontemporal_load(gate) <= gate_min) return; // this batch took the bucket path (bucket_path.hpp)
    // (opaque_zero: 0 -- a value the compiler cannot see through when the body sits in k_eval_lean_hot's loop, so that what
    // depends on `wave` is worked out inside the loop and not kept across it on the stack; see block_count3)
    const int lane = threadIdx.x & 63, wave = (int)((threadIdx.x >> 6) + opaque_zero);
    const uint32_t block_start = bid * (BS * ITEMS);
    const bool class_by_slot = (p.flags & F_REGISTERED) && !(p.flags & F_UNIFORM_CLASS); // uniform over the grid
    const RateClass rc_batch = p.classes[p.uniform_class];                                // (class 0 if there is none)
    uint32_t kk[ITEMS];
    uint64_t me[ITEMS];
    uint32_t edge[ITEMS]; // lane 0: slot before my row; lane 63: slot after it
    // every load is unconditional (indices clamped, results masked afterwards): a load under a branch
    // makes the compiler drain the earlier ones first
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        kk[j] = block_start + j * BS + threadIdx.x;
        me[j] = sorted[min(kk[j], n - 1u)];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t ek = kk[j];
        if (lane == 0 && ek > 0) ek -= 1;
        if (lane == 63) ek += 1;
        edge[j] = (uint32_t)(sorted[min(ek, n - 1u)] >> 32); // the other lanes re-read their own element (same lines)
    }
    // (vmcnt counts loads in issue order, so whatever the next load's address depends on goes first:
    // plan ids before cells, the batch-wide class before everything)
    Cell cell[ITEMS];
    RateClass rc[ITEMS];
    uint32_t rid[ITEMS];
    bool has_cell[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (kk[j] >= n) me[j] = ~0ull;
        has_cell[j] = kk[j] < n && (uint32_t)(me[j] >> 32) < p.capacity;
        rid[j] = 0;
    }
    if (class_by_slot) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rid[j] = (uint32_t)p.rate_id[has_cell[j] ? (uint32_t)(me[j] >> 32) : 0u];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        cell[j] = load_raw<FIXED>(p, has_cell[j] ? (uint32_t)(me[j] >> 32) : 0u);
        if (!has_cell[j]) {
            cell[j].tat = FIXED ? tc::TAT_VACANT : 0;
            cell[j].expiry = 0;
        }
    }
    if (class_by_slot) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rc[j] = p.classes[rid[j]]; // class 0 is all zero: burst 0 = never registered
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rc[j] = rc_batch;
    }

    // run boundaries per row; start of my run by a max-scan of head positions over the block's range
    __shared__ uint32_t s_rowmax[ITEMS][BS / 64];
    __shared__ uint32_t s_start;
    bool head[ITEMS], is_last[ITEMS];
    uint32_t hp[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t k = kk[j];
        const bool valid = k < n;
        const uint32_t slot = (uint32_t)(me[j] >> 32);
        uint32_t prev_slot = __shfl_up(slot, 1, 64), next_slot = __shfl_down(slot, 1, 64);
        if (lane == 0) prev_slot = edge[j];               // (k == 0: not looked at)
        if (lane == 63) next_slot = edge[j];              // (k + 1 >= n: not looked at)
        head[j] = valid && (k == 0 || prev_slot != slot);
        is_last[j] = valid && ((k + 1 == n) || next_slot != slot);
        uint32_t v = head[j] ? k + 1 : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v = max(v, o);
        }
        hp[j] = v;
        if (lane == 63) s_rowmax[j][wave] = v;
    }
    if (threadIdx.x == 0 && kk[0] < n && !head[0]) {
        // the run of the block's first position began in an earlier block
        s_start = run_start_of(sorted, kk[0], (uint32_t)(me[0] >> 32));
    }
    __syncthreads();

    uint32_t na = 0, nd = 0, ne = 0;
    uint32_t seg_start[ITEMS];
    bool writer[ITEMS], denied_here[ITEMS];
    Cell wcell[ITEMS];
    uint32_t carry = 0; // latest head position (+1) in the rows before mine
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        for (int w = 0; w < BS / 64; ++w) {
            // rows before (j, wave) in position order: all of the earlier sub-tiles, and waves < mine of this one
            if (w < wave) carry = max(carry, s_rowmax[j][w]);
        }
        const uint32_t h = max(hp[j], carry);
        seg_start[j] = h ? h - 1 : s_start;
        for (int w = wave; w < BS / 64; ++w) carry = max(carry, s_rowmax[j][w]);

        const uint32_t k = kk[j];
        const bool valid = k < n;
        const uint32_t slot = (uint32_t)(me[j] >> 32), idx = (uint32_t)me[j];
        const uint32_t orow = (!LEAN && p.order) ? k : idx;
        if (!LEAN && p.order && valid) p.order[k] = idx;
        writer[j] = false;
        denied_here[j] = false;
        wcell[j].tat = 0;
        wcell[j].expiry = 0;
        bool bit = false; // my decision (every valid lane of a DIRECT batch writes its own outputs exactly once)
        if (valid) {
            const uint32_t r = k - seg_start[j];
            if (is_last[j] && p.heavy_min != 0u && r + 1u >= p.heavy_min) heavy_note(p, slot, r + 1u);
            const Req rq = make_req_rc(p, slot, rc[j]);
            Decision d;
            d.allowed = false;
            d.remaining = d.reset_after = d.retry_after = 0;
            if (rq.status != tc::ST_OK) {
                ne += 1;
                put_out<LEAN>(p, orow, rq, d);
            } else {
                Cell c = FIXED ? tc::fixed_cell(cell[j].tat, rq.dvt) : cell[j];
                const Decision d0 = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now); // c = cell after request 0
                if (!d0.allowed) {
                    // request 0 denied => state untouched => every request of the run equals request 0
                    nd += 1;
                    denied_here[j] = true;
                    bit = put_out<LEAN>(p, orow, rq, d0);
                } else if (head[j] && is_last[j]) {
                    na += 1; // a key requested once in this batch: no closed form, no 64-bit division
                    bit = put_out<LEAN>(p, orow, rq, d0);
                    writer[j] = true;
                    wcell[j] = c;
                } else {
                    // closed form of the run: tc::run_form; decisions only: tc::run_lite, the same provisos without the
                    // 64-bit division (nearly every wave holds some key with two requests, so every wave would pay it)
                    bool regular, ok_r = true, owner = false;
                    Cell v = c;
                    if (FULL) {
                        const tc::RunForm f = tc::run_form(c, rq.ei, rq.dvt, rq.q, rq.now);
                        regular = f.regular;
                        if (regular && r != 0) {
                            const int64_t jj = (int64_t)r < f.n_tot ? (int64_t)r : f.n_tot;
                            v.tat = f.new0 + (jj - 1) * f.inc; // the state my r predecessors leave behind
                            v.expiry = UINT64_MAX;
                            d = tc::gcra_step<true>(v, rq.ei, rq.dvt, rq.q, rq.now);
                            ok_r = d.allowed;
                            owner = ok_r && (is_last[j] || (int64_t)r + 1 == f.n_tot);
                        } else if (regular) {
                            owner = is_last[j] || f.n_tot == 1;
                        }
                    } else {
                        const tc::RunLite f = tc::run_lite(c, rq.ei, rq.dvt, rq.q, rq.now);
                        regular = f.regular;
                        if (regular) {
                            ok_r = r == 0 || tc::rank_allowed(f, r);
                            owner = ok_r && (is_last[j] || !tc::rank_allowed(f, r + 1u));
                            if (owner && r != 0) v = tc::cell_after(f.new0 + (int64_t)r * f.inc, rq.dvt, rq.now);
                            d.allowed = ok_r;
                        }
                    }
                    if (r == 0) {
                        na += 1;
                        bit = put_out<LEAN>(p, orow, rq, d0);
                        if (regular || is_last[j]) {
                            writer[j] = owner || is_last[j];
                            wcell[j] = c;
                        } else {
                            // irregular run (saturation, zero increment, immediate expiry): walk the rest one by one
                            if (DIRECT) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]); // the host's proof was wrong: must stay 0
                            uint32_t wd = 0;
                            for (uint32_t q = k + 1; q < n; ++q) {
                                const uint64_t nx = sorted[q];
                                if ((uint32_t)(nx >> 32) != slot) break;
                                const Decision dj = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now);
                                na += dj.allowed;
                                wd += !dj.allowed;
                                put_out<LEAN>(p, (!LEAN && p.order) ? q : (uint32_t)nx, rq, dj);
                            }
                            nd += wd;
                            if (p.denied && wd) atomicAdd(&p.denied[slot], wd); // the whole run's denials sit in this lane
                            writer[j] = true;
                            wcell[j] = c;
                        }
                    } else if (regular) {
                        na += ok_r;
                        nd += !ok_r;
                        denied_here[j] = !ok_r;
                        bit = put_out<LEAN>(p, orow, rq, d);
                        writer[j] = owner;
                        wcell[j] = v;
                    }
                    // irregular && r > 0: the head lane produced this request's outputs
                }
            }
        }
        if (DIRECT && !LEAN && p.row_bits) { // (wave-uniform; a wave's j-th items are 64 consecutive rows)
            const unsigned long long bm = __ballot(bit);
            if (lane == 0 && valid) p.row_bits[k >> 6] = bm;
        }
    }

    // (every cell load of this lane has returned: the values were consumed above; pin that down)
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) asm volatile("" ::"v"((uint32_t)wcell[j].tat), "v"((uint32_t)writer[j]) : "memory");
    asm volatile("" ::"v"(na), "v"(nd) : "memory");
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t k = kk[j];
        const bool valid = k < n;
        const uint32_t slot = (uint32_t)(me[j] >> 32);
        if (DIRECT) {
            const uint32_t gw = k >> 6; // row number
            // announce "this row has read its cells" if a later row may have to wait for it
            if (lane == 63 && valid && !is_last[j] && !((p.flags & F_DEBUG_NO_ANNOUNCE) && gw == 0u))
                __hip_atomic_store(&loaded[gw], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t row_first = k - (uint32_t)lane;
            const bool must_wait = writer[j] && seg_start[j] < row_first; // at most one such run per row
            const unsigned long long wm = __ballot(must_wait);
            if (wm) {
                const int wl = __builtin_ctzll(wm);
                const uint32_t w0 = __shfl(seg_start[j], wl, 64) >> 6;
                for (uint32_t base = w0; base < gw; base += 64) { // 64 earlier rows per round trip
                    const uint32_t w = base + (uint32_t)lane;
                    tc::SpinGuard guard;
                    while (__ballot(w < gw && __hip_atomic_load(&loaded[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq)) {
                        if (tc::spin_expired(guard)) { // (see tc::SpinGuard: flagged, never hung)
                            if (lane == 0) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            if (writer[j]) store_state<FIXED>(p, slot, wcell[j]);
        } else {
            // does my whole run live inside this row?  (ballots taken by the full wave)
            const unsigned long long heads = __ballot(head[j]), lasts = __ballot(is_last[j]);
            const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            const bool seg_in_row = ((heads & upto) != 0ull) && ((lasts >> lane) != 0ull);
            if (!writer[j]) {
            } else if (seg_in_row) {
                store_state<FIXED>(p, slot, wcell[j]);
            } else {
                const uint32_t at = atomicAdd(pend_count, 1u);
                PendEntry pe;
                pe.cell = wcell[j];
                pe.slot = slot;
                pe.pad[0] = pe.pad[1] = pe.pad[2] = 0;
                pend[at] = pe;
            }
        }
        wave_denied_add(p, slot, denied_here[j]);
    }
    block_count3<BS>(na, nd, ne, p.counters, (LEAN && bid == grid / 2) ? hint : nullptr);
}

template <bool FULL, bool DIRECT, int ITEMS, bool FIXED>
__global__ __launch_bounds__(BLOCK) void k_eval_sorted(Params p, const uint64_t* __restrict__ sorted, PendEntry* __restrict__ pend,
                                                       uint32_t* __restrict__ pend_count, uint32_t* __restrict__ loaded, uint32_t seq,
                                                       const uint32_t* __restrict__ gate, uint32_t gate_min) {
    eval_sorted_body<FULL, DIRECT, ITEMS, FIXED, false>(p, sorted, pend, pend_count, loaded, seq, gate, gate_min, nullptr, blockIdx.x, p.n, gridDim.x);
}
// decisions only, direct stores, only the `allowed` byte column asked for: at most 80 scalar and 64 vector registers,
// so that EIGHT blocks fit a CU (k_eval_sorted's 106 SGPRs admit six) -- the kernel waits on random memory, and what
// hides that is resident waves
// (ITEMS = 4: half as many blocks of twice the work, four per CU -- what skewed streams run fastest on; 512-thread blocks, the
// same waves per CU in half as many blocks, measured slower on both streams: profiles/r04_v11_eval_block_ab.txt)
// Round 6, the HOT ROLE (range_part.hpp, PART_RANK): a request of a hot slot is not grouped at all.  What the closed form of a
// run needs of it is its rank among its slot's requests -- hot_P[tile][id] (requests of the slot in earlier tiles: the scan in
// rs::k_finish) + its rank inside its tile (the partition left id << 16 | rank in info[i]) -- and, of its slot, how many requests
// of the run are allowed.  A lean batch's requests are all alike (one timestamp, one quantity, the slot's plan), so that number,
// A[id] = the ranks r < requests in the batch with tc::rank_allowed(r), is a property of the SLOT: every hot-role block works it out for the
// (at most 512) hot slots once, into LDS, together with the tile's row of hot_P -- and then walks its half tile of the batch in
// REQUEST order: a request is allowed iff rank < A[id].  One coalesced load of info and at most one byte stored per request; no
// gather left in the loop.  (A first version decided every request on its own: three dependent gathers per request -- prefix,
// slot, cell -- made this role as slow as evaluating the sorted hot runs had been.)
// The new cells -- each block computes the same ones -- are stored by the LAST hot-role block to finish: by then no block is left
// that reads the old ones (the sorted part's blocks never touch a hot slot).  It also notes the runs for the host's hot list.
struct HotEval {
    const uint32_t* info;    // nullptr: no hot role, the whole grid is the sorted part
    const uint32_t* prefix;  // hot_P
    const uint32_t* n;       // [ids] requests per hot id in this batch; [ids]: the sorted part's length
    const uint32_t* slot;    // hot id -> slot
    const uint32_t* count;   // hot ids in use
    uint32_t* done;          // hot-role blocks of this launch that have finished (zero between launches)
    uint32_t ids, tile_shift;
};
// (the record lives in DEVICE memory, one per scratch set, and the kernel is handed its address: by value its 14 scalar
// registers, loaded at the kernel's entry for both roles, pushed the lean kernel over its 80 and into scratch)
constexpr int HOT_SPLIT = 2;       // hot-role blocks per tile of the partition
constexpr uint32_t HOT_IDS_MAX = 512; // (== rp::HOT_MAX: the LDS arrays below)
constexpr uint32_t HOT_ERR = 0xFFFFFFFFu;

template <bool FIXED>
__device__ __forceinline__ void eval_hot_role(const Params& p, const HotEval& he, uint32_t hot_blocks) {
    __shared__ uint32_t s_allow[HOT_IDS_MAX];  // A[id]: requests of the slot's run that are allowed; HOT_ERR: the slot's requests are errors
    __shared__ uint32_t s_before[HOT_IDS_MAX]; // hot_P[tile][id]
    __shared__ uint32_t s_last;
    // (the slots' new cells wait in LDS for step (3): in registers they cost the kernel -- both roles -- its 64-register budget)
    __shared__ long long s_tat[HOT_IDS_MAX];
    __shared__ unsigned long long s_exp[HOT_IDS_MAX];
    __shared__ uint32_t s_len[HOT_IDS_MAX]; // requests of the slot in this batch | 0x80000000: its cell changes
    const uint32_t n = p.n, tile_len = 1u << he.tile_shift;
    const uint32_t tile = blockIdx.x / HOT_SPLIT, part = blockIdx.x % HOT_SPLIT;
    const uint32_t count = min(*he.count, min(he.ids, HOT_IDS_MAX));
    const bool class_by_slot = (p.flags & F_REGISTERED) && !(p.flags & F_UNIFORM_CLASS);
    // this half tile's notes, on their way while the slots are worked out
    constexpr uint32_t PER = 4096u / HOT_SPLIT / BLOCK; // requests per lane (tile_len == 4096: checked by the host), in two rounds of
    constexpr uint32_t HALF = PER / 2;                   // HALF: eight notes in registers at once cost the 64-register variants spills
    uint32_t info[HALF], info2[HALF];
    const uint32_t first = tile * tile_len + part * (tile_len / HOT_SPLIT) + threadIdx.x;
#pragma unroll
    for (uint32_t j = 0; j < HALF; ++j) info[j] = he.info[min(first + j * BLOCK, n - 1u)];
    // (1) the hot slots: two per thread
    for (uint32_t id = threadIdx.x; id < count; id += BLOCK) {
        const uint32_t len = he.n[id], slot = he.slot[id];
        s_before[id] = he.prefix[(size_t)tile * he.ids + id];
        uint32_t allow = 0, changes = 0;
        if (len != 0u) {
            // (one load of the plan either way: a batch-wide copy selected against it made the compiler keep both on a stack)
            const RateClass rc = p.classes[class_by_slot ? (uint32_t)p.rate_id[slot] : p.uniform_class];
            const Req rq = make_req_rc(p, slot, rc);
            if (rq.status != tc::ST_OK) {
                allow = HOT_ERR;
            } else {
                const Cell raw = load_raw<FIXED>(p, slot);
                Cell c = FIXED ? tc::fixed_cell(raw.tat, rq.dvt) : raw;
                const Decision d0 = tc::gcra_step<false>(c, rq.ei, rq.dvt, rq.q, rq.now); // c = cell after request 0
                if (d0.allowed) {
                    allow = 1;
                    if (len > 1u) {
                        // (the same closed form as the sorted part's tc::run_lite / rank_allowed: rank r is allowed <=> r < n_tot;
                        // the host proved every run of this batch regular)
                        const tc::RunLite f = tc::run_lite(c, rq.ei, rq.dvt, rq.q, rq.now);
                        if (!f.regular) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]);
                        // the last allowed rank, by bisection over the very test the sorted part applies to each request
                        // (monotone in the rank; ~20 steps per slot, no 64-bit division -- which costs this kernel a stack)
                        uint32_t lo = 0, hi = len - 1u;
                        while (lo < hi) {
                            const uint32_t mid = lo + (hi - lo + 1u) / 2u;
                            if (tc::rank_allowed(f, mid)) lo = mid;
                            else hi = mid - 1u;
                        }
                        allow = lo + 1u;
                        if (lo != 0u) c = tc::cell_after(f.new0 + (int64_t)lo * f.inc, rq.dvt, rq.now);
                    }
                    s_tat[id] = c.tat;
                    s_exp[id] = c.expiry;
                    changes = 0x80000000u;
                }
            }
        }
        s_allow[id] = allow;
        s_len[id] = len | changes;
    }
    __syncthreads();
    // (2) the requests
    uint32_t na = 0, nd = 0, ne = 0;
#pragma unroll
    for (uint32_t j = 0; j < HALF; ++j) info2[j] = he.info[min(first + (HALF + j) * BLOCK, n - 1u)];
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {
        const uint32_t pos = first + j * BLOCK, note = j < HALF ? info[j % HALF] : info2[j % HALF];
        if (pos >= n || note == 0xFFFFFFFFu) continue;
        const uint32_t id = note >> 16, a = s_allow[id];
        const bool ok = a != HOT_ERR;
        const bool al = ok && s_before[id] + (note & 0xFFFFu) < a;
        na += al;
        nd += ok && !al;
        ne += !ok;
        // (prefilled batches: the byte is already there unless this decision is the batch's minority one -- put_out<LEAN>)
        if (!(p.flags & (al ? (F_PREFILL1 | F_DEBUG_NOSTORE) : (F_PREFILL0 | F_DEBUG_NOSTORE)))) p.allowed[pos] = al ? 1 : 0;
    }
    block_count3<BLOCK>(na, nd, ne, p.counters, nullptr);
    // (3) the last hot-role block to get here stores the slots' new cells (its own copy of them: every block computed the same)
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(he.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == hot_blocks ? 1u : 0u;
    __syncthreads();
    if (s_last == 0u) return;
    for (uint32_t id = threadIdx.x; id < count; id += BLOCK) {
        const uint32_t len = s_len[id] & 0x7FFFFFFFu, slot = he.slot[id];
        if (s_len[id] & 0x80000000u) {
            Cell c;
            c.tat = s_tat[id];
            c.expiry = s_exp[id];
            store_state<FIXED>(p, slot, c);
        }
        if (p.heavy_min != 0u && len >= p.heavy_min) heavy_note(p, slot, len);
    }
    if (threadIdx.x == 0) __hip_atomic_store(he.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ITEMS, bool FIXED, int BS = BLOCK>
__global__ __launch_bounds__(BS, (ITEMS <= 2 ? TC_EVAL_LEAN_WAVES : TC_EVAL_LEAN_WAVES / 2)) __attribute__((amdgpu_num_sgpr(80))) void k_eval_sorted_lean(
    Params p, const uint64_t* __restrict__ sorted, uint32_t* __restrict__ loaded, uint32_t seq, const uint32_t* __restrict__ gate,
    uint32_t gate_min, uint32_t* hint) {
    eval_sorted_body<false, true, ITEMS, FIXED, true, BS>(p, sorted, nullptr, nullptr, loaded, seq, gate, gate_min, hint, blockIdx.x, p.n, gridDim.x);
}

// The same with the HOT ROLE in front (a kernel of its own: the loop around the sorted part below costs the 64-register variant a
// few spills, which the headline's kernel above must not pay).  Blocks [0, hot_blocks): the hot role; the others: the sorted
// part -- the requests the ranges hold.  How many they are is known on the device only; the host sized that part of the grid
// (cold_grid blocks) by a recent batch's count (a grid of 3 072 blocks of which 1 200 leave at once lengthened the hand-over to
// the next kernel on the stream: tools/gapbench).  A block takes the stretches bid, bid + cold_grid, ...: stretches are still
// started in position order by blocks dispatched in order, which is all the direct stores' waits rely on.
template <int ITEMS, bool FIXED>
// (six blocks per CU instead of the plain kernel's eight: 74-77 vector registers and NO scratch -- the variants with 2 positions per
// lane kept six loop-invariant words on the stack until the late days of round 6: nothing at run time, but a kernel that needs
// scratch at all stalls its first dispatch in a process until the queue's scratch is allocated, which in the driver's form fell
// into the timed region of the skewed stream; block_count3 and eval_sorted_body's `opaque_zero` are what keeps them off it)
__global__ __launch_bounds__(BLOCK, (ITEMS <= 2 ? TC_LEAN_HOT_WAVES : TC_LEAN_HOT_WAVES4)) void k_eval_lean_hot(
    Params p, const uint64_t* __restrict__ sorted, uint32_t* __restrict__ loaded, uint32_t seq, uint32_t* hint, const HotEval* __restrict__ hep,
    uint32_t hot_blocks, uint32_t cold_grid) {
    if (blockIdx.x < hot_blocks) {
        eval_hot_role<FIXED>(p, *hep, hot_blocks);
        return;
    }
    // (a scalar: the loop's bounds and what the body derives from the part's length stay out of the vector registers)
    const uint32_t nc = __builtin_amdgcn_readfirstlane(hep->n[hep->ids]);
    for (uint32_t bid = blockIdx.x - hot_blocks; bid * (uint32_t)(BLOCK * ITEMS) < nc; bid += cold_grid) {
        uint32_t zero = 0u;
        asm volatile("" : "+v"(zero));
        eval_sorted_body<false, true, ITEMS, FIXED, true, BLOCK>(p, sorted, nullptr, nullptr, loaded, seq, nullptr, 0u, hint, bid, nc,
                                                                  (nc + BLOCK * ITEMS - 1) / (BLOCK * ITEMS), zero);
        __syncthreads(); // (the body's shared arrays are written again)
    }
}

// ---------------------------------------------------------------------------
// K2g: general batches (per-request now / quantity / rate) over the sorted batch.
// The update is not an associative scan (a denied request leaves the TAT, an
// allowed one moves it), but a DENIED request never changes the state, so a wave
// evaluates all requests of a segment piece against the piece's current state at
// once: everything before the first allowed request is final (denied against that
// state), the first allowed request is final and hands its new state to the lanes
// after it, repeat.  Iterations per wave = allowed requests in its longest piece
// + 1: a run of denials of a hot key costs one step per 64 requests, and a wave
// of 64 unrelated keys costs one step.  A segment that continues into the next
// wave hands (state, dirty) over through chain[]: the next wave (same or next
// block, dispatched no later than its successor) waits for it with its own
// request columns already loaded.  The last lane of a segment stores the cell.
// ---------------------------------------------------------------------------
struct __attribute__((aligned(32))) ChainRec {
    unsigned long long tat, expiry; // state the wave's last piece leaves (valid once fin == batch sequence number)
    uint32_t fin;                   // low half of the 8-byte flag word
    uint32_t spec;                  // == sequence number << 1 | strong: "my lanes are all denied if my segment still has its
                                    // resident state when it reaches me" (published before the wave waits); strong: "...
                                    // whatever state reaches me" (see k_eval_general)
    uint32_t dirty;
    uint32_t slot;                  // the key whose state this is (valid with fin): a wave that finds the record further back than its
                                    // direct predecessor knows by it whether the record still belongs to ITS segment
};

// (general batches, decisions only, TC_B_OUTPUTS_IDLE: the bytes were preset on the grouping stream like the lean
// kernel's -- the host sets F_PREFILLx only when `allowed` in request order is the one output of the batch)
template <bool FULL>
__device__ __forceinline__ void write_out_general(const Params& p, uint32_t i, const Req& r, const Decision& d) {
    if (!FULL && (p.flags & (F_PREFILL0 | F_PREFILL1)) != 0u) {
        const bool a = r.status == tc::ST_OK && d.allowed;
        if (!(p.flags & (a ? F_PREFILL1 : F_PREFILL0))) p.allowed[i] = a ? 1 : 0;
        return;
    }
    write_out(p, i, r, d);
}

// RUNS (round 6): the allowed-runs rule compiled in.  A stream whose keys are drained -- most requests denied, what a skewed
// stream looks like after its first batches -- never enters it, and without it the kernel needs fewer registers (one more wave
// per SIMD): the host picks the variant by what most decisions of a recent batch were (the fill hint), batch by batch; both
// variants give the same results (the rule only settles several allowed requests in one round).
template <bool FULL, bool RUNS = true>
__global__ __launch_bounds__(BLOCK) void k_eval_general(Params p, const uint64_t* __restrict__ sorted,
                                                        ChainRec* __restrict__ chain, uint32_t seq, uint32_t* hint) {
    const uint32_t n = p.n;
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t gw = k >> 6; // global wave number
    const bool valid = k < n;
    // (round 6: the element and the row's edge neighbour as two unconditional loads issued together, like the lean kernel's -- a
    // load under a branch waits for the loads before it)
    uint32_t ek = min(k, n - 1u);
    if (lane == 0 && ek > 0u) ek -= 1u;
    if (lane == 63) ek = min(ek + 1u, n - 1u);
    const uint64_t me_raw = sorted[min(k, n - 1u)];
    const uint32_t edge = (uint32_t)(sorted[ek] >> 32);
    const uint64_t me = valid ? me_raw : ~0ull;
    const uint32_t slot = (uint32_t)(me >> 32);
    const uint32_t idx = (uint32_t)me;
    const uint32_t orow = p.order ? k : idx;
    if (p.order && valid) p.order[k] = idx;
    uint32_t prev_slot = __shfl_up(slot, 1, 64), next_slot = __shfl_down(slot, 1, 64);
    if (lane == 0) prev_slot = edge;  // (k == 0: not looked at)
    if (lane == 63) next_slot = edge; // (k + 1 >= n: not looked at)
    const bool head = valid && (k == 0 || prev_slot != slot);
    const bool is_last = valid && (k + 1 == n || next_slot != slot);

    Req r;
    r.ei = r.dvt = r.q = r.now = r.limit = 0;
    r.status = tc::ST_INTERNAL;
    if (valid) r = make_req(p, idx, slot);
    const bool ok = valid && r.status == tc::ST_OK;

    // my piece = lanes [pstart, pend] of this wave that belong to my segment
    const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long hb = __ballot(head) & upto;
    const int pstart = hb ? 63 - __builtin_clzll(hb) : 0;
    const bool continued = valid && hb == 0ull; // my segment began in an earlier wave
    const unsigned long long lb = __ballot(is_last) >> lane;
    const int pend = lb ? lane + __builtin_ctzll(lb) : 63;
    const unsigned long long piece =
        (pend == 63 ? ~0ull : ((2ull << pend) - 1ull)) & ~((1ull << pstart) - 1ull);

    // State my piece starts from: the resident cell, or what the previous wave hands over.
    // A hot key's segment crosses hundreds of waves; waiting wave by wave would serialise
    // ~2 us hops (measured 3.5 ms for the Zipf head).  But a denied request leaves the state
    // alone, and a saturated key is denied almost always, so every wave of a continued segment
    // first checks its lanes against the segment's RESIDENT state c0 (every wave of the segment
    // can load it: the cell is only rewritten by the segment's last lane, which needs all of
    // them first).  "Nobody allowed under c0" is published at once (spec); a wave then looks
    // back over 64 predecessors per round trip: spec, spec, ..., fin(v) with v == c0 proves that
    // the state reaching it is c0.  Any mismatch falls back to the direct predecessor's record.
    Cell c;
    c.tat = 0;
    c.expiry = 0;
    bool dirty = false;
    bool strong_wave = false; // (wave-uniform) see below: this wave neither needs nor publishes the segment's state
    if (valid && slot < p.capacity) c = load_state_rt(p, slot, r.dvt); // continued lanes: c0, the guess
    if (__ballot(continued) != 0ull) { // wave-uniform: lane 0 is continued
        bool spec_allow = false, c0_dead = false;
        if (continued && ok) {
            Cell t0 = c;
            spec_allow = tc::gcra_step<false>(t0, r.ei, r.dvt, r.q, r.now).allowed;
            c0_dead = !(c.expiry > (uint64_t)r.now);
        }
        const bool transparent = __ballot(spec_allow) == 0ull;
        const bool through = __shfl((int)(continued && !is_last), 63, 64) != 0; // the segment runs through this whole wave
        // STRONG transparency (decisions only, TC_CFG_FIXED_PARAMS): under the 8-byte layout a key's plan never changes and
        // expiry == tat + dvt, so within a batch the key's TAT only grows and a state that is live at `now` stays live.
        // A request that finds c0 LIVE and is denied under c0 is therefore denied under every state the segment can be
        // in when it reaches this wave (allow_at = max(tat, now - dvt) + inc - dvt grows with tat): the wave's decisions
        // are final without knowing the incoming state, and it hands on whatever state reaches it.  Such a wave waits
        // for nobody and publishes no state; later waves look THROUGH it to the nearest wave that did.  (A request
        // that finds c0 expired is judged as a fresh key, which can be stricter than a live entry: not strong.)
        strong_wave = !FULL && (p.flags & F_FIXED) != 0u && transparent && through && __ballot(continued && ok && c0_dead) == 0ull;
        if (transparent && through && lane == 0)
            __hip_atomic_store(&chain[gw].spec, (seq << 1) | (strong_wave ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long c0_tat = __shfl((long long)c.tat, 0, 64);
        const unsigned long long c0_exp = __shfl((unsigned long long)c.expiry, 0, 64);
        long long in_tat = c0_tat;
        unsigned long long in_exp = c0_exp;
        uint32_t in_dirty = 0;
        bool direct = false;    // speculation failed: only an exact hand-over will do
        bool skipped_weak = false; // a weakly transparent wave lies between me and the records being examined
        bool walking = false;      // the search has passed open records: whatever it finds is an EARLIER state, not the one that reaches me
        uint32_t base = 0;      // records gw-1-base-lane are examined
        // EARLIER-STATE transparency (round 4; decisions only, TC_CFG_FIXED_PARAMS, the segment runs through this whole wave): the
        // argument that makes a wave strongly transparent under the RESIDENT state holds for ANY earlier state S of its own
        // segment -- the key's TAT only grows inside a batch and a live state stays live, so a request that finds S live and is
        // denied under S is denied under whatever state reaches the wave.  While the records between a wave and the nearest
        // published state of its segment (same slot) are still open, the wave tries that state: if it denies every lane, the
        // wave is final, says so (strong) and waits for nobody.  A key that enters a batch fresh -- resident state vacant, which
        // no wave can be transparent under -- and is drained by its first waves no longer chains the hundreds of waves behind
        // them one by one (the first Zipf batch on an empty table: one key, 1 800 waves, 4.7 ms).
        const bool may_try_earlier = !FULL && (p.flags & (F_FIXED | F_NO_EARLIER)) == F_FIXED && through;
        constexpr uint32_t WALK_LIMIT = 48u * 64u; // waves a search for an earlier state walks back before it starts over
        tc::SpinGuard guard;
        while (!strong_wave) {
            if (tc::spin_expired(guard)) { // (see tc::SpinGuard: flagged, never hung; this wave goes on from c0)
                if (lane == 0) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]);
                in_tat = c0_tat;
                in_exp = c0_exp;
                break;
            }
            const long long j = (long long)gw - 1 - (long long)base - lane;
            unsigned long long fl = 0ull;
            if (j >= 0)
                fl = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&chain[j].fin), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t sp = (uint32_t)(fl >> 32);
            const bool is_fin = (uint32_t)fl == seq, is_spec = (sp >> 1) == seq, is_strong = is_spec && (sp & 1u);
            const unsigned long long fm = __ballot(is_fin), sm = __ballot(is_spec), tm = __ballot(is_strong);
            // the nearest record that carries a state, and what lies between it and me
            int d = -1;
            bool exact = false, earlier = false;
            if (fm) {
                const int f = __builtin_ctzll(fm);
                const unsigned long long below = f ? ((1ull << f) - 1ull) : 0ull;
                if ((~tm & below) == 0ull && !skipped_weak && !walking) {
                    d = f;        // only strongly transparent waves in between: that state IS the one that reaches me
                    exact = true;
                } else if (!direct && !walking && (~sm & below) == 0ull) {
                    d = f;        // transparent if the state is still c0: to be checked
                } else if (may_try_earlier) {
                    d = f;        // an earlier state of (maybe) my segment: to be tried
                    earlier = true;
                }
            } else if (~tm == 0ull && !walking) {
                base += 64; // 64 strongly transparent waves: look further back
                continue;
            } else if (!direct && !walking && ~sm == 0ull) {
                skipped_weak = true;
                base += 64; // 64 transparent waves: look further back
                continue;
            } else if (may_try_earlier && base + 64u <= WALK_LIMIT && (long long)gw - 1 - (long long)base - 63 > 0) {
                walking = true; // open records in this window: walk on, looking for ANY state of my segment
                base += 64;
                continue;
            }
            if (d < 0) {
                base = 0; // something in between is not ready: look again from the nearest record
                skipped_weak = false;
                walking = false;
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            long long vt = 0;
            unsigned long long vx = 0;
            uint32_t vd = 0, vs = 0;
            if (lane == d) {
                vt = (long long)__hip_atomic_load(&chain[j].tat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vx = __hip_atomic_load(&chain[j].expiry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vd = __hip_atomic_load(&chain[j].dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vs = __hip_atomic_load(&chain[j].slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            vt = __shfl(vt, d, 64);
            vx = __shfl(vx, d, 64);
            vd = __shfl(vd, d, 64);
            vs = __shfl(vs, d, 64);
            if (earlier) {
                // (every lane of this wave is a request of one slot: `through`)
                bool final_here = false;
                if (vs == __shfl(slot, 0, 64)) {
                    bool allow_s = false, dead_s = false;
                    if (ok) {
                        Cell ts;
                        ts.tat = vt;
                        ts.expiry = vx;
                        dead_s = !(vx > (uint64_t)r.now);
                        allow_s = tc::gcra_step<false>(ts, r.ei, r.dvt, r.q, r.now).allowed;
                    }
                    final_here = __ballot(allow_s || dead_s) == 0ull;
                }
                if (final_here) {
                    strong_wave = true; // denied under an earlier LIVE state of my segment: denied under the one that reaches me
                    in_tat = vt;        // (the lanes are judged against it below: all denied)
                    in_exp = vx;
                    if (lane == 0)
                        __hip_atomic_store(&chain[gw].spec, (seq << 1) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                base = 0; // not (yet) decisive: wait for the records in between like everybody else
                skipped_weak = false;
                walking = false;
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            if (exact || (vt == c0_tat && vx == c0_exp)) { // (the weakly transparent waves in between were right)
                in_tat = vt;
                in_exp = vx;
                in_dirty = vd;
                break;
            }
            direct = true; // the state changed on the way: no shortcut through weakly transparent waves
            base = 0;
            skipped_weak = false;
        }
        if (continued) {
            c.tat = in_tat;
            c.expiry = in_exp;
            dirty = in_dirty != 0u;
        }
    }

    uint32_t na = 0, nd = 0, ne = 0;
    bool fin = !valid, was_allowed = false;
    Cell mine; // state after my request, if it was allowed
    mine.tat = 0;
    mine.expiry = 0;
    if (valid && !ok) { // errors leave the state alone (rate_limiter.rs:111-117)
        Decision z;
        z.allowed = false;
        z.remaining = z.reset_after = z.retry_after = 0;
        write_out_general<FULL>(p, orow, r, z);
        ne = 1;
        fin = true;
    }
    while (true) {
        Cell c2 = c;
        bool allow = false;
        if (!fin) allow = tc::gcra_step<false>(c2, r.ei, r.dvt, r.q, r.now).allowed;
        const unsigned long long open = __ballot(!fin);
        if (open == 0ull) break;
        const unsigned long long ap = __ballot(allow) & piece;
        const int first = ap ? __builtin_ctzll(ap) : 64;
        if (!fin && lane <= first) {
            // lanes before the first allowed request are denied against the current state,
            // the first allowed request is allowed against it: both final
            Decision d;
            if (FULL) {
                Cell tmp = c;
                d = tc::gcra_step<true>(tmp, r.ei, r.dvt, r.q, r.now);
            } else {
                d.allowed = allow;
                d.remaining = d.reset_after = d.retry_after = 0;
            }
            write_out_general<FULL>(p, orow, r, d);
            if (lane == first) {
                na = 1;
                was_allowed = true;
                mine = c2;
            } else {
                nd = 1;
            }
            fin = true;
        }
        // the lanes after `first` in its piece continue from the state it leaves
        const int src = first < 64 ? first : lane;
        const long long nt = __shfl((long long)c2.tat, src, 64);
        const unsigned long long nx = __shfl((unsigned long long)c2.expiry, src, 64);
        if (first < 64 && lane > first) { // (lanes beyond pend are masked out by `piece` in `ap`: first is in MY piece)
            c.tat = nt;
            c.expiry = nx;
            dirty = true;
        }
        // Round 5 -- a RUN of allowed requests in one round.  One round per allowed request is what a key costs while it still has
        // tokens: a piece of 60 allowed requests keeps its wave for 60 rounds, and the first ~25 batches of a Zipf stream -- all a
        // 20-batch measurement sees -- are full of such pieces (98 us per batch there, 74 once the keys are
        // drained).  If the requests behind the one just allowed carry the same increment and tolerance and are all allowed as well,
        // the state in front of the k-th of them is the state just left + k increments: every open lane runs the REAL step on
        // that predicted state and checks that it is allowed and leaves exactly the next predicted state (the clamp was a no-op,
        // the entry live, nothing saturated); the run ends in front of the first lane that fails, and everything before it is
        // final -- exact by induction, no special cases.  (Round 3's general version -- a max-plus prefix scan in every round --
        // cost more registers and instructions than it saved once the keys were drained; this one is an arithmetic progression,
        // and it is only entered when some piece still has open lanes behind an allowed request: wave-uniform.)
        if (RUNS && (p.flags & F_GENERAL_RUNS) != 0u && __ballot(!fin && first < 64) != 0ull) {
            const long long inc_me = tc::sat_mul(r.ei, r.q);
            const long long inc_f = __shfl(inc_me, src, 64), dvt_f = __shfl((long long)r.dvt, src, 64);
            const bool cand = !fin && first < 64; // (an open lane behind the allowed request of its piece)
            const unsigned long long behind = __ballot(cand) & piece & ~((2ull << (first & 63)) - 1ull); // ... of MY piece
            const uint32_t kk = (uint32_t)__popcll(behind & ((1ull << lane) - 1ull)); // open lanes of my piece between `first` and me
            Cell sk;
            sk.tat = nt + (long long)kk * inc_f;
            sk.expiry = (unsigned long long)(sk.tat + dvt_f);
            const long long want_tat = sk.tat + inc_f;
            bool good = cand && inc_me == inc_f && r.dvt == dvt_f && inc_f > 0 && inc_f < (1ll << 40) && nt > -(1ll << 61) && nt < (1ll << 61) &&
                        dvt_f >= 0 && dvt_f < (1ll << 61) && nx == (unsigned long long)(nt + dvt_f);
            Cell s2 = sk;
            if (good) {
                good = tc::gcra_step<false>(s2, r.ei, r.dvt, r.q, r.now).allowed && s2.tat == want_tat && s2.expiry == (unsigned long long)(want_tat + dvt_f);
            }
            const unsigned long long bad = __ballot(cand && !good) & behind;
            const int stop = bad ? __builtin_ctzll(bad) : 64; // the first lane of my piece the progression does not hold for
            const unsigned long long run = stop < 64 ? (behind & ((1ull << stop) - 1ull)) : behind;
            if (cand && ((run >> lane) & 1ull)) { // allowed against the predicted state: final
                Decision d;
                if (FULL) {
                    Cell tmp = sk;
                    d = tc::gcra_step<true>(tmp, r.ei, r.dvt, r.q, r.now);
                } else {
                    d.allowed = true;
                    d.remaining = d.reset_after = d.retry_after = 0;
                }
                write_out_general<FULL>(p, orow, r, d);
                na = 1;
                was_allowed = true;
                mine = s2;
                fin = true;
            } else if (first < 64 && lane > first && run != 0ull) {
                // behind the run: go on from the state it leaves.  (Every lane of the piece behind `first`, settled or not: an
                // error request at the end of a piece owns the state the piece hands on, like any last lane.)
                c.tat = nt + (long long)__popcll(run) * inc_f;
                c.expiry = (unsigned long long)(c.tat + dvt_f);
            }
        }
    }
    // the last lane of the piece owns the state the piece leaves
    if (valid && lane == pend && !strong_wave) { // (a strongly transparent wave hands on a state it never learnt: nothing to publish)
        const Cell out = was_allowed ? mine : c;
        const bool out_dirty = dirty || was_allowed;
        if (is_last) {
            if (out_dirty && slot < p.capacity) store_state_rt(p, slot, out);
        } else {
            ChainRec* o = &chain[gw];
            __hip_atomic_store(&o->tat, (unsigned long long)out.tat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&o->expiry, (unsigned long long)out.expiry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&o->dirty, out_dirty ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&o->slot, slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the record must be performed at agent scope before the flag that announces it
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(&o->fin, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wave_denied_add(p, slot, nd != 0);
    block_count3(na, nd, ne, p.counters, blockIdx.x == gridDim.x / 2 ? hint : nullptr);
}

static __global__ __launch_bounds__(BLOCK) void k_commit_list(const PendEntry* __restrict__ pend,
                                                       uint32_t* __restrict__ pend_count, Cell* __restrict__ cells,
                                                       int64_t* __restrict__ tat8) {
    // pend_count[0] = entries, pend_count[1] = blocks of this launch that are done
    const uint32_t cnt = pend_count[0];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < cnt; i += gridDim.x * BLOCK) {
        const PendEntry pe = pend[i];
        if (tat8) tat8[pe.slot] = pe.cell.tat;
        else tc::store_cell(&cells[pe.slot], pe.cell);
    }
    __syncthreads(); // every lane of this block has consumed `cnt`
    if (threadIdx.x == 0) {
        // the last block to finish re-arms the list for the next batch (no extra launch)
        if (atomicAdd(&pend_count[1], 1u) == gridDim.x - 1) {
            pend_count[0] = 0;
            pend_count[1] = 0;
        }
    }
}

static __global__ __launch_bounds__(NSHARD) void k_fold_counters(unsigned long long* counters) {
    __shared__ unsigned long long s[3][NSHARD / 64];
    const unsigned long long* shard = counters + (TC_CNT_COUNT + 1) + threadIdx.x * SHARD_WORDS;
    unsigned long long v[3] = {shard[0], shard[1], shard[2]};
    for (int off = 32; off > 0; off >>= 1)
        for (int j = 0; j < 3; ++j) v[j] += __shfl_down(v[j], off, 64);
    if ((threadIdx.x & 63) == 0)
        for (int j = 0; j < 3; ++j) s[j][threadIdx.x >> 6] = v[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t[3] = {0, 0, 0};
        for (int j = 0; j < 3; ++j)
            for (int w = 0; w < NSHARD / 64; ++w) t[j] += s[j][w];
        counters[TC_CNT_ALLOWED] = t[0];
        counters[TC_CNT_DENIED] = t[1];
        counters[TC_CNT_ERRORS] = t[2];
        counters[TC_CNT_TOTAL] = t[0] + t[1] + t[2];
    }
}

// allowed[] bytes -> bitmask (wavefront ballot, one u64 per wave)
static __global__ __launch_bounds__(BLOCK) void k_pack_bits(const uint8_t* __restrict__ allowed, uint32_t n,
                                                     uint64_t* __restrict__ bits) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool a = i < n && allowed[i];
    const unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0 && i < n) bits[i >> 6] = m;
}

} // namespace ev
