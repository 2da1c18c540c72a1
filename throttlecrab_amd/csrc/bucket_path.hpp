// bucket_path.hpp -- grouping WITHOUT a sort for uniform batches (one `now`, one `quantity`, rate per
// slot or scalar: BASELINE configs 1-4), gfx950.
//
// What the reference's sequence needs from a batch (rate_limiter.rs:102-250 applied in index order) is,
// per request, its RANK among the requests of the same key -- not a sorted batch.  So instead of three
// global radix passes + an evaluation over the sorted copy:
//   k_tile_hist    per 4096-request tile, how many requests fall into each BUCKET of 2^lb consecutive slots
//   k_bucket_scan  exclusive prefix of those counts over the tiles (per bucket) + bucket totals
//   k_scatter      stable partition of the batch by bucket: element = low slot bits << 20 | request index,
//                  written to the bucket's range in request order  (ONE pass over the batch, 4 B per request)
//   k_bucket_eval  ONE WAVE PER BUCKET: counts the bucket's requests per slot in an LDS table, walks
//                  them once more in reverse index order taking every request's rank off the table
//                  (conflicts inside a 64-lane step are resolved with ballots), and evaluates the closed
//                  form of gcra_math.hpp against the cells gathered in the meantime.  A bucket's cells
//                  are touched by exactly one wave, so there is no cross-wave protocol at all: every
//                  cell load of the wave is complete before its first store.
// No kernel of this path waits for another block (no look-back, no spinning, no dispatch-order
// assumption).  Eligibility is decided on the host (uniform batch, every run provably regular --
// all_runs_regular() --, n <= 2^20, <= MAX_BUCKETS buckets); everything else takes the sort path.
// A bucket of ANY size is evaluated correctly (a long one is walked in pieces with its cell stores
// parked until the bucket's last load is done), only slowly: k_scatter reports the largest bucket
// of every batch and the engine sends skewed streams (Zipf: one key = 11 % of the batch) down the
// sort path instead.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eval_kernels.hpp"
#include "gcra_math.hpp"

namespace bp {

using ev::Params;
using ev::PendEntry;
using ev::Req;
using tc::Cell;
using tc::Decision;
using tc::RateClass;

constexpr int LB_MAX = 12;                     // log2(slots per bucket) <= 12: element = low bits << 20 | request index
constexpr uint32_t IDX_BITS = 20;
constexpr uint32_t IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr uint32_t MAX_N = 1u << IDX_BITS;     // largest batch this path takes
constexpr uint32_t MAX_BUCKETS = 8192;
constexpr int TILE_THREADS = 256;
constexpr int TILE_ITEMS = 16;
constexpr uint32_t TILE = TILE_THREADS * TILE_ITEMS; // 4096 requests
constexpr int SCAN_THREADS = 1024;             // k_bucket_scan: 16 waves x 64 buckets
__host__ __device__ inline uint32_t scan_blocks(uint32_t nbk) { return (nbk >> 6) + 1u; } // the end marker `nbk` has a lane too
// k_bucket_eval's table entry (u16): requests of the slot not yet walked | TOUCHED.  The gate bounds a
// bucket by `skew` <= MAX_SKEW requests, so 15 bits are enough.
constexpr uint32_t CNT_MASK = 0x7FFFu, TOUCHED = 0x8000u, MAX_SKEW = 0x7FFFu;

// the clamp value `capacity` (every out-of-range slot) has a bucket too
__host__ __device__ inline uint32_t buckets_of(uint64_t capacity, int lb) { return (uint32_t)(capacity >> lb) + 1u; }
__host__ __device__ inline uint32_t tiles_of(uint32_t n) { return (n + TILE - 1) / TILE; }
// Bucket width for an engine: about 256 requests per bucket when a full batch is spread evenly over the
// key space (4 steps of one wave; 10 M keys, 1 Mi requests: 2048 slots), at most MAX_BUCKETS buckets.
// Returns -1 if the key space is too large for this path.
inline int pick_lb(uint64_t capacity, uint64_t max_n) {
    int lb = 6;
    while (lb < LB_MAX && ((uint64_t)2 << lb) * max_n <= 256 * capacity) ++lb;
    while (lb < LB_MAX && buckets_of(capacity, lb) > MAX_BUCKETS) ++lb;
    return buckets_of(capacity, lb) <= MAX_BUCKETS ? lb : -1;
}

// scratch of one grouping (one per scratch set of the engine's ring)
struct Work {
    uint16_t* tile_hist; // [tiles][nbk]   requests of tile t in bucket j
    uint32_t* tile_off;  // [tiles][nbk]   ... in the tiles before t
    uint32_t* local;     // [nbk + 64]     start of a bucket's range in elems, relative to its group of 64 buckets
    uint32_t* base;      // [nbk / 64 + 2] start of each group of 64 buckets (k_bucket_scan adds a group's total to every
                         //                later group's word; zeroed by k_tile_hist)
    uint32_t* elems;     // [n]
    uint32_t* maxb;      // [1]            largest bucket of this batch: THE GATE.  Written by k_bucket_scan; the rest of
                         //                this path runs iff it is <= `skew`, the sort path (enqueued as well) iff not
    uint32_t skew;       //                longest bucket this path takes on
    uint32_t nbk;
    int lb;
    volatile uint32_t* gate_host; // pinned host word or NULL: k_scatter mirrors the gate there, so that the host -- which
                         //                cannot wait for it -- at least learns what recent batches looked like
};
inline size_t work_bytes(uint32_t max_n, uint32_t nbk) {
    const size_t tiles = tiles_of(max_n);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return up(tiles * nbk * 2) + up(tiles * nbk * 4) + up(((size_t)nbk + 64) * 4) + up(((size_t)nbk / 64 + 2) * 4) + up((size_t)max_n * 4) + 256;
}
inline Work carve(void* base, uint32_t max_n, uint32_t nbk, int lb) {
    const size_t tiles = tiles_of(max_n);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    uint8_t* p = static_cast<uint8_t*>(base);
    Work w;
    w.tile_hist = reinterpret_cast<uint16_t*>(p);
    p += up(tiles * nbk * 2);
    w.tile_off = reinterpret_cast<uint32_t*>(p);
    p += up(tiles * nbk * 4);
    w.local = reinterpret_cast<uint32_t*>(p);
    p += up(((size_t)nbk + 64) * 4);
    w.base = reinterpret_cast<uint32_t*>(p);
    p += up(((size_t)nbk / 64 + 2) * 4);
    w.elems = reinterpret_cast<uint32_t*>(p);
    p += up((size_t)max_n * 4);
    w.maxb = reinterpret_cast<uint32_t*>(p);
    w.nbk = nbk;
    w.lb = lb;
    w.skew = 1024;
    w.gate_host = nullptr;
    return w;
}
// where bucket j's requests start in elems (j == nbk: the end of the last bucket)
__device__ __forceinline__ uint32_t bucket_start(const Work& w, uint32_t j) { return w.base[j >> 6] + w.local[j];
}

__device__ __forceinline__ uint32_t bucket_of(uint32_t slot, uint32_t cap, int lb) { return (slot < cap ? slot : cap) >> lb; }

// ---------------------------------------------------------------------------
// Groups of equal keys inside one 64-lane step, in lane (= request) order.  Every active lane writes its
// lane number to mark[v] and reads it back: a lane that reads another lane's number shares its key with
// somebody, and each such group is then settled with one ballot (one loop turn per GROUP, not per lane).
//   g      lanes of the wave holding my key (>= 1)
//   below  ... of them with a lower lane number (= earlier requests)
//   top    nobody with my key has a higher lane number
// `mark` needs no initialisation (a lane only ever compares against values written in this very step), and
// it may be smaller than the key space: `mi` is any function of the key (two keys sharing a mark byte only
// cost a loop turn: every real group still has a member that reads somebody else's number, and the
// ballot compares the keys themselves).  Must be reached by the whole wave.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void settle_groups(bool active, uint32_t v, uint32_t mi, uint8_t* __restrict__ mark, int lane, uint32_t& g,
                                              uint32_t& below, bool& top) {
    if (active) mark[mi] = (uint8_t)lane;
    __builtin_amdgcn_wave_barrier(); // (same-wave LDS operations execute in program order)
    const uint32_t seen = active ? (uint32_t)mark[mi] : (uint32_t)lane;
    g = 1;
    below = 0;
    top = true;
    unsigned long long pending = __ballot(active && seen != (uint32_t)lane);
    while (pending) {
        const int l = __builtin_ctzll(pending);
        const uint32_t vl = __shfl(v, l, 64);
        const bool mine = active && v == vl;
        const unsigned long long mm = __ballot(mine);
        if (mine) {
            g = (uint32_t)__popcll(mm);
            below = (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
            top = (mm >> lane) == 1ull;
        }
        pending &= ~mm;
    }
}

// ---------------------------------------------------------------------------
// K1: requests per bucket, per tile
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(TILE_THREADS) void k_tile_hist(const uint32_t* __restrict__ slot, uint32_t n, uint32_t cap, Work w) {
    extern __shared__ uint32_t s_h[]; // [nbk]
    const uint32_t nbk = w.nbk;
    const uint32_t base = blockIdx.x * TILE + threadIdx.x;
    uint32_t s[TILE_ITEMS];
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j) {
        const uint32_t pos = base + j * TILE_THREADS;
        s[j] = pos < n ? slot[pos] : 0xFFFFFFFFu;
    }
    for (uint32_t i = threadIdx.x; i < nbk; i += TILE_THREADS) s_h[i] = 0;
    if (blockIdx.x == 0) { // k_bucket_scan (next on this stream) accumulates into these
        if (threadIdx.x == 0) *w.maxb = 0u;
        for (uint32_t i = threadIdx.x; i < nbk / 64 + 2; i += TILE_THREADS) w.base[i] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j)
        if (base + j * TILE_THREADS < n) atomicAdd(&s_h[bucket_of(s[j], cap, w.lb)], 1u);
    __syncthreads();
    uint16_t* row = w.tile_hist + (size_t)blockIdx.x * nbk;
    for (uint32_t i = threadIdx.x; i < nbk; i += TILE_THREADS) row[i] = (uint16_t)s_h[i]; // <= 4096
}

// ---------------------------------------------------------------------------
// K2: per bucket, exclusive prefix of the tile counts over the tiles + the bucket's total.
// Block = 16 waves x 64 consecutive buckets; wave k owns a contiguous run of tiles.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(SCAN_THREADS) void k_bucket_scan(Work w, uint32_t tiles) {
    __shared__ uint32_t s_tot[SCAN_THREADS / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nbk = w.nbk;
    const uint32_t j = blockIdx.x * 64 + lane;
    const bool ok = j < nbk;
    const uint32_t per = (tiles + (SCAN_THREADS / 64) - 1) / (SCAN_THREADS / 64);
    const uint32_t t0 = min(tiles, (uint32_t)wave * per), t1 = min(tiles, t0 + per);
    uint32_t run = 0;
    for (uint32_t t = t0; t < t1; t += 16) {
        uint32_t c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (ok && t + u < t1) ? (uint32_t)w.tile_hist[(size_t)(t + u) * nbk + j] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) run += c[u];
    }
    s_tot[wave][lane] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        const uint32_t v = s_tot[k][lane];
        if (k < wave) before += v;
        total += v;
    }
    run = before;
    for (uint32_t t = t0; t < t1; t += 16) {
        uint32_t c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (ok && t + u < t1) ? (uint32_t)w.tile_hist[(size_t)(t + u) * nbk + j] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (ok && t + u < t1) w.tile_off[(size_t)(t + u) * nbk + j] = run;
            run += c[u];
        }
    }
    if (wave == 0) {
        // bucket starts: prefix inside my group of 64 buckets + the group's total added to every later group's base
        const uint32_t tot = ok ? total : 0u;
        uint32_t v = tot, mx = tot;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v += o;
        }
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_down(mx, off, 64));
        w.local[j] = v - tot; // (entries up to nbk + 63 exist: bucket_start(nbk) is the end of the last bucket)
        const uint32_t group_total = __shfl(v, 63, 64);
        const uint32_t groups = (nbk >> 6) + 1; // groups holding a bucket or the end marker nbk
        if (group_total)
            for (uint32_t k = blockIdx.x + 1 + lane; k < groups; k += 64) atomicAdd(&w.base[k], group_total);
        if (lane == 0 && mx) atomicMax(w.maxb, mx);
    }
}

// ---------------------------------------------------------------------------
// K3: stable partition by bucket.  Wave k of a tile owns the tile's k-th strip of 1024 requests and walks
// it in 16 steps of 64 (request order); the step's lanes are ranked inside their bucket with
// settle_groups(), per-wave counters carry the rank from step to step, a prefix over the four waves, the
// tile's offsets and the bucket starts (K2) give the position.
// dynamic LDS: s_cnt u16[4][nbkp] | s_mark u8[4][SCATTER_MARKS]   (47 KB at 4883 buckets: leaves room for
// the evaluation kernel of an earlier batch on the same CU)
// ---------------------------------------------------------------------------
__host__ __device__ inline uint32_t pad4(uint32_t v) { return (v + 3u) & ~3u; }
constexpr uint32_t SCATTER_MARKS = 2048; // mark bytes per wave (hashed by the low bucket bits)
inline size_t scatter_lds_bytes(uint32_t nbk) { return (size_t)pad4(nbk) * 4 * 2 + (size_t)SCATTER_MARKS * 4; }

static __global__ __launch_bounds__(TILE_THREADS) void k_scatter(const uint32_t* __restrict__ slot, uint32_t n, uint32_t cap, Work w) {
    extern __shared__ uint32_t s_mem[];
    const uint32_t longest = __builtin_nontemporal_load(w.maxb);
    if (w.gate_host && blockIdx.x == 0 && threadIdx.x == 0) *w.gate_host = longest;
    if (longest > w.skew) return; // a skewed batch: the sort path (enqueued as well) takes it
    const uint32_t nbk = w.nbk, nbkp = pad4(nbk);
    const int lb = w.lb;
    uint16_t* s_cnt = reinterpret_cast<uint16_t*>(s_mem);
    uint8_t* s_mark = reinterpret_cast<uint8_t*>(s_cnt + 4 * nbkp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;

    uint32_t s[TILE_ITEMS];
    const uint32_t first = tile * TILE + wave * (64 * TILE_ITEMS) + lane;
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j) {
        const uint32_t pos = first + j * 64;
        s[j] = pos < n ? slot[pos] : 0xFFFFFFFFu;
    }
    for (uint32_t i = threadIdx.x; i < 4 * nbkp / 2; i += TILE_THREADS) reinterpret_cast<uint32_t*>(s_cnt)[i] = 0;
    __syncthreads();

    uint16_t* cnt = s_cnt + wave * nbkp;
    uint8_t* mark = s_mark + wave * SCATTER_MARKS;
    uint32_t rank[TILE_ITEMS];
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j) {
        const bool valid = first + j * 64 < n;
        const uint32_t b = valid ? bucket_of(s[j], cap, lb) : 0u;
        const uint32_t old = valid ? (uint32_t)cnt[b] : 0u;
        uint32_t g, below;
        bool top;
        settle_groups(valid, b, b & (SCATTER_MARKS - 1u), mark, lane, g, below, top);
        rank[j] = old + below;
        if (valid && below == 0) cnt[b] = (uint16_t)(old + g);
        __builtin_amdgcn_wave_barrier();
    }
    // where my requests' buckets start for this tile (three small arrays that stay in L2), requested
    // before the barrier so that they are on their way while the block settles
    uint32_t at[TILE_ITEMS];
    const uint32_t* off_row = w.tile_off + (size_t)tile * nbk;
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j) {
        const uint32_t b = bucket_of(s[j], cap, lb);
        at[j] = (first + j * 64 < n) ? w.base[b >> 6] + w.local[b] + off_row[b] : 0u;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nbk; j += TILE_THREADS) { // counts -> exclusive prefix over the waves
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = s_cnt[k * nbkp + j];
            s_cnt[k * nbkp + j] = (uint16_t)run;
            run += c;
        }
    }
    __syncthreads();
    const uint32_t low = (1u << lb) - 1u;
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; ++j) {
        const uint32_t pos = first + j * 64;
        if (pos < n) {
            const uint32_t sl = s[j] < cap ? s[j] : cap;
            w.elems[at[j] + cnt[sl >> lb] + rank[j]] = ((sl & low) << IDX_BITS) | pos;
        }
    }
}

// ---------------------------------------------------------------------------
// K4: one wave per bucket
// ---------------------------------------------------------------------------
// decision counters of a one-wave block
__device__ __forceinline__ void wave_count3(uint32_t a, uint32_t b, uint32_t c, unsigned long long* counters) {
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if (threadIdx.x == 0) {
        unsigned long long* shard = counters + (TC_CNT_COUNT + 1) + (blockIdx.x % ev::NSHARD) * ev::SHARD_WORDS;
        if (a) atomicAdd(&shard[0], (unsigned long long)a);
        if (b) atomicAdd(&shard[1], (unsigned long long)b);
        if (c) atomicAdd(&shard[2], (unsigned long long)c);
    }
}

struct Tally {
    uint32_t na, nd, ne;
};

// pass A: one more request for slot v (u16 counters, two to a word: a 32-bit LDS atomic on the half's word;
// a bucket holds at most MAX_SKEW requests, so the low half never carries into the high one)
__device__ __forceinline__ void count_slot(uint16_t* __restrict__ table, uint32_t v) {
    atomicAdd(reinterpret_cast<uint32_t*>(table) + (v >> 1), 1u << (16u * (v & 1u)));
}

// One 64-lane step of a bucket, LATEST requests first: rank off the table, closed-form evaluation, outputs.
// Reached by the whole wave.
//   e / raw       the lane's element, its slot's resident state as loaded
//   rq0           the batch-wide request (BYSLOT: rebuilt from the slot's own rate class `rc`)
//   park          nullptr: the owner of a slot's new state stores it;  else: parks it at park[parked ...]
template <bool FULL, bool FIXED, bool BYSLOT>
__device__ __forceinline__ void eval_step(const Params& p, const Req& rq0, uint32_t bucket, int lb, bool valid, uint32_t e,
                                          Cell raw, const RateClass& rc, uint16_t* __restrict__ table, uint8_t* __restrict__ mark,
                                          int lane, Tally& t, PendEntry* __restrict__ park, uint32_t& parked) {
    const uint32_t v = e >> IDX_BITS, idx = e & IDX_MASK;
    const uint32_t slot = (bucket << lb) | v;
    const bool has_cell = valid && slot < p.capacity;
    const uint32_t cur = has_cell ? (uint32_t)table[v] : 0u;
    uint32_t g, below;
    bool top;
    settle_groups(has_cell, v, v, mark, lane, g, below, top);
    const uint32_t left = cur & CNT_MASK;         // requests of my slot not yet walked (mine included)
    const uint32_t r = left - g + below;          // my rank among the slot's requests of this batch
    const bool is_last = top && !(cur & TOUCHED); // nobody after me in the whole batch
    if (has_cell && below == 0) table[v] = (uint16_t)((left - g) | TOUCHED); // (a 2-byte store: the neighbour's half is not touched)
    __builtin_amdgcn_wave_barrier();
    bool writer = false;
    Cell wcell;
    wcell.tat = 0;
    wcell.expiry = 0;
    if (valid) {
        const uint32_t orow = idx; // (grouped output rows, p.order, are the sort path's)
        Req rq = rq0;
        if (BYSLOT) rq = ev::make_req_rc(p, slot, rc);
        else if (slot >= p.capacity) rq.status = tc::ST_INTERNAL;
        Decision d;
        d.allowed = false;
        d.remaining = d.reset_after = d.retry_after = 0;
        bool denied_here = false;
        if (rq.status != tc::ST_OK) {
            t.ne += 1;
            ev::write_out(p, orow, rq, d);
        } else {
            Cell c = FIXED ? tc::fixed_cell(raw.tat, rq.dvt) : raw;
            const Decision d0 = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now); // c = state after request 0
            if (!d0.allowed) {
                // request 0 denied => state untouched => every request of the run equals request 0
                t.nd += 1;
                denied_here = true;
                ev::write_out(p, orow, rq, d0);
            } else if (r == 0 && is_last) {
                t.na += 1; // a key requested once in this batch
                ev::write_out(p, orow, rq, d0);
                writer = true;
                wcell = c;
            } else if (FULL) {
                const tc::RunForm f = tc::run_form(c, rq.ei, rq.dvt, rq.q, rq.now);
                if (!f.regular) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]); // the host's proof was wrong: must stay 0
                if (r == 0) {
                    t.na += 1;
                    ev::write_out(p, orow, rq, d0);
                    writer = f.n_tot == 1;
                    wcell = c;
                } else {
                    const int64_t jj = (int64_t)r < f.n_tot ? (int64_t)r : f.n_tot;
                    Cell s;
                    s.tat = f.new0 + (jj - 1) * f.inc; // state my r predecessors leave behind
                    s.expiry = UINT64_MAX;
                    d = tc::gcra_step<true>(s, rq.ei, rq.dvt, rq.q, rq.now);
                    t.na += d.allowed;
                    t.nd += !d.allowed;
                    denied_here = !d.allowed;
                    ev::write_out(p, orow, rq, d);
                    writer = d.allowed && (is_last || (int64_t)r + 1 == f.n_tot);
                    wcell = s;
                }
            } else {
                // decisions only: rank r is allowed <=> r * inc <= room (no 64-bit division)
                const tc::RunLite f = tc::run_lite(c, rq.ei, rq.dvt, rq.q, rq.now);
                if (!f.regular) tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]);
                const bool ok_r = r == 0 || tc::rank_allowed(f, r);
                d.allowed = ok_r;
                t.na += ok_r;
                t.nd += !ok_r;
                denied_here = !ok_r;
                ev::write_out(p, orow, rq, d);
                if (ok_r && (is_last || !tc::rank_allowed(f, r + 1u))) {
                    writer = true;
                    wcell = r == 0 ? c : tc::cell_after(f.new0 + (int64_t)r * f.inc, rq.dvt, rq.now);
                }
            }
            if (FIXED && writer && wcell.expiry != (uint64_t)(wcell.tat + rq.dvt))
                tc::invariant_failed(&p.counters[(TC_CNT_COUNT + 1) + 3]); // the 8-byte layout would lose information: must stay 0
        }
        if (p.denied && denied_here) atomicAdd(&p.denied[slot], 1u); // (skewed streams take the sort path: one atomic per key run there)
    }
    if (park) {
        const unsigned long long wm = __ballot(writer);
        if (writer) {
            PendEntry pe;
            pe.cell = wcell;
            pe.slot = slot;
            pe.pad[0] = pe.pad[1] = pe.pad[2] = 0;
            park[parked + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull))] = pe;
        }
        parked += (uint32_t)__popcll(wm);
    } else if (writer) {
        ev::store_state<FIXED>(p, slot, wcell);
    }
}

// F = steps of a bucket held in registers (F * 64 requests): the bucket's elements, rate ids and cells are
// all requested before anything waits, the counting pass runs while the cells are on their way.
template <bool FULL, bool FIXED, bool BYSLOT, int F>
__device__ __forceinline__ void bucket_in_registers(const Params& p, const Req& rq0, const uint32_t* __restrict__ elems, uint32_t bucket,
                                                    int lb, uint32_t beg, uint32_t m, uint16_t* __restrict__ table,
                                                    uint8_t* __restrict__ mark, int lane, Tally& t) {
    uint32_t e[F];
    bool valid[F];
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const uint32_t k = (uint32_t)j * 64u + (uint32_t)lane;
        valid[j] = k < m;
        e[j] = valid[j] ? elems[beg + k] : 0u;
    }
    uint32_t rid[BYSLOT ? F : 1];
    Cell raw[F];
    RateClass rc[BYSLOT ? F : 1];
    rc[0].ei = rc[0].dvt = rc[0].burst = rc[0].pad = 0;
    if (BYSLOT) {
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const uint32_t slot = (bucket << lb) | (e[j] >> IDX_BITS);
            rid[BYSLOT ? j : 0] = (valid[j] && slot < p.capacity) ? (uint32_t)p.rate_id[slot] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const uint32_t slot = (bucket << lb) | (e[j] >> IDX_BITS);
        raw[j].tat = 0;
        raw[j].expiry = 0;
        if (valid[j] && slot < p.capacity) raw[j] = ev::load_raw<FIXED>(p, slot);
    }
    // pass A: requests per slot (order does not matter: LDS atomics, nothing waits)
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const uint32_t slot = (bucket << lb) | (e[j] >> IDX_BITS);
        if (valid[j] && slot < p.capacity) count_slot(table, e[j] >> IDX_BITS);
    }
    if (BYSLOT) {
#pragma unroll
        for (int j = 0; j < F; ++j) rc[BYSLOT ? j : 0] = p.classes[rid[BYSLOT ? j : 0]];
    }
    // every cell load of this wave has returned before its first store (a bucket belongs to one wave)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    uint32_t parked = 0;
#pragma unroll
    for (int j = F - 1; j >= 0; --j) {
        if ((uint32_t)j * 64u >= m) continue; // (wave-uniform)
        eval_step<FULL, FIXED, BYSLOT>(p, rq0, bucket, lb, valid[j], e[j], raw[j], rc[BYSLOT ? j : 0], table, mark, lane, t,
                                       nullptr, parked);
    }
}

constexpr int LONG_STEPS = 8; // steps per piece of a long bucket
inline size_t eval_lds_bytes(int lb) { return ((size_t)2 << lb) + ((size_t)1 << lb); }

template <bool FULL, bool FIXED, bool BYSLOT>
__global__ __launch_bounds__(64) void k_bucket_eval(Params p, Work w, PendEntry* __restrict__ park) {
    extern __shared__ uint32_t s_tab[]; // table u16[2^lb] | mark u8[2^lb]
    if (__builtin_nontemporal_load(w.maxb) > w.skew) return; // a skewed batch: the sort path takes it
    const int lb = w.lb;
    const uint32_t* __restrict__ elems = w.elems;
    uint16_t* table = reinterpret_cast<uint16_t*>(s_tab);
    uint8_t* mark = reinterpret_cast<uint8_t*>(table + (1u << lb));
    const int lane = threadIdx.x;
    const uint32_t bucket = blockIdx.x;
    const uint32_t beg = bucket_start(w, bucket), m = bucket_start(w, bucket + 1) - beg;
    if (m == 0) return;
    {
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t i = lane; i < (1u << lb) / 8; i += 64) t4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __builtin_amdgcn_wave_barrier();
    // the batch-wide request (one `now`, one `quantity`; rate: scalar, or the one registered plan)
    const RateClass rc_batch = p.classes[p.uniform_class]; // (class 0 if there is none)
    const Req rq0 = ev::make_req_rc(p, 0u, rc_batch);
    Tally t;
    t.na = t.nd = t.ne = 0;
    if (m <= 64u * 4u) {
        bucket_in_registers<FULL, FIXED, BYSLOT, 4>(p, rq0, elems, bucket, lb, beg, m, table, mark, lane, t);
    } else if (m <= 64u * 8u) {
        bucket_in_registers<FULL, FIXED, BYSLOT, 8>(p, rq0, elems, bucket, lb, beg, m, table, mark, lane, t);
    } else {
        // A long bucket (skewed stream, or a small key space).  Count, then walk it backwards in pieces; the
        // new states are parked in the bucket's own range of park[] until the last cell of the bucket has
        // been read (an earlier request of a key must not see the state a later one leaves), then stored.
        const uint32_t low = (1u << lb) - 1u;
        for (uint32_t k0 = 0; k0 < m; k0 += 64u * LONG_STEPS) {
            uint32_t e[LONG_STEPS];
#pragma unroll
            for (int j = 0; j < LONG_STEPS; ++j) {
                const uint32_t k = k0 + (uint32_t)j * 64u + (uint32_t)lane;
                e[j] = k < m ? elems[beg + k] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int j = 0; j < LONG_STEPS; ++j) {
                const uint32_t k = k0 + (uint32_t)j * 64u + (uint32_t)lane;
                const uint32_t v = (e[j] >> IDX_BITS) & low;
                if (k < m && ((bucket << lb) | v) < p.capacity) count_slot(table, v);
            }
        }
        PendEntry* mine = park + beg; // park[] has one entry per request of the batch
        uint32_t parked = 0;
        const uint32_t pieces = (m + 64u * LONG_STEPS - 1) / (64u * LONG_STEPS);
        for (uint32_t pc = pieces; pc-- > 0;) {
            const uint32_t k0 = pc * 64u * LONG_STEPS;
            uint32_t e[LONG_STEPS];
            bool valid[LONG_STEPS];
            Cell raw[LONG_STEPS];
            RateClass rc[BYSLOT ? LONG_STEPS : 1];
            rc[0] = rc_batch;
#pragma unroll
            for (int j = 0; j < LONG_STEPS; ++j) {
                const uint32_t k = k0 + (uint32_t)j * 64u + (uint32_t)lane;
                valid[j] = k < m;
                e[j] = valid[j] ? elems[beg + k] : 0u;
            }
#pragma unroll
            for (int j = 0; j < LONG_STEPS; ++j) {
                const uint32_t slot = (bucket << lb) | (e[j] >> IDX_BITS);
                raw[j].tat = 0;
                raw[j].expiry = 0;
                if (BYSLOT) rc[BYSLOT ? j : 0] = rc_batch;
                if (valid[j] && slot < p.capacity) {
                    raw[j] = ev::load_raw<FIXED>(p, slot);
                    if (BYSLOT) rc[BYSLOT ? j : 0] = p.classes[p.rate_id[slot]];
                }
            }
#pragma unroll
            for (int j = LONG_STEPS - 1; j >= 0; --j) {
                if (k0 + (uint32_t)j * 64u >= m) continue; // (wave-uniform)
                eval_step<FULL, FIXED, BYSLOT>(p, rq0, bucket, lb, valid[j], e[j], raw[j], rc[BYSLOT ? j : 0], table, mark, lane, t,
                                               mine, parked);
            }
        }
        // all loads of the bucket are done (their values were consumed): apply the parked states
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        for (uint32_t i = lane; i < parked; i += 64) {
            const PendEntry pe = mine[i];
            ev::store_state<FIXED>(p, pe.slot, pe.cell);
        }
    }
    wave_count3(t.na, t.nd, t.ne, p.counters);
}

} // namespace bp
