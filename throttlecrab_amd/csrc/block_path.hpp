// block_path.hpp -- grouping a uniform batch with ONE pass over global memory.
//
//   rs::k_hist       counts the requests of each of the 1024 equal SUB-RANGES of the key space and publishes the
//                    longest one (the gate) and the requests per RANGE (4 sub-ranges);
//   rs::k_onesweep   (rmul != 0) stable partition of (slot, request index) into the 256 ranges: the same kernel
//                    as a radix pass, coalesced 8-byte elements, ~one full memory line per range and tile;
//   k_block_eval     ONE 256-THREAD BLOCK PER SUB-RANGE: picks its ~n/1024 elements out of its range (stable
//                    ballot compaction; the 4 blocks of a range run on the same XCD, so three of them read the
//                    range from L2), requests their cells, sorts the elements by slot inside LDS while the cells
//                    are on their way (two 8-bit passes, wave64 ballot ranking), takes every request's rank inside
//                    its key from the sorted order and evaluates.
//
// Against the sort path this drops two of the three passes over global memory (2 x 16 MB of traffic, 2 launches,
// 2 chains of look-back) and the separate read of the sorted batch.
//
// Semantics are k_eval_sorted's (eval_kernels.hpp): requests of a key in request order (partition, compaction
// and both LDS passes are stable), closed form for regular runs, request-by-request walk for irregular ones.  A
// key's requests all sit in one block, so every cell load of the block is complete (block barrier) before the
// first cell store: no parked stores, no commit launch, nothing waits for another block -- for every uniform
// batch, not only those whose runs the host can prove regular.
//
// Skew: a sub-range holding more than CAP requests does not fit a block.  The kernels here leave at once when
// the gate exceeds the limit, and the gated sort path (enqueued behind them by the host, which may be many
// batches ahead) runs instead.
//
// Reference: the evaluated function is RateLimiter::rate_limit (throttlecrab/src/core/rate_limiter.rs:102-250)
// applied in queue order (throttlecrab-server/src/actor.rs:217-236).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eval_kernels.hpp"
#include "radix_sort.hpp"

namespace bk {

using ev::Params;
using ev::Req;
using tc::Cell;
using tc::Decision;
using tc::RateClass;

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int ITEMS = 6;
constexpr uint32_t CAP = THREADS * ITEMS; // requests of one sub-range a block takes
constexpr int SUBS = rs::SUBS;
constexpr int SUB_BITS = rs::SUB_BITS;
constexpr int SEG_ITEMS = 8;              // compaction: the range is read in segments of THREADS * SEG_ITEMS elements
constexpr uint32_t MAX_WIDTH = 65536;     // slots per sub-range: the in-block sort key has 16 bits

// slot -> sub-range: __umulhi(slot, mul), monotone, < SUBS for every slot <= capacity (the clamp value included)
inline uint32_t range_mul(uint64_t capacity) { return (uint32_t)((((uint64_t)SUBS) << 32) / (capacity + 1)); }
// first slot of sub-range b: the smallest s with s * mul >= b << 32
__host__ __device__ inline uint32_t range_first(uint32_t b, uint32_t mul) { return (uint32_t)((((uint64_t)b << 32) + mul - 1) / mul); }
inline uint64_t range_width(uint32_t mul) { return (((uint64_t)1 << 32) + mul - 1) / mul + 1; } // (upper bound)
// does the key space fit the path?  (>= 4096 keys; every sub-range narrower than MAX_WIDTH)
inline bool fits(uint64_t capacity) {
    if (capacity < 4096 || capacity >= ((uint64_t)1 << 31)) return false;
    return range_width(range_mul(capacity)) <= MAX_WIDTH;
}

// One stable 8-bit pass of the block's elements through LDS.  e[]/valid[]: the elements at this thread's
// wave-striped positions (item j of lane l of wave w = position w * chunk + j * 64 + l); on return s_elem holds
// them ordered by digit and at[j] is where element j went.  The caller's reads of s_elem precede this call's
// first barrier.
template <int SHIFT>
__device__ __forceinline__ void lds_pass(const uint64_t (&e)[ITEMS], const bool (&valid)[ITEMS], uint32_t first, uint64_t* __restrict__ s_elem,
                                         uint32_t (*__restrict__ s_wave)[rs::RADIX], uint32_t* __restrict__ s_dstart,
                                         uint32_t* __restrict__ s_scan, uint32_t (&at)[ITEMS]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < WAVES * rs::RADIX; i += THREADS) (&s_wave[0][0])[i] = 0;
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = (((uint32_t)(e[j] >> 32) - first) >> SHIFT) & 255u;
        unsigned long long m = __ballot(valid[j]);
        at[j] = 0;
        if (m == 0ull) continue; // (wave-uniform: this wave's stripe ended)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t before = valid[j] ? s_wave[wave][d] : 0u; // all lanes read, then the group's first lane adds
        at[j] = before + (uint32_t)__popcll(m & lt);
        if (valid[j] && (m & lt) == 0ull) s_wave[wave][d] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier(); // same-wave LDS ops execute in program order
    }
    __syncthreads();
    {
        // digit = thread (RADIX == THREADS): prefix over the waves, then over the digits
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_wave[w][d];
            s_wave[w][d] = run;
            run += c;
        }
        uint32_t v = run;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v += o;
        }
        if (lane == 63) s_scan[wave] = v;
        __syncthreads();
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += s_scan[w];
        s_dstart[d] = carry + v - run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (valid[j]) {
            const uint32_t d = (((uint32_t)(e[j] >> 32) - first) >> SHIFT) & 255u;
            at[j] += s_dstart[d] + s_wave[wave][d];
            s_elem[at[j]] = e[j];
        }
    }
    __syncthreads();
}

// Block -> sub-range: the 4 blocks of a range get ids that are equal mod 8 (the same XCD) and close together
// (dispatched together), so the range's elements come from memory once and from that XCD's L2 three times.
__device__ __forceinline__ uint32_t sub_of_block(uint32_t id) {
    const uint32_t x = id & 7u, y = id >> 3;
    const uint32_t range = (y >> SUB_BITS) * 8u + x;
    return (range << SUB_BITS) | (y & ((1u << SUB_BITS) - 1u));
}

// `part`: the batch partitioned by range (rs::k_onesweep, rmul != 0); ws.hist row RANGE_ROW: requests per range;
// ws.sub: requests per sub-range.  Grid: SUBS blocks.
template <bool FULL, bool FIXED>
__global__ __launch_bounds__(THREADS) void k_block_eval(Params p, const uint64_t* __restrict__ part, rs::Workspace ws, uint32_t rmul,
                                                         uint32_t sort_bits, uint32_t longest) {
    static_assert(rs::RADIX == THREADS, "one thread per digit / per range");
    if (__builtin_nontemporal_load(ws.gate) > longest) return; // (longest <= CAP) a sub-range too long for a block: this batch is sorted
    __shared__ uint64_t s_elem[CAP + 1];
    __shared__ uint32_t s_wave[WAVES][rs::RADIX];
    __shared__ uint32_t s_dstart[rs::RADIX];
    __shared__ uint32_t s_scan[WAVES];
    __shared__ uint32_t s_cnt[2][WAVES];
    __shared__ uint16_t s_map[CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sub = sub_of_block(blockIdx.x), range = sub >> SUB_BITS;
    const bool class_by_slot = (p.flags & ev::F_REGISTERED) && !(p.flags & ev::F_UNIFORM_CLASS);
    const RateClass rc_batch = p.classes[p.uniform_class];

    // my range of the partitioned batch: requests of the ranges before it, its requests, those of my sub-range
    const uint32_t m = ws.sub[sub];
    const uint32_t mr = ws.hist[rs::RANGE_ROW * rs::RADIX + range];
    {
        uint32_t v = (uint32_t)threadIdx.x < range ? ws.hist[rs::RANGE_ROW * rs::RADIX + threadIdx.x] : 0u;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_scan[wave] = v;
    }
    const uint32_t first = range_first(sub, rmul);
    __syncthreads();
    const uint32_t rstart = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    if (m == 0) return;
    // rows of TC_B_GROUPED_OUTPUT: my sub-range starts after the range's earlier sub-ranges
    uint32_t bstart = rstart;
    for (uint32_t u = range << SUB_BITS; u < sub; ++u) bstart += ws.sub[u];

    // ---- stable compaction: the range's elements that fall into my sub-range -> s_elem[0 .. m) ---------------
    uint32_t filled = 0;
    for (uint32_t seg = 0; seg < mr; seg += THREADS * SEG_ITEMS) {
        uint64_t x[SEG_ITEMS];
        bool keep[SEG_ITEMS];
        unsigned long long km[SEG_ITEMS];
        uint32_t mine = 0; // kept by my wave in this segment (wave-uniform)
        const uint32_t wfirst = seg + (uint32_t)wave * (64u * SEG_ITEMS) + (uint32_t)lane;
#pragma unroll
        for (int j = 0; j < SEG_ITEMS; ++j) {
            const uint32_t i = wfirst + (uint32_t)j * 64u;
            x[j] = part[rstart + (i < mr ? i : 0u)];
        }
#pragma unroll
        for (int j = 0; j < SEG_ITEMS; ++j) {
            const uint32_t i = wfirst + (uint32_t)j * 64u;
            keep[j] = i < mr && __umulhi((uint32_t)(x[j] >> 32), rmul) == sub;
            km[j] = __ballot(keep[j]);
            mine += (uint32_t)__popcll(km[j]);
        }
        const int par = (seg / (THREADS * SEG_ITEMS)) & 1; // (double-buffered: one barrier per segment)
        if (lane == 0) s_cnt[par][wave] = mine;
        __syncthreads();
        uint32_t at = filled;
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_cnt[par][w];
            if (w < wave) at += c;
            filled += c;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < SEG_ITEMS; ++j) {
            if (keep[j]) s_elem[at + (uint32_t)__popcll(km[j] & lt)] = x[j];
            at += (uint32_t)__popcll(km[j]);
        }
    }
    __syncthreads();

    // ---- my elements (wave-striped); their cells are requested before the sort, which hides the latency ------
    const uint32_t chunk = ((m + THREADS - 1) / THREADS) * 64u; // every wave gets the same share (a multiple of 64)
    uint64_t e[ITEMS];
    bool valid[ITEMS], has_cell[ITEMS];
    uint32_t rid[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t q = (uint32_t)wave * chunk + (uint32_t)j * 64u + (uint32_t)lane;
        valid[j] = (uint32_t)j * 64u < chunk && q < m;
        e[j] = valid[j] ? s_elem[q] : ~0ull;
        has_cell[j] = valid[j] && (uint32_t)(e[j] >> 32) < p.capacity;
        rid[j] = 0;
    }
    if (class_by_slot) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rid[j] = (uint32_t)p.rate_id[has_cell[j] ? (uint32_t)(e[j] >> 32) : 0u];
    }
    Cell cell[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        cell[j] = ev::load_raw<FIXED>(p, has_cell[j] ? (uint32_t)(e[j] >> 32) : 0u);
        if (!has_cell[j]) {
            cell[j].tat = FIXED ? tc::TAT_VACANT : 0;
            cell[j].expiry = 0;
        }
    }
    RateClass rc[ITEMS];
    if (class_by_slot) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rc[j] = p.classes[rid[j]];
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) rc[j] = rc_batch;
    }

    // ---- sort by slot inside LDS; at[j] = where my element j ends up ----------------------------------------
    uint32_t at[ITEMS];
    lds_pass<0>(e, valid, first, s_elem, s_wave, s_dstart, s_scan, at);
    if (sort_bits > 8) {
        // the second pass takes the elements in the order the first one left them; my own elements follow
        // through the position -> position map of that pass
        uint64_t e2[ITEMS];
        uint32_t at2[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t q = (uint32_t)wave * chunk + (uint32_t)j * 64u + (uint32_t)lane;
            e2[j] = valid[j] ? s_elem[q] : ~0ull;
        }
        lds_pass<8>(e2, valid, first, s_elem, s_wave, s_dstart, s_scan, at2);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const uint32_t q = (uint32_t)wave * chunk + (uint32_t)j * 64u + (uint32_t)lane;
            if (valid[j]) s_map[q] = (uint16_t)at2[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (valid[j]) at[j] = s_map[at[j]];
    }

    // ---- evaluation of my own elements: neighbours and rank from the sorted array ---------------------------
    // A request that is not the first of its key finds the start of its run by galloping backwards and bisecting
    // (2-3 LDS reads for the short runs of a uniform stream, <= 2 log2 m for a key that fills the sub-range).
    uint32_t na = 0, nd = 0, ne = 0;
    bool writer[ITEMS];
    Cell wcell[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        writer[j] = false;
        wcell[j].tat = 0;
        wcell[j].expiry = 0;
        if (!valid[j]) continue;
        const uint32_t k = at[j];
        const uint32_t slot = (uint32_t)(e[j] >> 32), idx = (uint32_t)e[j];
        const bool head = k == 0 || (uint32_t)(s_elem[k - 1] >> 32) != slot;
        const bool is_last = k + 1 == m || (uint32_t)(s_elem[k + 1] >> 32) != slot;
        uint32_t r = 0;
        if (!head) {
            uint32_t pos = k - 1, step = 1; // s_elem[pos] is of my key
            while (pos >= step && (uint32_t)(s_elem[pos - step] >> 32) == slot) {
                pos -= step;
                step <<= 1;
            }
            uint32_t lo = pos >= step ? pos - step + 1 : 0u; // s_elem[lo - 1] is not (or lo == 0)
            while (lo < pos) {
                const uint32_t mid = (lo + pos) >> 1;
                if ((uint32_t)(s_elem[mid] >> 32) == slot) pos = mid;
                else lo = mid + 1;
            }
            r = k - pos;
        }
        const uint32_t orow = p.order ? bstart + k : idx; // (TC_B_GROUPED_OUTPUT: sub-ranges in order, sorted inside: rows grouped by key)
        if (p.order) p.order[bstart + k] = idx;
        const Req rq = ev::make_req_rc(p, slot, rc[j]);
        Decision d;
        d.allowed = false;
        d.remaining = d.reset_after = d.retry_after = 0;
        if (rq.status != tc::ST_OK) {
            ne += 1;
            ev::write_out(p, orow, rq, d);
            continue;
        }
        Cell c = FIXED ? tc::fixed_cell(cell[j].tat, rq.dvt) : cell[j];
        const Decision d0 = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now); // c = cell after request 0
        if (!d0.allowed) {
            // request 0 denied => state untouched => every request of the run equals request 0
            nd += 1;
            if (p.denied) atomicAdd(&p.denied[slot], 1u);
            ev::write_out(p, orow, rq, d0);
            continue;
        }
        if (head && is_last) {
            na += 1; // a key requested once in this batch
            ev::write_out(p, orow, rq, d0);
            writer[j] = true;
            wcell[j] = c;
            continue;
        }
        // closed form of the run (tc::run_form; decisions only: tc::run_lite, the same provisos without the division)
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
                owner = ok_r && (is_last || (int64_t)r + 1 == f.n_tot);
            } else if (regular) {
                owner = is_last || f.n_tot == 1;
            }
        } else {
            const tc::RunLite f = tc::run_lite(c, rq.ei, rq.dvt, rq.q, rq.now);
            regular = f.regular;
            if (regular) {
                ok_r = r == 0 || tc::rank_allowed(f, r);
                owner = ok_r && (is_last || !tc::rank_allowed(f, r + 1u));
                if (owner && r != 0) v = tc::cell_after(f.new0 + (int64_t)r * f.inc, rq.dvt, rq.now);
                d.allowed = ok_r;
            }
        }
        if (r == 0) {
            na += 1;
            ev::write_out(p, orow, rq, d0);
            if (regular || is_last) {
                writer[j] = owner || is_last;
                wcell[j] = c;
            } else {
                // irregular run (saturation, zero increment, immediate expiry): walk the rest one by one
                uint32_t wd = 0;
                for (uint32_t q = k + 1; q < m; ++q) {
                    const uint64_t nx = s_elem[q];
                    if ((uint32_t)(nx >> 32) != slot) break;
                    const Decision dj = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now);
                    na += dj.allowed;
                    wd += !dj.allowed;
                    ev::write_out(p, p.order ? bstart + q : (uint32_t)nx, rq, dj);
                }
                nd += wd;
                if (p.denied && wd) atomicAdd(&p.denied[slot], wd); // the whole run's denials sit in this lane
                writer[j] = true;
                wcell[j] = c;
            }
        } else if (regular) {
            na += ok_r;
            nd += !ok_r;
            if (p.denied && !ok_r) atomicAdd(&p.denied[slot], 1u); // (skewed streams trip the gate: one atomic per key run there)
            ev::write_out(p, orow, rq, d);
            writer[j] = owner;
            wcell[j] = v;
        }
        // irregular && r > 0: the run's first lane produced this request's outputs
    }
    // every cell of this sub-range has been read (the values were consumed above) before the first one is replaced
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (writer[j]) ev::store_state<FIXED>(p, (uint32_t)(e[j] >> 32), wcell[j]);
    ev::block_count3<THREADS>(na, nd, ne, p.counters);
}

} // namespace bk
