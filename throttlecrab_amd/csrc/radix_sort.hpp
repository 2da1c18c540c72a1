// radix_sort.hpp -- stable LSD radix sort of one batch's (slot, request index)
// pairs for gfx950, hand-written (no rocPRIM): 8-bit digits, one histogram
// kernel for all passes + one "onesweep" kernel per pass with decoupled
// look-back between tiles (single launch per pass, no inter-kernel scan).
//
// Element = u64 (slot << 32 | request index); only the slot bits are sorted
// on, so equal slots keep ascending request index (stability is what gives the
// reference's sequential order inside a key, rate_limiter.rs:102-250 applied in
// queue order, throttlecrab-server/src/actor.rs:217-236).
//
// Wave64 specifics: ranking uses 64-bit __ballot match masks (8 ballots per
// 8-bit digit) and per-wave LDS digit counters; no 32-lane assumptions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gcra_math.hpp" // tc::SpinGuard

#ifndef RS_SPIN_SLEEP
#define RS_SPIN_SLEEP 8 // s_sleep argument (x64 clocks) between look-back polls
#endif
#ifndef RS_ABLATE
#define RS_ABLATE 0 // tools/sortbench.hip sets 1..4 to time the kernel with phases cut off
#endif

namespace rs {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int HIST_THREADS = 1024; // k_hist: few big blocks -> few global atomics per histogram line
constexpr int HIST_BLOCKS = 128;
constexpr int RADIX = 256;
constexpr int MAX_PASSES = 4;
constexpr int GROUP = 16;          // tiles per look-back group
constexpr int GROUP_WINDOW = 16;   // predecessor groups examined per look-back round trip
constexpr uint32_t RESIDENT_TILES = 1024; // <= this many tiles are co-resident on 256 CUs (>= 4 blocks/CU)

constexpr uint32_t FLAG_PARTIAL = 1u << 30;
constexpr uint32_t FLAG_INCLUSIVE = 2u << 30;
constexpr uint32_t FLAG_MASK = 3u << 30;
constexpr uint32_t VALUE_MASK = ~FLAG_MASK;
constexpr uint32_t GACC_SHIFT = 24;               // group accumulator: arrivals << 24 | sum of counts
constexpr uint32_t GACC_SUM_MASK = (1u << GACC_SHIFT) - 1u;

// Scratch the sort needs, laid out in one device allocation.
//
// Look-back state of one pass (two levels, so that a 1 Mi batch whose tiles all
// start together does not serialise ~tiles/window L2 round trips on a chain of
// inclusive prefixes -- measured 12 of 23 us per pass with a one-level chain):
//   part [tile ][RADIX]  FLAG_PARTIAL | digit count of the tile          (plain store)
//   gacc [group][RADIX]  arrivals << 24 | sum of the group's counts      (atomicAdd by each tile)
//   gincl[group][RADIX]  FLAG_INCLUSIVE | prefix through the group       (stored by the group's last tile)
// A tile sums the part words of its predecessors inside its group (<= 15 loads in
// flight) and, walking backwards over whole groups, complete gacc words until it
// meets a published gincl (big grids: the first window) or runs out of groups.
struct Workspace {
    uint32_t* hist;      // [MAX_PASSES][RADIX] digit histograms of this batch (zero on entry)
    uint32_t* hist_next; // the other parity's histograms: cleared here for the next batch
    uint32_t* ticket;    // [MAX_PASSES] dynamic tile ids (forward progress for the look-back)
    uint32_t* status;    // [MAX_PASSES] x { part[max_tiles] | gacc[max_groups] | gincl[max_groups] } x RADIX
    uint32_t max_tiles;
    uint32_t max_groups;
    unsigned long long* violations; // the engine's invariant counter (a look-back that outlasts tc::SPIN_LIMIT_TICKS)
};

__host__ __device__ inline uint32_t groups_of(uint32_t tiles) { return (tiles + GROUP - 1) / GROUP; }
__host__ __device__ inline size_t pass_status_words(uint32_t max_tiles, uint32_t max_groups) {
    return ((size_t)max_tiles + 2 * (size_t)max_groups) * RADIX;
}
// words of the whole workspace (two histogram parities | tickets | status)
inline size_t workspace_words(uint32_t max_tiles) {
    return (size_t)2 * MAX_PASSES * RADIX + MAX_PASSES + (size_t)MAX_PASSES * pass_status_words(max_tiles, groups_of(max_tiles));
}
inline Workspace carve(uint32_t* base, uint32_t parity, uint32_t max_tiles) {
    Workspace ws;
    ws.hist = base + (size_t)parity * MAX_PASSES * RADIX;
    ws.hist_next = base + (size_t)(parity ^ 1u) * MAX_PASSES * RADIX;
    ws.ticket = base + (size_t)2 * MAX_PASSES * RADIX;
    ws.status = ws.ticket + MAX_PASSES;
    ws.max_tiles = max_tiles;
    ws.max_groups = groups_of(max_tiles);
    ws.violations = nullptr;
    return ws;
}

__device__ __forceinline__ uint32_t clamp_slot(uint32_t s, uint32_t cap) { return s < cap ? s : cap; }

// ---------------------------------------------------------------------------
// histogram of every pass's digit in one read of the slot column; also clears
// the look-back state of this sort (so no separate memset launches)
// ---------------------------------------------------------------------------
// `gate`: a batch that is enqueued on BOTH grouping paths (bucket_path.hpp) is sorted only if the partition
// found a bucket longer than `gate_min` requests (*gate, published by bp::k_bucket_scan earlier on the same
// stream); otherwise every block leaves at once -- block 0 after clearing the other parity's histograms,
// which is the one duty the host's parity flip relies on.
__device__ __forceinline__ bool gated_off(const uint32_t* __restrict__ gate, uint32_t gate_min) {
    return gate != nullptr && __builtin_nontemporal_load(gate) <= gate_min;
}

// `fill` (TC_B_OUTPUTS_IDLE batches): the batch's decision bytes, set to fill_value here, ahead of the evaluation,
// which then only stores the decisions that differ (16-byte stores; the tail bytewise)
template <int NT>
__global__ __launch_bounds__(NT) void k_hist(const uint32_t* __restrict__ slot, uint32_t n, uint32_t cap,
                                                  int passes, Workspace ws, uint32_t tiles, const uint32_t* __restrict__ gate,
                                                  uint32_t gate_min, uint8_t* __restrict__ fill, uint32_t fill_value) {
    __shared__ uint32_t s_h[MAX_PASSES][RADIX];
    if (fill != nullptr) {
        const uint32_t v4 = fill_value * 0x01010101u;
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)fill & 15u)) & 15u); // bytes before the first aligned 16
        const uint32_t h = head < n ? head : n;
        uint4* f16 = reinterpret_cast<uint4*>(fill + h);
        const uint32_t n16 = (n - h) / 16u;
        for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < n16; i += gridDim.x * NT) f16[i] = make_uint4(v4, v4, v4, v4);
        if (blockIdx.x == 0) {
            for (uint32_t i = threadIdx.x; i < h; i += NT) fill[i] = (uint8_t)fill_value;
            for (uint32_t i = h + n16 * 16u + threadIdx.x; i < n; i += NT) fill[i] = (uint8_t)fill_value;
        }
    }
    if (gated_off(gate, gate_min)) {
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < MAX_PASSES * RADIX; i += NT) ws.hist_next[i] = 0;
        return;
    }
    for (int i = threadIdx.x; i < MAX_PASSES * RADIX; i += NT) (&s_h[0][0])[i] = 0;
    // clear the look-back words this sort will use (every pass: part | gacc | gincl) + tickets
    {
        const uint32_t groups = groups_of(tiles);
        const uint32_t per_pass = (tiles + 2 * groups) * RADIX;
        const size_t pass_stride = pass_status_words(ws.max_tiles, ws.max_groups);
        const uint32_t total_status = (uint32_t)passes * per_pass;
        for (uint32_t i = blockIdx.x * NT + threadIdx.x; i < total_status; i += gridDim.x * NT) {
            const uint32_t p = i / per_pass;
            uint32_t r = i % per_pass;
            size_t at;
            if (r < tiles * RADIX) at = r;
            else if ((r -= tiles * RADIX) < groups * RADIX) at = (size_t)ws.max_tiles * RADIX + r;
            else at = ((size_t)ws.max_tiles + ws.max_groups) * RADIX + (r - groups * RADIX);
            ws.status[(size_t)p * pass_stride + at] = 0;
        }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < MAX_PASSES) ws.ticket[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < MAX_PASSES * RADIX; i += NT) ws.hist_next[i] = 0;
    }
    __syncthreads();
    {
        // 4 independent loads in flight per lane (the loop is latency-, not bandwidth-bound)
        const uint32_t stride = gridDim.x * NT;
        uint32_t i = blockIdx.x * NT + threadIdx.x;
        for (; i + 3 * stride < n; i += 4 * stride) {
            uint32_t k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = slot[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t kk = clamp_slot(k[u], cap);
                for (int p = 0; p < passes; ++p) atomicAdd(&s_h[p][(kk >> (8 * p)) & 255u], 1u);
            }
        }
        for (; i < n; i += stride) {
            const uint32_t kk = clamp_slot(slot[i], cap);
            for (int p = 0; p < passes; ++p) atomicAdd(&s_h[p][(kk >> (8 * p)) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * RADIX; i += NT) {
        const uint32_t v = (&s_h[0][0])[i];
        if (v) atomicAdd(&ws.hist[i], v);
    }
}

// ---------------------------------------------------------------------------
// one LSD pass.  FIRST: input is the raw slot column (value = position).
// ---------------------------------------------------------------------------
template <int ITEMS, bool FIRST>
__global__ __launch_bounds__(THREADS) void k_onesweep(const uint32_t* __restrict__ slot_in,
                                                      const uint64_t* __restrict__ elem_in,
                                                      uint64_t* __restrict__ elem_out, uint32_t n, uint32_t cap,
                                                      int pass, Workspace ws, const uint32_t* __restrict__ gate, uint32_t gate_min) {
    constexpr int TILE = THREADS * ITEMS;
    if (gated_off(gate, gate_min)) return; // (the whole grid: nobody is left waiting in a look-back)
    __shared__ uint32_t s_base[RADIX];          // global exclusive start of each digit
    __shared__ uint32_t s_wave[WAVES][RADIX];   // per-wave digit counts -> exclusive prefix over waves
    __shared__ uint32_t s_off[RADIX];           // where this tile's run of each digit starts
    __shared__ uint32_t s_tstart[RADIX];        // where each digit's run starts inside this tile
    __shared__ uint64_t s_elem[TILE];           // the tile in sorted order (coalesced write-out)
    __shared__ uint32_t s_scan[WAVES];
    __shared__ uint32_t s_tile;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Tile order must guarantee that every predecessor a tile waits for is running.
    // With <= RESIDENT_TILES tiles the whole grid is co-resident, so blockIdx will do;
    // beyond that, tiles take a ticket (one contended atomic per block: ~12 ns each).
    if (threadIdx.x == 0) s_tile = (gridDim.x <= RESIDENT_TILES) ? blockIdx.x : atomicAdd(&ws.ticket[pass], 1u);
    for (int i = threadIdx.x; i < WAVES * RADIX; i += THREADS) (&s_wave[0][0])[i] = 0;

    // exclusive scan of the batch histogram of this pass's digit (RADIX == THREADS)
    {
        const uint32_t h = ws.hist[pass * RADIX + threadIdx.x];
        uint32_t v = h;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v += o;
        }
        if (lane == 63) s_scan[wave] = v;
        __syncthreads();
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += s_scan[w];
        s_base[threadIdx.x] = carry + v - h;
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t shift = 8u * (uint32_t)pass;

    // wave-striped tile: item j of lane l of wave w sits at tile*TILE + w*64*ITEMS + j*64 + l
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const uint32_t wbase = tile * TILE + wave * 64 * ITEMS + lane;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = wbase + j * 64;
        if (pos < n) {
            if (FIRST) {
                key[j] = clamp_slot(slot_in[pos], cap);
                val[j] = pos;
            } else {
                const uint64_t e = elem_in[pos];
                key[j] = (uint32_t)(e >> 32);
                val[j] = (uint32_t)e;
            }
        } else {
            key[j] = 0xFFFFFFFFu;
            val[j] = 0;
        }
    }
    if (RS_ABLATE == 1) { // loads only
        uint32_t acc = 0;
        for (int j = 0; j < ITEMS; ++j) acc ^= key[j] ^ val[j];
        if (acc == 0x12345678u) elem_out[0] = acc;
        return;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool valid = (wbase + j * 64) < n;
        const uint32_t d = (key[j] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        // m = lanes of this wave (this round) holding the same digit
        const uint32_t before = valid ? s_wave[wave][d] : 0u; // all lanes read, then the leader adds
        rank[j] = before + (uint32_t)__popcll(m & lt);
        if (valid && (m & lt) == 0ull) s_wave[wave][d] = before + (uint32_t)__popcll(m);
        // same-wave LDS ops execute in program order: the next round's read follows this write
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    if (RS_ABLATE == 2) { // + ranking
        uint32_t acc = 0;
        for (int j = 0; j < ITEMS; ++j) acc ^= rank[j];
        if (acc == 0x12345678u) elem_out[0] = acc;
        return;
    }
    // digit = threadIdx.x: prefix over waves -> tile count; tile-local digit starts
    uint32_t run = 0;
    {
        const int d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_wave[w][d];
            s_wave[w][d] = run;
            run += c;
        }
        // publish this tile's digit counts right away so successors can look back
        uint32_t* part = ws.status + (size_t)pass * pass_status_words(ws.max_tiles, ws.max_groups);
        uint32_t* gacc = part + (size_t)ws.max_tiles * RADIX;
        __hip_atomic_store(&part[(size_t)tile * RADIX + d], FLAG_PARTIAL | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile / GROUP + 1 < groups_of(gridDim.x)) // nobody reads the last group's accumulator
            __hip_atomic_fetch_add(&gacc[(size_t)(tile / GROUP) * RADIX + d], (1u << GACC_SHIFT) | run, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        uint32_t v = run;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(v, off, 64);
            if (lane >= off) v += o;
        }
        if (lane == 63) s_scan[wave] = v;
        __syncthreads();
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += s_scan[w];
        s_tstart[d] = carry + v - run;
    }
    __syncthreads();

    // stage the tile in LDS in its sorted order: the global writes below are then
    // contiguous per digit run (full lines) instead of 4096 scattered 8-byte stores
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((wbase + j * 64) < n) {
            const uint32_t d = (key[j] >> shift) & 255u;
            s_elem[s_tstart[d] + s_wave[wave][d] + rank[j]] = ((uint64_t)key[j] << 32) | val[j];
        }
    }

    // look back for the exclusive prefix of each digit over the preceding tiles
    {
        const int d = threadIdx.x;
        uint32_t* part = ws.status + (size_t)pass * pass_status_words(ws.max_tiles, ws.max_groups);
        uint32_t* gacc = part + (size_t)ws.max_tiles * RADIX;
        uint32_t* gincl = gacc + (size_t)ws.max_groups * RADIX;
        const uint32_t g = tile / GROUP, j = tile % GROUP;
        uint32_t excl = 0;
        if (tile != 0 && RS_ABLATE != 3) {
            // (1) predecessors inside my group: every part word in flight at once
            if (j != 0) {
                const uint32_t* pp = part + (size_t)(g * GROUP) * RADIX + d;
                const uint32_t want = (1u << j) - 1u;
                uint32_t got = 0;
                tc::SpinGuard guard;
                while (true) {
                    if (tc::spin_expired(guard)) { // (flagged, never hung)
                        tc::invariant_failed(ws.violations);
                        break;
                    }
                    uint32_t s[GROUP - 1];
#pragma unroll
                    for (int u = 0; u < GROUP - 1; ++u)
                        s[u] = ((uint32_t)u < j && !((got >> u) & 1u))
                                   ? __hip_atomic_load(&pp[(size_t)u * RADIX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : 0u;
#pragma unroll
                    for (int u = 0; u < GROUP - 1; ++u)
                        if (s[u] & FLAG_MASK) {
                            excl += s[u] & VALUE_MASK;
                            got |= 1u << u;
                        }
                    if (got == want) break;
                    __builtin_amdgcn_s_sleep(RS_SPIN_SLEEP);
                }
            }
            // (2) whole groups before mine, newest first: a published inclusive prefix ends the
            // walk, a complete accumulator contributes the group's sum, anything else is retried
            int gg = (int)g - 1;
            tc::SpinGuard guard2;
            while (gg >= 0) {
                if (tc::spin_expired(guard2)) {
                    tc::invariant_failed(ws.violations);
                    break;
                }
                uint32_t a[GROUP_WINDOW], b[GROUP_WINDOW];
#pragma unroll
                for (int u = 0; u < GROUP_WINDOW; ++u) {
                    if (gg - u >= 0) {
                        a[u] = __hip_atomic_load(&gincl[(size_t)(gg - u) * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        b[u] = __hip_atomic_load(&gacc[(size_t)(gg - u) * RADIX + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        a[u] = FLAG_INCLUSIVE; // virtual group -1: inclusive prefix 0
                        b[u] = 0;
                    }
                }
                bool done = false;
                int used = 0;
#pragma unroll
                for (int u = 0; u < GROUP_WINDOW; ++u) {
                    if (done || used < u) continue;
                    if (a[u] & FLAG_MASK) {
                        excl += a[u] & VALUE_MASK;
                        used = u + 1;
                        done = true;
                    } else if ((b[u] >> GACC_SHIFT) == (uint32_t)GROUP) {
                        excl += b[u] & GACC_SUM_MASK;
                        used = u + 1;
                    }
                }
                if (done) break;
                gg -= used;
                if (used == 0) __builtin_amdgcn_s_sleep(RS_SPIN_SLEEP);
            }
            if (j == GROUP - 1 && g + 1 < groups_of(gridDim.x))
                __hip_atomic_store(&gincl[(size_t)g * RADIX + d], FLAG_INCLUSIVE | (excl + run), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        s_off[d] = s_base[d] + excl - s_tstart[d]; // global position = tile-local position + this
    }
    __syncthreads();

    if (RS_ABLATE == 4) return; // everything but the write-out
    const uint32_t tile_first = tile * TILE;
    const uint32_t nvalid = (n - tile_first) < (uint32_t)TILE ? (n - tile_first) : (uint32_t)TILE;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t i = j * THREADS + threadIdx.x;
        if (i < nvalid) {
            const uint64_t e = s_elem[i];
            const uint32_t d = ((uint32_t)(e >> 32) >> shift) & 255u;
            elem_out[i + s_off[d]] = e;
        }
    }
}

} // namespace rs
