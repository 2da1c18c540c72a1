// route_kernels.hpp -- routing of a GLOBAL request stream to the GPU that owns each key (gfx950).
//
// The reference has no distributed mode ("use client-side sharding by key", README.md:247-249); here keys are
// independent units, so a key space of `world * keys_per_shard` global ids is sharded with no data-path
// collective: every global id maps to (owner, shard-local slot) by a BIJECTION of [0, world * keys_per_shard)
//     q = id div world,  r = id mod world:    slot = q,    owner = (r + mix(q)) mod world,
//     mix(q) = the top bits of q * 0x9E3779B1 scaled to [0, world)
// -- every shard gets exactly keys_per_shard dense slots, consecutive ids land on different owners, ids that
// share a residue mod world (a strided id space) are still spread over all owners, no routing table is needed,
// and the inverse is as cheap (id = slot * world + (owner - mix(slot)) mod world).  The router looks at every id
// of the global batch on every GPU, so the map has to be cheap: ~12 integer instructions, no division (round 2's
// first map, a multiplicative permutation modulo world * keys_per_shard, needed two 64-bit reductions per id and
// made the router ALU-bound: 62 us of kernels for 8 Mi ids).
// A GPU is handed the global batch -- or its part of it -- and keeps what it owns:
//   k_route_count    requests per destination, per 4096-request tile
//   k_route_scan     prefix of those counts over the tiles + destination totals            (one block per destination)
//   k_route_scatter  stable compaction per destination (wave ballots): shard-local slots and, optionally, the
//                    requests' positions in the global batch, in request order
//   k_route_publish  (callers that poll pinned host memory) the totals and the caller's tag
// Stable: the requests of a key keep their order, which is all the sequence semantics need.
// `only` >= 0 writes that destination's requests alone (what one rank of bench.py --gpus N does); -1 writes every
// destination's segment one after the other (a front end that forwards segments over xGMI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gcra_math.hpp" // tc::SpinGuard

namespace rt {

constexpr int THREADS = 256, ITEMS = 16;
constexpr uint32_t TILE = THREADS * ITEMS;
constexpr uint32_t MAX_WORLD = 64;
constexpr uint32_t MIX = 0x9E3779B1u;

struct Map {
    uint64_t modulus; // world * keys_per_shard: ids at or above it are taken modulo
    uint32_t world;
    uint32_t inv_world; // floor(2^32 / world) (world >= 2): the device divides by multiplying
};

// the map of a (world, keys_per_shard) pair: the same on every rank, the host mirror and the device
inline bool make_map(uint32_t world, uint64_t keys_per_shard, Map* out) {
    if (world == 0 || world > MAX_WORLD || keys_per_shard == 0 || keys_per_shard > ((uint64_t)1 << 32)) return false;
    out->modulus = (uint64_t)world * keys_per_shard;
    out->world = world;
    out->inv_world = world > 1 ? (uint32_t)(((uint64_t)1 << 32) / world) : 0u;
    return true;
}
inline uint32_t mix_of(uint32_t q, uint32_t world) { return (uint32_t)(((uint64_t)(uint32_t)(q * MIX) * world) >> 32); }
// host mirror (and the definition): plain division
inline void route_of_host(const Map& m, uint32_t id, uint32_t& owner, uint32_t& slot) {
    const uint64_t v = (uint64_t)id < m.modulus ? id : (uint64_t)id % m.modulus;
    const uint32_t q = (uint32_t)(v / m.world), r = (uint32_t)(v % m.world);
    slot = q;
    owner = (r + mix_of(q, m.world)) % m.world;
}
inline uint64_t route_inverse_host(const Map& m, uint32_t owner, uint32_t slot) {
    const uint32_t r = (owner + m.world - mix_of(slot, m.world)) % m.world;
    return (uint64_t)slot * m.world + r;
}

#if defined(__HIPCC__)
// the same values without a division instruction: q' = floor(v * floor(2^32 / world) / 2^32) is q or q - 1
__device__ __forceinline__ void route_of(const Map& m, uint32_t id, uint32_t& owner, uint32_t& slot) {
    uint32_t v = id;
    if ((uint64_t)v >= m.modulus) v = (uint32_t)((uint64_t)v % m.modulus); // (ids outside the key space: rare, slow path)
    uint32_t q = v, r = 0;
    if (m.world > 1) {
        q = __umulhi(v, m.inv_world);
        r = v - q * m.world;
        if (r >= m.world) {
            r -= m.world;
            q += 1;
        }
    }
    uint32_t o = r + __umulhi(q * MIX, m.world);
    if (o >= m.world) o -= m.world;
    owner = o;
    slot = q;
}
#endif

struct Work {
    uint32_t* tile_cnt; // [world][tiles] requests per destination and tile -> exclusive prefix over the tiles
    uint32_t* totals;   // [world]
    uint32_t tiles;
    volatile uint32_t* host_totals; // NULL or pinned host memory [world + 1]: the totals, then `tag` (written last)
    uint32_t tag;
};

static __global__ __launch_bounds__(THREADS) void k_route_count(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w) {
    __shared__ uint32_t s_c[MAX_WORLD];
    if (threadIdx.x < MAX_WORLD) s_c[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t base = blockIdx.x * TILE + threadIdx.x;
    const int bits = m.world > 1 ? 32 - __clz((int)m.world - 1) : 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) v[j] = base + j * THREADS < n ? id[base + j * THREADS] : 0u;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        // one LDS atomic per destination and wave step (64 lanes adding to a handful of words serialise otherwise)
        const bool valid = base + j * THREADS < n;
        uint32_t owner, slot;
        route_of(m, v[j], owner, slot);
        unsigned long long mm = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const unsigned long long bb = __ballot((owner >> b) & 1u);
            mm &= ((owner >> b) & 1u) ? bb : ~bb;
        }
        if (valid && (mm & lt) == 0ull) atomicAdd(&s_c[owner], (uint32_t)__popcll(mm));
    }
    __syncthreads();
    // (tile counts, their prefixes and the totals travel between three small kernels whose blocks sit on different XCDs:
    // written and read at agent scope -- past the XCDs' private L2s -- so that they never depend on what a kernel
    // boundary does to a few dirty words)
    if (threadIdx.x < m.world)
        __hip_atomic_store(&w.tile_cnt[(size_t)threadIdx.x * w.tiles + blockIdx.x], s_c[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one block per destination: exclusive prefix of its counts over the tiles (each thread a contiguous piece), total
static __global__ __launch_bounds__(THREADS) void k_route_scan(Work w) {
    __shared__ uint32_t s_part[THREADS];
    uint32_t* row = w.tile_cnt + (size_t)blockIdx.x * w.tiles;
    const uint32_t per = (w.tiles + THREADS - 1) / THREADS;
    const uint32_t lo = min(threadIdx.x * per, w.tiles), hi = min(lo + per, w.tiles);
    uint32_t sum = 0;
    for (uint32_t t = lo; t < hi; ++t) sum += __hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // exclusive prefix of the pieces (Hillis-Steele over 256 values in LDS)
    for (uint32_t off = 1; off < THREADS; off <<= 1) {
        const uint32_t o = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (uint32_t t = lo; t < hi; ++t) {
        const uint32_t c = __hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&row[t], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        run += c;
    }
    if (threadIdx.x == THREADS - 1) __hip_atomic_store(&w.totals[blockIdx.x], s_part[THREADS - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SPLIT (only < 0): segment d goes to dst.ptr[d], from its start -- the destination's inbox in ITS memory (peer memory
// over xGMI) -- instead of into one column: the exchange of a sharded deployment needs no separate forwarding step
struct SplitOut {
    uint32_t* ptr[MAX_WORLD];
};
template <bool SPLIT>
__global__ __launch_bounds__(THREADS) void k_route_scatter(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w, int only,
                                                            uint32_t* __restrict__ out_slot, uint32_t* __restrict__ out_pos, SplitOut dst) {
    __shared__ uint32_t s_wave[THREADS / 64][MAX_WORLD]; // per-wave counts -> exclusive prefix over the waves
    __shared__ uint32_t s_start[MAX_WORLD];              // where each destination's segment starts in the output
    __shared__ uint32_t s_tile[MAX_WORLD];               // ... and this tile's requests inside it (k_route_scan's prefix)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (THREADS / 64) * MAX_WORLD; i += THREADS) (&s_wave[0][0])[i] = 0;
    if (threadIdx.x < m.world) {
        uint32_t at = 0;
        if (only < 0 && !SPLIT)
            for (uint32_t k = 0; k < threadIdx.x; ++k) at += __hip_atomic_load(&w.totals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_start[threadIdx.x] = at;
        s_tile[threadIdx.x] = __hip_atomic_load(&w.tile_cnt[(size_t)threadIdx.x * w.tiles + blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // wave-striped: step j of wave k covers requests tile*TILE + k*1024 + j*64 + lane (request order inside a wave)
    const uint32_t first = blockIdx.x * TILE + wave * (64 * ITEMS) + lane;
    uint32_t dest[ITEMS], slot[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = first + j * 64;
        route_of(m, pos < n ? id[pos] : 0u, dest[j], slot[j]);
        if (pos >= n) dest[j] = 0xFFFFFFFFu;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        rank[j] = 0;
        for (uint32_t d = 0; d < m.world; ++d) { // (wave-uniform loop: `world` ballots per step)
            if (only >= 0 && d != (uint32_t)only) continue;
            const unsigned long long mm = __ballot(dest[j] == d);
            if (dest[j] == d) rank[j] = s_wave[wave][d] + (uint32_t)__popcll(mm & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && mm) s_wave[wave][d] += (uint32_t)__popcll(mm);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (threadIdx.x < m.world) {
        uint32_t run = 0;
        for (int k = 0; k < THREADS / 64; ++k) {
            const uint32_t c = s_wave[k][threadIdx.x];
            s_wave[k][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = first + j * 64, d = dest[j];
        if (pos < n && (only < 0 || d == (uint32_t)only)) {
            const uint32_t at = s_start[d] + s_tile[d] + s_wave[wave][d] + rank[j];
            if (SPLIT) {
                dst.ptr[d][at] = slot[j];
            } else {
                out_slot[at] = slot[j];
                if (out_pos) out_pos[at] = pos;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// `only` >= 0 in ONE pass over the ids: count, offset and scatter in the same kernel.  A tile's offset in the
// output is the number of kept requests in the tiles before it -- a single running sum, so tiles chain through one
// status word each (decoupled look-back: a tile publishes its own count at once, then sums its predecessors' words,
// 64 per round trip, until it meets one that already carries an inclusive prefix).  Words are tagged with the
// call's sequence number: nothing to clear between calls.  A tile waits for lower-numbered tiles only, which the
// dispatcher started earlier (blocks of a grid are dispatched in index order: the assumption the radix sort's
// look-back and the direct stores of k_eval_sorted rest on as well, guarded by the same watchdog, tc::SpinGuard).
// ONE_PASS_TILES bounds the status array; bigger batches take the three kernels above.
// ---------------------------------------------------------------------------
constexpr uint32_t ONE_PASS_TILES = 1024;
// status word: sequence number (30 bits) | flag (2 bits) | value (32 bits)
constexpr unsigned long long ST_PARTIAL = 1ull << 32, ST_INCLUSIVE = 2ull << 32, ST_FLAGS = 3ull << 32, ST_VALUE = 0xFFFFFFFFull;
constexpr uint32_t SEQ_MASK = 0x3FFFFFFFu;

template <int IT>
__global__ __launch_bounds__(THREADS) void k_route_one(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w, uint32_t only,
                                                        unsigned long long* __restrict__ status, uint32_t seq,
                                                        uint32_t* __restrict__ out_slot, uint32_t* __restrict__ out_pos,
                                                        unsigned long long* __restrict__ violations) {
    __shared__ uint32_t s_c[MAX_WORLD];
    __shared__ uint32_t s_wcnt[THREADS / 64];
    __shared__ uint32_t s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < MAX_WORLD) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint32_t first = tile * (THREADS * IT) + wave * (64 * IT) + lane; // wave-striped: request order inside a wave
    const int bits = m.world > 1 ? 32 - __clz((int)m.world - 1) : 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t v[IT], slot[IT], rank[IT];
    bool keep[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) v[j] = first + j * 64 < n ? id[first + j * 64] : 0u;
    uint32_t mine = 0; // kept by my wave so far (wave-uniform)
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const bool valid = first + j * 64 < n;
        uint32_t owner;
        route_of(m, v[j], owner, slot[j]);
        // requests per destination (the totals): one LDS atomic per destination and wave step
        unsigned long long mm = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            const unsigned long long bb = __ballot((owner >> b) & 1u);
            mm &= ((owner >> b) & 1u) ? bb : ~bb;
        }
        if (valid && (mm & lt) == 0ull) atomicAdd(&s_c[owner], (uint32_t)__popcll(mm));
        keep[j] = valid && owner == only;
        const unsigned long long km = __ballot(keep[j]);
        rank[j] = mine + (uint32_t)__popcll(km & lt);
        mine += (uint32_t)__popcll(km);
    }
    if (lane == 0) s_wcnt[wave] = mine;
    __syncthreads();
    if (threadIdx.x < m.world) __hip_atomic_store(&w.tile_cnt[(size_t)threadIdx.x * w.tiles + tile], s_c[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t before = 0, total = 0;
    for (int k = 0; k < THREADS / 64; ++k) {
        if (k < wave) before += s_wcnt[k];
        total += s_wcnt[k];
    }
    if (wave == 0) {
        const unsigned long long tagged = (unsigned long long)(seq & SEQ_MASK) << 34;
        if (lane == 0)
            __hip_atomic_store(&status[tile], tagged | (tile == 0 ? ST_INCLUSIVE : ST_PARTIAL) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int at = (int)tile - 1; // nearest predecessor not yet summed
        tc::SpinGuard guard;
        while (at >= 0) {
            if (tc::spin_expired(guard)) { // (flagged, never hung)
                if (lane == 0) tc::invariant_failed(violations);
                break;
            }
            const int t = at - lane;
            const unsigned long long sv = t >= 0 ? __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tagged | ST_INCLUSIVE);
            const bool ready = (uint32_t)(sv >> 34) == (seq & SEQ_MASK) && (sv & ST_FLAGS) != 0ull;
            const unsigned long long rm = __ballot(ready), im = __ballot(ready && (sv & ST_FLAGS) == ST_INCLUSIVE);
            const int run = rm == ~0ull ? 64 : __builtin_ctzll(~rm);            // lanes 0 .. run-1: words that are there
            const int stop = im ? __builtin_ctzll(im) : 64;                       // the first inclusive prefix among them
            const int take = stop < run ? stop + 1 : run;                         // words that can be summed now
            uint32_t part = lane < take ? (uint32_t)(sv & ST_VALUE) : 0u;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
            excl += __shfl(part, 0, 64);
            if (stop < run) break;
            at -= take;
            if (take == 0) __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) {
            s_excl = excl;
            if (tile != 0)
                __hip_atomic_store(&status[tile], tagged | ST_INCLUSIVE | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const uint32_t base = s_excl + before;
#pragma unroll
    for (int j = 0; j < IT; ++j)
        if (keep[j]) {
            out_slot[base + rank[j]] = slot[j];
            if (out_pos) out_pos[base + rank[j]] = first + j * 64;
        }
}

// ---------------------------------------------------------------------------
// EVERY destination into a buffer of its own (the exchange: out_dst) in ONE pass over the ids -- count | scan | scatter were
// three launches on the stream that also groups batches.  A request's place in its destination's buffer is the number of
// requests for that destination in the tiles before + in the waves of this tile before + in this wave before: the first of
// the three is a running sum per destination over the tiles.  Every tile publishes its `world` counts in tagged words (the
// call's sequence number: nothing to clear) and sums the words of the tiles before it directly -- wave k takes destinations
// k, k + 4, ..., its lanes the tiles, spinning on words that are not there yet (lower-numbered tiles only, which the
// dispatcher started earlier; the watchdog as everywhere).  The last tile leaves the totals for k_route_publish.
// status: [world][tiles] words of seq << 32 | count.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(THREADS) void k_route_split_one(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w,
                                                             unsigned long long* __restrict__ status, uint32_t seq, SplitOut dst,
                                                             unsigned long long* __restrict__ violations) {
    __shared__ uint32_t s_wave[THREADS / 64][MAX_WORLD]; // per-wave counts -> exclusive prefix over the waves
    __shared__ uint32_t s_before[MAX_WORLD];             // this destination's requests in the tiles before
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x, tiles = gridDim.x;
    for (uint32_t i = threadIdx.x; i < (THREADS / 64) * MAX_WORLD; i += THREADS) (&s_wave[0][0])[i] = 0;
    __syncthreads();
    // wave-striped: step j of wave k covers requests tile*TILE + k*1024 + j*64 + lane (request order inside a wave)
    const uint32_t first = tile * TILE + wave * (64 * ITEMS) + lane;
    uint32_t dest[ITEMS], slot[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = first + j * 64;
        route_of(m, pos < n ? id[pos] : 0u, dest[j], slot[j]);
        if (pos >= n) dest[j] = 0xFFFFFFFFu;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        rank[j] = 0;
        for (uint32_t d = 0; d < m.world; ++d) { // (wave-uniform loop: `world` ballots per step)
            const unsigned long long mm = __ballot(dest[j] == d);
            if (dest[j] == d) rank[j] = s_wave[wave][d] + (uint32_t)__popcll(mm & lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && mm) s_wave[wave][d] += (uint32_t)__popcll(mm);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    const unsigned long long tagged = (unsigned long long)seq << 32;
    if (threadIdx.x < m.world) {
        uint32_t run = 0;
        for (int k = 0; k < THREADS / 64; ++k) {
            const uint32_t c = s_wave[k][threadIdx.x];
            s_wave[k][threadIdx.x] = run;
            run += c;
        }
        __hip_atomic_store(&status[(size_t)threadIdx.x * tiles + tile], tagged | run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_before[threadIdx.x] = run; // (this tile's own count for now: the last tile adds it to the sum below)
    }
    __syncthreads();
    {
        tc::SpinGuard guard;
        bool gave_up = false;
        for (uint32_t d = wave; d < m.world && !gave_up; d += THREADS / 64) {
            const unsigned long long* row = status + (size_t)d * tiles;
            uint32_t sum = 0;
            for (uint32_t t0 = 0; t0 < tile && !gave_up; t0 += 64) {
                const uint32_t t = t0 + lane;
                unsigned long long sv = t < tile ? __hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tagged;
                while ((uint32_t)(sv >> 32) != seq) { // a tile dispatched before mine that has not counted yet
                    if (tc::spin_expired(guard)) { // (flagged, never hung)
                        tc::invariant_failed(violations);
                        gave_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    sv = __hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                gave_up = __any(gave_up);
                sum += (uint32_t)sv;
            }
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
            if (lane == 0) {
                const uint32_t own = s_before[d];
                s_before[d] = sum;
                if (tile == tiles - 1) __hip_atomic_store(&w.totals[d], sum + own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (k_route_publish)
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = first + j * 64, d = dest[j];
        if (pos < n) dst.ptr[d][s_before[d] + s_wave[wave][d] + rank[j]] = slot[j];
    }
}

// Behind the one-pass router of ONE destination: the totals of every destination (block d sums row d of the tile counts) and,
// for a caller that polls host memory, the totals and then the tag -- written by whichever block is the last to finish
// (`world` <= 64 arrivals on one word), so that the tag follows every total.  One launch where k_route_scan + k_route_publish
// were two; the exclusive prefixes k_route_scan also leaves in tile_cnt are for k_route_scatter, which does not run here.
static __global__ __launch_bounds__(THREADS) void k_route_totals(Work w, uint32_t world, uint32_t* __restrict__ arrivals) {
    __shared__ uint32_t s_part[THREADS / 64];
    const uint32_t* row = w.tile_cnt + (size_t)blockIdx.x * w.tiles;
    uint32_t sum = 0;
    for (uint32_t t = threadIdx.x; t < w.tiles; t += THREADS) sum += __hip_atomic_load(&row[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int k = 0; k < THREADS / 64; ++k) total += s_part[k];
        __hip_atomic_store(&w.totals[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w.host_totals) {
            w.host_totals[blockIdx.x] = total;
            __threadfence_system();
            const uint32_t before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1u == world) {
                __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (for the next call on this scratch)
                w.host_totals[world] = w.tag;
                __threadfence_system();
            }
        }
    }
}

// A caller that polls host memory instead of synchronising: behind the scatter on the same stream, the totals and,
// after them, the caller's tag.  (A ticket taken by every tile of the scatter kernel -- "the last one publishes" --
// cost 100 ns per tile: 2048 device-scope atomics on one word are 0.2 ms.)
static __global__ __launch_bounds__(MAX_WORLD) void k_route_publish(Work w, uint32_t world) {
    if (threadIdx.x < world) w.host_totals[threadIdx.x] = __hip_atomic_load(&w.totals[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        w.host_totals[world] = w.tag;
        __threadfence_system();
    }
}

} // namespace rt
