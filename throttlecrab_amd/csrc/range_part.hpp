// range_part.hpp -- round 6: the first half of the range path rebuilt, and HOT SLOTS peeled out of it.
//
// What was there (radix_sort.hpp, rs::k_tile_ranges): one 256-thread block per tile of 4 096 requests -- one wave per SIMD, every
// phase (loads, ballot ranking, prefix, staging, write-out) exposed at its full latency: 16 us to move 12 MB.  And a stream with
// hot keys (Zipf: the hottest of 10 M keys takes 11 % of a batch) could not use the path at all: the range that holds the key does
// not fit the block that finishes it, so such streams paid a histogram launch and three look-back passes (71 us of grouping
// kernels per batch against 37).
//
// k_tile_part<HOT>   the same tile of 4 096 requests on 1 024 threads (16 waves, 4 requests per lane): four waves per SIMD
//                    interleave their ranking chains and their memory waits.  HOT: a request whose slot is in the set's hot table
//                    (<= HOT_MAX slots, looked up in an LDS hash) is ranked into a bucket of its own, NR + hot id, instead of its
//                    key range: the tile is written back grouped by (range | hot id), index order kept, with a table row of
//                    NR + HOT_MAX words.  (LDS: 42 KB plain, 76 KB hot: a block shares its CU with one of rs::k_finish.)
// k_hot_gather       the hot buckets need no sorting -- a hot slot's requests in index order are its pieces in tile order -- only
//                    copying, to the HOT REGION behind the ranges' elements: position = sum of the ranges' sizes + sizes of the hot
//                    ids before it + the pieces of earlier tiles.  The copy is cut by elements (2 048 to a block), not by cells.
// rs::k_finish       unchanged but for the table's row stride: with the hot slots gone every range is small again.
//
// The grouped batch is then: [ranges' elements, sorted by (slot, index)] [hot id 0's requests by index] [hot id 1's] ...  Every
// slot's requests are contiguous and in index order -- all the evaluation kernels rely on (eval_kernels.hpp: a run's start is
// found by galloping back over equal slots, not by a search that assumes one sorted sequence).  Batches that promise grouped
// ROWS (TC_B_GROUPED_OUTPUT) never take the hot form.
//
// Who is hot is decided by the HOST from what the evaluation reports (ev::heavy_note: every run of at least heavy_min requests
// leaves slot | length | batch tag in a small device table, copied to pinned memory now and then, never waited for); the hot list
// travels to the device BY VALUE in k_hot_install's arguments (3.8 KB), so no host buffer has to outlive an enqueue.
// Reference workloads with hot keys: throttlecrab-server/benches/store_performance.rs:131-140, examples/access_patterns.rs:43-44.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_sort.hpp"

namespace rp {

constexpr int PT_THREADS = 1024;
constexpr int PT_WAVES = PT_THREADS / 64;
constexpr int PT_ITEMS = 4;
constexpr uint32_t PT_TILE = PT_THREADS * PT_ITEMS; // 4096
constexpr uint32_t NR = rs::NRANGE;                  // key ranges (512)
constexpr uint32_t HOT_MAX = 512;                    // hot slots per table (the list is a kernel argument: 2 KB; a Zipf(1.1) stream's 513th key
                                                     // takes ~130 of 1 Mi requests: what stays in the ranges is within the counting finish's reach)
constexpr uint32_t NB_HOT = NR + HOT_MAX;            // buckets of a hot partition = words of its table row
constexpr uint32_t HASH_BITS = 10, HASH_SIZE = 1u << HASH_BITS;
constexpr unsigned long long HASH_EMPTY = ~0ull;
static_assert(PT_TILE <= 65535, "a tile's starts and counts are packed into 16 bits each");
static_assert(HOT_MAX * 2 <= HASH_SIZE, "the hash stays at most half full");

__host__ __device__ inline uint32_t hot_hash(uint32_t slot) { return (slot * 0x9E3779B1u) >> (32 - HASH_BITS); }

// the hot table of one scratch set (device memory)
struct HotDev {
    uint32_t count;
    uint32_t pad[15];
    uint32_t slot[HOT_MAX];             // hot id -> slot
    unsigned long long hash[HASH_SIZE]; // slot << 32 | hot id, open addressing, linear probing
};
// ... as the host hands it over: by value
struct HotList {
    uint32_t count;
    uint32_t slot[HOT_MAX];
};
static_assert(sizeof(HotList) <= 3900, "fits the kernel argument segment beside a pointer");
static_assert(HOT_MAX <= PT_THREADS / 2 && HASH_SIZE * sizeof(unsigned long long) == (PT_THREADS / 2) * 16, "k_hot_install: a slot per thread; k_tile_part: 16 hash bytes per thread of the first half");

static __global__ __launch_bounds__(PT_THREADS) void k_hot_install(HotList l, HotDev* __restrict__ d) {
    __shared__ unsigned long long s_hash[HASH_SIZE];
    for (uint32_t i = threadIdx.x; i < HASH_SIZE; i += PT_THREADS) s_hash[i] = HASH_EMPTY;
    __syncthreads();
    if (threadIdx.x < l.count) {
        const uint32_t slot = l.slot[threadIdx.x];
        uint32_t h = hot_hash(slot);
        while (atomicCAS(&s_hash[h], HASH_EMPTY, ((unsigned long long)slot << 32) | threadIdx.x) != HASH_EMPTY) h = (h + 1u) & (HASH_SIZE - 1u);
        d->slot[threadIdx.x] = slot;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HASH_SIZE; i += PT_THREADS) d->hash[i] = s_hash[i];
    if (threadIdx.x == 0) d->count = l.count;
}

// ---------------------------------------------------------------------------
// every tile partitioned by (key range | hot id) in place + its table row
// ---------------------------------------------------------------------------
// table[tile * stride + b] = (elements of bucket b in the tile) << 16 | where they start inside the tile;
// totals[b] += elements of bucket b (zero on entry: rs::k_finish of the set's previous batch cleared it).
// `fill` (TC_B_OUTPUTS_IDLE batches): the decision bytes, preset here (rs::k_hist).
// MODE: PART_PLAIN (ranges only), PART_GATHER (hot slots in buckets of their own, written back with the tile: k_hot_gather
// follows), PART_RANK (hot slots ranked but NOT written back: request i of a hot slot leaves hot id << 16 | rank inside its tile
// in hot_info[i], every other request HOT_NONE; the tile's cold requests are written back alone -- for the evaluation's hot role,
// eval_kernels.hpp, which needs a hot request's rank among its slot's requests and nothing else)
constexpr int PART_PLAIN = 0, PART_GATHER = 1, PART_RANK = 2;
constexpr uint32_t HOT_NONE = 0xFFFFFFFFu;
template <int MODE>
__global__ __launch_bounds__(PT_THREADS) void k_tile_part(const uint32_t* __restrict__ slot_in, uint64_t* __restrict__ elem_out,
                                                          uint32_t* __restrict__ table, uint32_t stride, uint32_t* __restrict__ totals, uint32_t n,
                                                          uint32_t cap, uint32_t msd_mul, uint8_t* __restrict__ fill, uint32_t fill_value,
                                                          const HotDev* __restrict__ hot, uint32_t* __restrict__ hot_info) {
    constexpr bool HOT = MODE != PART_PLAIN;
    constexpr uint32_t NB = HOT ? NB_HOT : NR;
    constexpr int DB = HOT ? 10 : 9; // bits of a bucket number
    static_assert(NB <= (1u << DB) && NB <= 2u * PT_THREADS, "bucket numbers fit DB bits; two buckets per thread at most");
    __shared__ uint16_t s_cnt[PT_WAVES][NB]; // per-wave bucket counters -> exclusive prefix over the waves
    __shared__ uint32_t s_tstart[NB];        // where each bucket starts inside the tile
    __shared__ uint64_t s_elem[PT_TILE];     // the tile in grouped order (coalesced write-out)
    __shared__ uint32_t s_scan[2][PT_WAVES];
    __shared__ unsigned long long s_hash[HOT ? HASH_SIZE : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
    uint32_t key[PT_ITEMS], rank[PT_ITEMS], dig[PT_ITEMS];
    const uint32_t wbase = tile * PT_TILE + wave * 64 * PT_ITEMS + lane; // wave-striped: (wave, item, lane) order is index order
#pragma unroll
    for (int j = 0; j < PT_ITEMS; ++j) key[j] = slot_in[min(wbase + j * 64, n - 1u)]; // (raw: nothing here waits for a load)
    // every wave clears its own counter row (as 32-bit words)
    {
        uint32_t* row = reinterpret_cast<uint32_t*>(&s_cnt[wave][0]);
        for (uint32_t i = lane; i < NB / 2; i += 64) row[i] = 0;
    }
    if (HOT) {
        // the hot table's hash: 8 KB, 16 bytes per thread of the first half
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hot->hash);
        if (threadIdx.x < PT_THREADS / 2) reinterpret_cast<ulonglong2*>(s_hash)[threadIdx.x] = src[threadIdx.x];
    }
    if (fill != nullptr) {
        const uint32_t v4 = fill_value * 0x01010101u;
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)fill & 15u)) & 15u);
        const uint32_t h = head < n ? head : n;
        uint4* f16 = reinterpret_cast<uint4*>(fill + h);
        const uint32_t n16 = (n - h) / 16u;
        for (uint32_t i = blockIdx.x * PT_THREADS + threadIdx.x; i < n16; i += gridDim.x * PT_THREADS) f16[i] = make_uint4(v4, v4, v4, v4);
        if (blockIdx.x == 0) {
            for (uint32_t i = threadIdx.x; i < h; i += PT_THREADS) fill[i] = (uint8_t)fill_value;
            for (uint32_t i = h + n16 * 16u + threadIdx.x; i < n; i += PT_THREADS) fill[i] = (uint8_t)fill_value;
        }
    }
    if (HOT) __syncthreads(); // (the hash is complete)
    else __builtin_amdgcn_wave_barrier();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < PT_ITEMS; ++j) {
        const bool valid = (wbase + j * 64) < n;
        key[j] = rs::clamp_slot(key[j], cap);
        uint32_t d = valid ? __umulhi(key[j], msd_mul) : 0u;
        if (HOT && valid) {
            uint32_t h = hot_hash(key[j]);
            while (true) {
                const unsigned long long e = s_hash[h];
                if ((uint32_t)(e >> 32) == key[j]) {
                    d = NR + (uint32_t)e;
                    break;
                }
                if (e == HASH_EMPTY) break;
                h = (h + 1u) & (HASH_SIZE - 1u);
            }
        }
        dig[j] = d;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        // m = lanes of this wave (this round) in the same bucket
        const uint32_t before = valid ? (uint32_t)s_cnt[wave][d] : 0u; // all lanes read, then the leader adds
        rank[j] = before + (uint32_t)__popcll(m & lt);
        if (valid && (m & lt) == 0ull) s_cnt[wave][d] = (uint16_t)(before + (uint32_t)__popcll(m));
        __builtin_amdgcn_wave_barrier(); // (same-wave LDS operations execute in program order)
    }
    __syncthreads();
    {
        // bucket b = threadIdx.x (+ PT_THREADS): prefix over the waves -> the tile's count; then the buckets' starts
        uint32_t run[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t b = threadIdx.x + u * PT_THREADS;
            if (b < NB) {
                uint32_t acc = 0;
#pragma unroll
                for (int w = 0; w < PT_WAVES; ++w) {
                    const uint32_t c = s_cnt[w][b];
                    s_cnt[w][b] = (uint16_t)acc;
                    acc += c;
                }
                run[u] = acc;
            }
        }
        uint32_t incl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint32_t v = run[u];
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(v, off, 64);
                if (lane >= off) v += o;
            }
            incl[u] = v;
            if (lane == 63) s_scan[u][wave] = v;
        }
        __syncthreads();
        uint32_t carry0 = 0, total0 = 0, carry1 = 0;
#pragma unroll
        for (int w = 0; w < PT_WAVES; ++w) {
            const uint32_t a = s_scan[0][w];
            if (w < wave) carry0 += a;
            total0 += a;
            if (NB > PT_THREADS && w < wave) carry1 += s_scan[1][w];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t b = threadIdx.x + u * PT_THREADS;
            if (b < NB) {
                const uint32_t start = (u == 0 ? carry0 : total0 + carry1) + incl[u] - run[u];
                s_tstart[b] = start;
                table[(size_t)tile * stride + b] = (run[u] << 16) | start;
                // the buckets' sizes over the whole batch; nothing here waits for it  (PART_RANK: the hot ids' sizes come out of
                // rs::k_finish's scan of the table)
                if (run[u] && (MODE != PART_RANK || b < NR)) atomicAdd(&totals[b], run[u]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT_ITEMS; ++j) {
        const uint32_t pos = wbase + j * 64;
        if (pos < n) {
            const uint32_t in_bucket = (uint32_t)s_cnt[wave][dig[j]] + rank[j];
            if (MODE == PART_RANK) {
                const bool is_hot = dig[j] >= NR;
                hot_info[pos] = is_hot ? ((dig[j] - NR) << 16) | in_bucket : HOT_NONE;
                if (is_hot) continue;
            }
            s_elem[s_tstart[dig[j]] + in_bucket] = ((uint64_t)key[j] << 32) | pos;
        }
    }
    __syncthreads();
    const uint32_t tile_first = tile * PT_TILE;
    uint32_t nvalid = (n - tile_first) < PT_TILE ? (n - tile_first) : PT_TILE;
    if (MODE == PART_RANK) nvalid = s_tstart[NR]; // (the cold buckets come first: the tile's cold requests are [0, start of hot id 0))
#pragma unroll
    for (int j = 0; j < PT_ITEMS; ++j) {
        const uint32_t i = j * PT_THREADS + threadIdx.x;
        if (i < nvalid) elem_out[tile_first + i] = s_elem[i];
    }
}

// ---------------------------------------------------------------------------
// the hot buckets: pieces in tile order -> the hot region
// ---------------------------------------------------------------------------
// The hot region is the hot ids' requests one id after the other, each id's pieces in tile order: element e of the region
// belongs to one (hot id, tile) CELL of the table, and where it goes is simply e.  So the copy is cut by ELEMENTS, not by
// cells: the ids form UNITS (the host orders the list heaviest first: ids 0..HG_SINGLES-1 a unit each, the rest `group` ids to
// a unit, so that a unit's cells -- ids x tiles -- fit LDS), a unit's elements are cut into chunks of HG_CHUNK, and block b
// takes the b-th chunk of the whole region: it finds its unit from the buckets' sizes, loads the unit's cells, scans their
// counts, and every thread finds the cells of its 8 elements by bisection -- 8 independent gathers per lane, writes
// coalesced.  (A first version cut the work into fixed rectangles of cells and copied cell by cell: one wave ended up with
// all the long pieces of a heavy id -- 47 us per Zipf batch, 242 us for a batch of one key.)  The grid is an upper bound
// (hg_grid); blocks beyond the region leave at once.
constexpr int HG_THREADS = 256;
constexpr uint32_t HG_PER = 8, HG_CHUNK = HG_THREADS * HG_PER; // elements per block
constexpr uint32_t HG_SINGLES = 64;                             // the heaviest ids: a unit each
constexpr uint32_t HG_CELLS = 4096;                             // cells of a unit (ids x tiles) held in LDS
constexpr uint32_t HG_UNITS_MAX = 2 * HG_THREADS;
__host__ __device__ inline uint32_t hg_group(uint32_t tiles) { return tiles <= 256u ? 16u : (tiles <= 512u ? 8u : 4u); } // (tiles <= rs::FIN_THREADS = 1024)
__host__ __device__ inline uint32_t hg_units(uint32_t count, uint32_t group) {
    return count <= HG_SINGLES ? count : HG_SINGLES + (count - HG_SINGLES + group - 1u) / group;
}
__host__ __device__ inline uint32_t hg_grid(uint32_t n, uint32_t count, uint32_t group) { return n / HG_CHUNK + hg_units(count, group) + 1u; }
static_assert(HG_SINGLES + (HOT_MAX - HG_SINGLES + 3) / 4 <= HG_UNITS_MAX, "two units per thread at most");
// (what hot_refresh's "same list?" goes by: who is among the singles)
constexpr uint32_t HG_A_IDS = 8, HG_B_END = HG_SINGLES;

static __global__ __launch_bounds__(HG_THREADS) void k_hot_gather(const uint64_t* __restrict__ tiled, const uint32_t* __restrict__ table, uint32_t stride,
                                                                  const uint32_t* __restrict__ totals, uint64_t* __restrict__ out, uint32_t tiles,
                                                                  uint32_t tile_len, const HotDev* __restrict__ hot) {
    __shared__ uint32_t s_unit[HG_UNITS_MAX];      // elements per unit
    __shared__ uint32_t s_uchunk[HG_UNITS_MAX + 1]; // chunks before unit u
    __shared__ uint32_t s_uoff[HG_UNITS_MAX + 1];   // elements before unit u
    __shared__ uint32_t s_w[2][HG_THREADS / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_cell[HG_CELLS];           // count << 16 | start inside the tile, id-major
    __shared__ uint32_t s_pre[HG_CELLS + 1];        // elements of the unit before the cell
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t hc = hot->count, group = hg_group(tiles), units = hg_units(hc, group);
    if (hc == 0u) return;
    for (uint32_t u = threadIdx.x; u < HG_UNITS_MAX; u += HG_THREADS) s_unit[u] = 0;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    // the buckets' sizes: ranges -> where the hot region starts; hot ids -> their units  (every load first, then the sums: a
    // loop of load-then-add waits for each load in turn)
    {
        constexpr uint32_t TV = (NB_HOT + HG_THREADS - 1) / HG_THREADS;
        uint32_t tv[TV];
#pragma unroll
        for (uint32_t k = 0; k < TV; ++k) {
            const uint32_t i = threadIdx.x + k * HG_THREADS;
            tv[k] = i < NR + hc ? totals[i] : 0u;
        }
        uint32_t r = 0;
#pragma unroll
        for (uint32_t k = 0; k < TV; ++k) {
            const uint32_t i = threadIdx.x + k * HG_THREADS;
            if (i < NR) r += tv[k]; // (compile-time per round: NR is a multiple of HG_THREADS)
            else if (tv[k]) {
                const uint32_t h = i - NR;
                atomicAdd(&s_unit[h < HG_SINGLES ? h : HG_SINGLES + (h - HG_SINGLES) / group], tv[k]);
            }
        }
        static_assert(NR % HG_THREADS == 0, "a round of loads is ranges or hot ids, not both");
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        if (lane == 0 && r) atomicAdd(&s_base, r);
    }
    __syncthreads();
    // exclusive scans over the units (two per thread): chunks and elements
    {
        const uint32_t u0 = 2u * threadIdx.x;
        const uint32_t e0 = s_unit[u0], e1 = s_unit[u0 + 1u];
        const uint32_t c0 = (e0 + HG_CHUNK - 1u) / HG_CHUNK, c1 = (e1 + HG_CHUNK - 1u) / HG_CHUNK;
        uint32_t ci = c0 + c1, ei = e0 + e1;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t oc = __shfl_up(ci, off, 64), oe = __shfl_up(ei, off, 64);
            if (lane >= off) ci += oc, ei += oe;
        }
        if (lane == 63) s_w[0][wave] = ci, s_w[1][wave] = ei;
        __syncthreads();
        uint32_t cc = 0, ce = 0;
        for (int w = 0; w < wave; ++w) cc += s_w[0][w], ce += s_w[1][w];
        s_uchunk[u0] = cc + ci - c0 - c1;
        s_uchunk[u0 + 1u] = cc + ci - c1;
        s_uoff[u0] = ce + ei - e0 - e1;
        s_uoff[u0 + 1u] = ce + ei - e1;
        if (threadIdx.x == HG_THREADS - 1) s_uchunk[HG_UNITS_MAX] = cc + ci, s_uoff[HG_UNITS_MAX] = ce + ei;
    }
    __syncthreads();
    if (blockIdx.x >= s_uchunk[HG_UNITS_MAX]) return; // (block-uniform: beyond the region)
    uint32_t u;
    {
        uint32_t lo = 0, hi = HG_UNITS_MAX; // the last unit whose chunks start at or before mine
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_uchunk[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        u = lo;
    }
    (void)units;
    const uint32_t chunk = blockIdx.x - s_uchunk[u], unit_n = s_unit[u], out0 = s_base + s_uoff[u];
    const uint32_t h0 = u < HG_SINGLES ? u : HG_SINGLES + (u - HG_SINGLES) * group, G = u < HG_SINGLES ? 1u : group;
    const uint32_t H = min(G, hc - h0), cells = H * tiles; // <= HG_CELLS
    // the unit's cells, id-major (the table holds a tile's ids side by side); all loads in flight at once
    {
        constexpr uint32_t CV = HG_CELLS / HG_THREADS;
        const uint32_t gs = G == 1u ? 0u : (G == 4u ? 2u : (G == 8u ? 3u : 4u)), gt = G * tiles;
        uint32_t cv[CV];
#pragma unroll
        for (uint32_t k = 0; k < CV; ++k) {
            const uint32_t i = threadIdx.x + k * HG_THREADS, t = i >> gs, hl = i & (G - 1u);
            cv[k] = (i < gt && hl < H) ? table[(size_t)t * stride + NR + h0 + hl] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < CV; ++k) {
            const uint32_t i = threadIdx.x + k * HG_THREADS, t = i >> gs, hl = i & (G - 1u);
            if (i < gt && hl < H) s_cell[hl * tiles + t] = cv[k];
        }
    }
    __syncthreads();
    {
        const uint32_t per = (cells + HG_THREADS - 1u) / HG_THREADS, c0 = min(threadIdx.x * per, cells), c1 = min(c0 + per, cells);
        uint32_t sum = 0;
        for (uint32_t i = c0; i < c1; ++i) sum += s_cell[i] >> 16;
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_w[0][wave] = incl;
        __syncthreads();
        uint32_t at = incl - sum;
        for (int w = 0; w < wave; ++w) at += s_w[0][w];
        for (uint32_t i = c0; i < c1; ++i) {
            s_pre[i] = at;
            at += s_cell[i] >> 16;
        }
        if (threadIdx.x == HG_THREADS - 1) s_pre[cells] = at; // (== unit_n)
    }
    __syncthreads();
    const uint32_t e0 = chunk * HG_CHUNK + threadIdx.x;
    uint64_t v[HG_PER];
#pragma unroll
    for (uint32_t k = 0; k < HG_PER; ++k) {
        const uint32_t e = e0 + k * HG_THREADS;
        v[k] = 0;
        if (e < unit_n) {
            uint32_t lo = 0; // the last cell that starts at or before e (cells without requests share their successor's start);
                             // a fixed number of steps without branches: the 8 searches of a lane interleave
#pragma unroll
            for (uint32_t step = HG_CELLS / 2u; step > 0u; step >>= 1) {
                const uint32_t probe = min(lo + step, cells); // (s_pre[cells] = the unit's size > e)
                lo = s_pre[probe] <= e ? probe : lo;
            }
            const uint32_t t = G == 1u ? lo : lo % tiles;
            v[k] = tiled[(size_t)t * tile_len + (s_cell[lo] & 0xFFFFu) + (e - s_pre[lo])];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < HG_PER; ++k) {
        const uint32_t e = e0 + k * HG_THREADS;
        if (e < unit_n) out[out0 + e] = v[k];
    }
}

} // namespace rp
