// range_part.hpp -- round 6: the first half of the range path rebuilt, and HOT SLOTS peeled out of it.
//
// What was there (radix_sort.hpp, rs::k_tile_ranges): one 256-thread block per tile of 4 096 requests -- one wave per SIMD, every
// phase (loads, ballot ranking, prefix, staging, write-out) exposed at its full latency: 16 us to move 12 MB.  And a stream with
// hot keys (Zipf: the hottest of 10 M keys takes 11 % of a batch) could not use the path at all: the range that holds the key does
// not fit the block that finishes it, so such streams paid a histogram launch and three look-back passes (71 us of grouping
// kernels per batch against 37).
//
// k_tile_part<MODE>  the same tile of 4 096 requests on 1 024 threads (16 waves, 4 requests per lane): four waves per SIMD
//                    interleave their ranking chains and their memory waits; 512 ranges.  PART_RANK: a request whose slot is in
//                    the set's hot table (<= HOT_MAX slots, looked up in an LDS hash) is ranked in a bucket of its own, NR + hot
//                    id, instead of its key range -- and NOT written back with the tile: it leaves hot id << 16 | rank inside
//                    its tile in hot_info[i].  (LDS: 42 KB plain, 76 KB hot: a block shares its CU with one of rs::k_finish.)
// rs::k_finish       512 blocks of 512 threads finish the ranges as before (with the hot slots gone every range is small
//                    again); 16 more scan the hot ids' columns of the table: hot_P[tile][id] = the slot's requests in earlier
//                    tiles, hot_n[id] = its requests in the batch.
// the evaluation     ev::k_eval_sorted_lean's HOT ROLE walks the batch in request order: a hot request's rank among its slot's
//                    requests is hot_P + its rank inside its tile -- all the closed form of a run needs.  Nothing is gathered.
//
// This RANK FORM serves the batches the lean kernel serves (one timestamp, one quantity, decisions only, every run regular):
// BASELINE configs[2].  Other batches of a skewed stream stay on the LSD passes.  A GATHER FORM -- the hot buckets written
// back with the tile and copied behind the ranges' elements by a third kernel, so that any evaluation kernel could read
// them -- was built first, exact on every distribution of tools/partbench (profiles/r06_v3_partbench_gather_form.txt), and
// removed: pipelined it cost 45.0 us per Zipf batch against 40.2 for the LSD passes (three heavy kernels per chain,
// profiles/r06_v7_gather_form_pipelined.txt), and under k_eval_general a batch took 476 us against 87 (the hot runs side by
// side at the end of the batch fill the chip with waves waiting in their chains, profiles/r06_v15_hot_forms_driver_ab.txt).
//
// Who is hot is decided by the HOST from what the evaluation reports (ev::heavy_note: every run of at least heavy_min requests
// notes slot | length in a small device table, copied to pinned memory now and then, never waited for); the hot list
// travels to the device BY VALUE in k_hot_install's arguments (2 KB), so no host buffer has to outlive an enqueue.
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
// MODE: PART_PLAIN (ranges only), PART_RANK (hot slots ranked but NOT written back: request i of a hot slot leaves hot id << 16 |
// rank inside its tile in hot_info[i], every other request HOT_NONE; the tile's cold requests are written back alone)
constexpr int PART_PLAIN = 0, PART_RANK = 2;
constexpr uint32_t HOT_NONE = 0xFFFFFFFFu;
// ILV: the ranges interleaved chunk by chunk (radix_sort.hpp: string mode)
template <int MODE, bool ILV = false>
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
        uint32_t d = valid ? (ILV ? rs::ilv_digit(key[j]) : __umulhi(key[j], msd_mul)) : 0u;
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

} // namespace rp
