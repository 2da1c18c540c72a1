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

#ifndef RS_TIMING
#define RS_TIMING 0 // tools/sortbench.hip sets 1: one block of k_tile_ranges / k_finish stamps the 100 MHz wall clock at its phase boundaries
#endif
#if RS_TIMING
__device__ long long g_rs_stamps[2][16];
#define RS_STAMP(k, i, blk) do { if (threadIdx.x == 0 && blockIdx.x == (blk)) g_rs_stamps[k][i] = wall_clock64(); } while (0)
#else
#define RS_STAMP(k, i, blk) do { } while (0)
#endif

namespace rs {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int HIST_THREADS = 1024; // k_hist: few big blocks -> few global atomics per histogram line
constexpr int HIST_BLOCKS = 128;
constexpr int RADIX = 256;
constexpr int MAX_PASSES = 4;
constexpr int NRANGE = 512;                         // key ranges of the range path (round 6: was 256 = RADIX)
constexpr int HIST_ROWS = MAX_PASSES + NRANGE / RADIX; // histogram rows of a batch: one per LSD digit + the range digit's (k_hist counts it beside them)
constexpr int HIST_WORDS = HIST_ROWS * RADIX;
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
    uint32_t* hist;      // [HIST_ROWS][RADIX] digit histograms of this batch (zero on entry): the LSD digits' rows, then the range digit's NRANGE words
    uint32_t* hist_next; // the other parity's histograms: cleared here for the next batch
    uint32_t* ticket;    // [MAX_PASSES] dynamic tile ids (forward progress for the look-back)
    uint32_t* status;    // [MAX_PASSES] x { part[max_tiles] | gacc[max_groups] | gincl[max_groups] } x RADIX
    uint32_t max_tiles;
    uint32_t max_groups;
    unsigned long long* violations; // the engine's invariant counter (a look-back that outlasts tc::SPIN_LIMIT_TICKS)
};

// ---- the range path (round 4): two launches, no histogram pass, no chain between tiles -----------------------------
// A 1 Mi batch over 10 M slots needs 24 key bits = a histogram launch + three LSD passes, each pass a launch with a
// look-back chain (19 us alone, 25-35 us beside the other streams).  The range path cuts the key space into 256 equal
// RANGES (digit = umulhi(slot, mul): monotone in the slot, so range r holds the slots [range_lo(r), range_lo(r + 1))):
//   k_tile_ranges   every tile of the batch is partitioned by range IN PLACE (tile t's elements stay in [t * TILE, ...),
//                   grouped by range, index order kept) and writes a 256-entry table row: where each range starts in
//                   the tile and how many elements it has there.  Tiles do not depend on one another: no histogram
//                   first, no look-back.
//   k_finish        one 1024-thread block per range collects the range's pieces from all tiles (tile order = index
//                   order), sorts its <= FIN_CAP elements by their offset inside the range (<= 16 bits: two 8-bit
//                   passes with ballot ranking) in LDS and writes them where the range belongs in the sorted batch --
//                   the same sorted (slot << 32 | index) array the LSD passes produce.  The range's place is the sum
//                   of the earlier ranges' sizes: every block publishes its size at once and reads the others' (256
//                   co-resident blocks, one word each).
// A range that holds more than FIN_CAP elements (a skewed batch the host did not foresee: it chooses the path from the
// largest range of a RECENT batch, mirrored into pinned memory) is sorted by its block through global memory, slowly but
// exactly; streams that are skewed stay on the LSD passes.
// Round 6: 512 ranges finished by 512-thread blocks of HALF the LDS (73 KB against 147): two of them share a CU, or one shares it
// with a block of the partition (rp::k_tile_part, 42-76 KB).  With one 147 KB block per CU, no other grouping kernel could run
// beside a k_finish on ANY CU: the grouping chains of the three streams took turns (tools/host_bound.py: the device, not the
// host, held a Zipf batch at 54 us).
constexpr int FIN_THREADS = 512;
constexpr int FIN_WAVES = FIN_THREADS / 64;
constexpr int FIN_ITEMS = 8;
constexpr uint32_t FIN_CAP = FIN_THREADS * FIN_ITEMS; // elements of a range that are finished in LDS
constexpr uint32_t FIN_POS_BITS = 12;                  // FIN_CAP == 1 << FIN_POS_BITS
constexpr int FIN_ROW = RADIX + 1;                     // per-wave counter rows, padded against bank conflicts
constexpr int MSD_ROW = MAX_PASSES;                    // first histogram row of the range digit (NRANGE / RADIX rows) when k_hist counts it beside the LSD digits
static_assert(NRANGE == FIN_THREADS, "k_finish: one thread per range when the ranges' sizes are summed");
static_assert(FIN_CAP == (1u << FIN_POS_BITS), "positions inside a range must fit FIN_POS_BITS");
// k_finish, counting path: ranges of at most CNT_W_MAX slots whose slots hold at most CNT_DUP_MAX elements each are sorted by
// COUNTING (one LDS atomic per element on a byte counter per slot) instead of two ballot-ranked passes
constexpr uint32_t CNT_W_MAX = 24576;
constexpr uint32_t CNT_DUP_MAX = 255; // (round 6: was 16 -- behind the hot slots of a skewed stream come slots with dozens of requests each; the four counters of a word are summed with v_sad_u8, which does not care whether the sum fits a byte)
constexpr uint32_t FIN_UNION_WORDS = CNT_W_MAX / 4 + CNT_W_MAX / 8 + FIN_CAP; // >= 2 * FIN_CAP + FIN_WAVES * FIN_ROW + 2 * RADIX
static_assert(FIN_UNION_WORDS >= 2 * FIN_CAP + FIN_WAVES * FIN_ROW + 2 * RADIX, "the ballot path's arrays fit the union");

// digit = umulhi(slot, mul) with mul = floor(NRANGE * 2^32 / (cap + 1)): < NRANGE for every slot <= cap (cap itself is the
// sentinel of out-of-range slots).  Needs cap + 1 > NRANGE.
__host__ __device__ inline uint32_t range_mul(uint32_t cap) { return (uint32_t)(((uint64_t)NRANGE << 32) / ((uint64_t)cap + 1ull)); }
// smallest slot whose digit is >= d:  slot * mul >= d * 2^32
__host__ __device__ inline uint32_t range_lo(uint32_t d, uint32_t mul) { return (uint32_t)((((uint64_t)d << 32) + mul - 1u) / mul); }
// widest range (slots), conservatively
__host__ __device__ inline uint32_t range_width(uint32_t mul) { return (uint32_t)((1ull << 32) / mul) + 2u; }

// Round 6 (late), string mode: INTERLEAVED ranges.  A key table hands out neighbouring slots to the keys of one batch (its new keys
// pop consecutive entries of the free stack, a benchmark's hot keys were bound one after the other): a fifth of a configs[4] batch
// falls into a handful of the contiguous ranges above, and key batches stayed on the LSD passes.  With ILV the ranges take turns
// chunk by chunk -- a chunk is 8 neighbouring slots, one 128-byte line of 16-byte cells --: slot s lies in range (s >> 3) mod NRANGE,
// at offset ((s >> 12) << 3) | (s & 7) inside it.  Any stretch of consecutive slots spreads evenly over all ranges; requests of one
// slot still meet in one range, and a range is still finished in ascending order of its offsets.  The sorted batch is then grouped
// by slot, not ascending in it -- nothing that reads it needs more (the evaluations look at neighbours for EQUAL slots only).
// msd_mul = RANGE_ILV selects it (a plain multiplier is below 2^25: the path wants more than 65 536 slots).
constexpr uint32_t RANGE_ILV = 0x80000000u;
constexpr uint32_t ILV_CHUNK_BITS = 3, ILV_RANGE_BITS = 9;
static_assert(NRANGE == (1 << ILV_RANGE_BITS), "interleaved ranges: NRANGE is a power of two");
__host__ __device__ inline uint32_t ilv_digit(uint32_t slot) { return (slot >> ILV_CHUNK_BITS) & (uint32_t)(NRANGE - 1); }
__host__ __device__ inline uint32_t ilv_sub(uint32_t slot) {
    return ((slot >> (ILV_CHUNK_BITS + ILV_RANGE_BITS)) << ILV_CHUNK_BITS) | (slot & ((1u << ILV_CHUNK_BITS) - 1u));
}
__host__ __device__ inline uint32_t ilv_slot(uint32_t r, uint32_t sub) {
    return ((sub >> ILV_CHUNK_BITS) << (ILV_CHUNK_BITS + ILV_RANGE_BITS)) | (r << ILV_CHUNK_BITS) | (sub & ((1u << ILV_CHUNK_BITS) - 1u));
}
// offsets inside a range are below this (cap itself, the sentinel of out-of-range slots, included)
__host__ __device__ inline uint32_t ilv_width(uint32_t cap) { return ((cap >> (ILV_CHUNK_BITS + ILV_RANGE_BITS)) + 1u) << ILV_CHUNK_BITS; }

__host__ __device__ inline uint32_t groups_of(uint32_t tiles) { return (tiles + GROUP - 1) / GROUP; }
__host__ __device__ inline size_t pass_status_words(uint32_t max_tiles, uint32_t max_groups) {
    return ((size_t)max_tiles + 2 * (size_t)max_groups) * RADIX;
}
// words of the whole workspace (two histogram parities | tickets | status)
inline size_t workspace_words(uint32_t max_tiles) {
    return (size_t)2 * HIST_WORDS + MAX_PASSES + (size_t)MAX_PASSES * pass_status_words(max_tiles, groups_of(max_tiles));
}
inline Workspace carve(uint32_t* base, uint32_t parity, uint32_t max_tiles) {
    Workspace ws;
    ws.hist = base + (size_t)parity * HIST_WORDS;
    ws.hist_next = base + (size_t)(parity ^ 1u) * HIST_WORDS;
    ws.ticket = base + (size_t)2 * HIST_WORDS;
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
// `msd_mul` != 0: the range digit umulhi(slot, msd_mul) is counted beside the LSD digits, into the rows from MSD_ROW on, from
// which the first LSD pass mirrors the largest range to the host (the hint the range path is chosen by)
template <int NT>
__global__ __launch_bounds__(NT) void k_hist(const uint32_t* __restrict__ slot, uint32_t n, uint32_t cap,
                                                  int passes, Workspace ws, uint32_t tiles, const uint32_t* __restrict__ gate,
                                                  uint32_t gate_min, uint8_t* __restrict__ fill, uint32_t fill_value,
                                                  uint32_t msd_mul) {
    __shared__ uint32_t s_h[HIST_ROWS][RADIX];
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
            for (int i = threadIdx.x; i < HIST_WORDS; i += NT) ws.hist_next[i] = 0;
        return;
    }
    for (int i = threadIdx.x; i < HIST_WORDS; i += NT) (&s_h[0][0])[i] = 0;
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
        for (int i = threadIdx.x; i < HIST_WORDS; i += NT) ws.hist_next[i] = 0;
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
                if (msd_mul) atomicAdd(&(&s_h[MSD_ROW][0])[(msd_mul & RANGE_ILV) ? ilv_digit(kk) : __umulhi(kk, msd_mul)], 1u); // (NRANGE words: rows MSD_ROW ..)
            }
        }
        for (; i < n; i += stride) {
            const uint32_t kk = clamp_slot(slot[i], cap);
            for (int p = 0; p < passes; ++p) atomicAdd(&s_h[p][(kk >> (8 * p)) & 255u], 1u);
            if (msd_mul) atomicAdd(&(&s_h[MSD_ROW][0])[(msd_mul & RANGE_ILV) ? ilv_digit(kk) : __umulhi(kk, msd_mul)], 1u); // (NRANGE words: rows MSD_ROW ..)
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_WORDS; i += NT) {
        const int row = i / RADIX;
        if (!(row < passes || (msd_mul && row >= MSD_ROW))) continue;
        const uint32_t v = (&s_h[0][0])[i];
        if (v) atomicAdd(&ws.hist[i], v);
    }
}

// ---------------------------------------------------------------------------
// one LSD pass.  FIRST: input is the raw slot column (value = position).
// ---------------------------------------------------------------------------
// `range_hint` (block 0 of the first pass, when k_hist counted the ranges beside the LSD digits: row MSD_ROW):
// n << 32 | largest range, into pinned host memory.
template <int ITEMS, bool FIRST>
__global__ __launch_bounds__(THREADS) void k_onesweep(const uint32_t* __restrict__ slot_in,
                                                      const uint64_t* __restrict__ elem_in,
                                                      uint64_t* __restrict__ elem_out, uint32_t n, uint32_t cap,
                                                      int pass, Workspace ws, const uint32_t* __restrict__ gate, uint32_t gate_min,
                                                      unsigned long long* __restrict__ range_hint) {
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
    const bool ticketed = gridDim.x > RESIDENT_TILES; // (grid-uniform)
    if (ticketed && threadIdx.x == 0) s_tile = atomicAdd(&ws.ticket[pass], 1u);
    // every wave clears its own counter row: its ranking needs no block barrier
#pragma unroll
    for (int i = 0; i < RADIX / 64; ++i) s_wave[wave][i * 64 + lane] = 0;

    if (range_hint != nullptr && blockIdx.x == 0) { // (block-uniform)
        uint32_t m = 0;
        for (int i = threadIdx.x; i < NRANGE; i += THREADS) m = max(m, ws.hist[MSD_ROW * RADIX + i]);
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if (lane == 0) s_scan[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < WAVES; ++w) m = max(m, s_scan[w]);
            __hip_atomic_store(range_hint, ((unsigned long long)n << 32) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    }
    uint32_t tile = blockIdx.x;
    if (ticketed) {
        __syncthreads();
        tile = s_tile;
    }
    const uint32_t shift = 8u * (uint32_t)pass;

    // wave-striped tile: item j of lane l of wave w sits at tile*TILE + w*64*ITEMS + j*64 + l.  The loads go out first and
    // nothing waits for them until the ranking reaches their item: the histogram scan below runs while they are on their way
    // (round 4: the tile's loads were 4 of a pass's 16 us; the same change made k_tile_ranges 3.7 us shorter)
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    uint64_t raw[FIRST ? 1 : ITEMS];
    const uint32_t wbase = tile * TILE + wave * 64 * ITEMS + lane;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = min(wbase + j * 64, n - 1u);
        if (FIRST) key[j] = slot_in[pos];
        else raw[j] = elem_in[pos];
    }
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
        s_base[threadIdx.x] = carry + v - h; // (read after the look-back, several barriers from here)
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = wbase + j * 64;
        if (pos < n) {
            if (FIRST) {
                key[j] = clamp_slot(key[j], cap);
                val[j] = pos;
            } else {
                key[j] = (uint32_t)(raw[j] >> 32);
                val[j] = (uint32_t)raw[j];
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

// ---------------------------------------------------------------------------
// range path, second half: one block per range sorts the range by the slot's offset inside it
// ---------------------------------------------------------------------------
// a load that is served by the L2, not by this CU's vector cache (which may still hold the line from before this block's own
// stores to it went through)
__device__ __forceinline__ uint64_t ld_l2(const uint64_t* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rank the digits of one strip-distributed chunk: every wave ranks its own items with ballots against its counter row
// (as k_onesweep does), then the rows become exclusive prefixes over the waves and s_tot[] the chunk's digit counts.
// Wave w holds positions [w * strip, (w + 1) * strip) of the chunk, item j of lane l at w * strip + j * 64 + l, so
// (wave, item, lane) order is position order: the ranks are stable.  Ends with a barrier.
template <int ITEMS_>
__device__ __forceinline__ void fin_rank(const uint32_t (&dig)[ITEMS_], const bool (&valid)[ITEMS_], uint32_t (&rank)[ITEMS_],
                                         uint32_t (*s_cnt)[FIN_ROW], uint32_t* s_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < FIN_WAVES * FIN_ROW; i += FIN_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < ITEMS_; ++j) {
        if (__ballot(valid[j]) == 0ull) { // (wave-uniform: the strip ended)
            rank[j] = 0;
            continue;
        }
        const uint32_t d = dig[j];
        unsigned long long m = __ballot(valid[j]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t before = valid[j] ? s_cnt[wave][d] : 0u;
        rank[j] = before + (uint32_t)__popcll(m & lt);
        if (valid[j] && (m & lt) == 0ull) s_cnt[wave][d] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier(); // (same-wave LDS operations execute in program order)
    }
    __syncthreads();
    // exclusive prefix over the FIN_WAVES waves of every digit: FIN_WAVES consecutive lanes take one digit's column
    static_assert(FIN_WAVES == 8, "the column mapping below: threadIdx.x >> 3, & 7");
#pragma unroll
    for (int round = 0; round < RADIX / (FIN_THREADS / FIN_WAVES); ++round) {
        const int d = round * (FIN_THREADS / FIN_WAVES) + (threadIdx.x >> 3), w = threadIdx.x & (FIN_WAVES - 1);
        const uint32_t v = s_cnt[w][d];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < FIN_WAVES; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, FIN_WAVES);
            if (w >= off) incl += o;
        }
        s_cnt[w][d] = incl - v;
        if (w == FIN_WAVES - 1) s_tot[d] = incl;
    }
    __syncthreads();
}

// exclusive scan of s_tot[RADIX] into s_start[RADIX] (every thread of the block calls it; ends with a barrier)
__device__ __forceinline__ void fin_scan_digits(const uint32_t* s_tot, uint32_t* s_start, uint32_t* s_part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = 0, incl = 0;
    if (threadIdx.x < RADIX) {
        v = s_tot[threadIdx.x];
        incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_part[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < RADIX) {
        uint32_t carry = 0;
        for (int w = 0; w < wave; ++w) carry += s_part[w];
        s_start[threadIdx.x] = carry + incl - v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// range path, second half: one block per range
// ---------------------------------------------------------------------------
// `tiled`: k_tile_ranges' / rp::k_tile_part's output (tile size `tile_len`, `tiles` <= FIN_THREADS of them) with its `table` (rows of
// `stride` words: the ranges first; totals likewise, `totals_words` of them); `elem_out`: the batch
// sorted by (slot, index); `scratch`: a third array of the batch's size (ranges that do not fit LDS); `totals`: the ranges' sizes
// (k_tile_ranges), `totals_next`: the other parity's, cleared here for the set's next batch -- no block waits for another
// (a first version had every block publish its size and read the others': under load, blocks that had been dispatched
// early then sat on their CU waiting for the late ones, 41 us per launch in the pipelined run against 14 alone);
// sub_passes = 8-bit digits of the offset inside a range (1 or 2: the host takes this path only when the widest range
// is at most 65536 slots); `range_hint`: n << 32 | largest range, into pinned host memory (block NRANGE - 1).
// ILV: interleaved ranges (above) -- msd_mul then carries the capacity (the ranges' width follows from it)
template <bool ILV>
__global__ __launch_bounds__(FIN_THREADS) void k_finish(const uint64_t* __restrict__ tiled, const uint32_t* __restrict__ table,
                                                               uint64_t* __restrict__ elem_out, uint64_t* __restrict__ scratch,
                                                               const uint32_t* __restrict__ totals, uint32_t* __restrict__ totals_next, uint32_t n,
                                                               uint32_t tiles, uint32_t tile_len, uint32_t msd_mul, int sub_passes,
                                                               unsigned long long* __restrict__ range_hint, uint32_t stride,
                                                               uint32_t totals_words, uint32_t* __restrict__ hot_P, uint32_t* __restrict__ hot_n,
                                                               uint32_t hot_ids, unsigned long long* __restrict__ cold_hint) {
    __shared__ uint32_t s_idx[FIN_CAP];            // request index of position p (never moves)
    __shared__ uint32_t s_u[FIN_UNION_WORDS];      // the two ways of sorting a range share this
    // ballot path: offset inside the range << FIN_POS_BITS | p in the order reached so far (x2), per-wave digit counters, digit totals / starts
    uint32_t (*s_kv)[FIN_CAP] = reinterpret_cast<uint32_t (*)[FIN_CAP]>(s_u);
    uint32_t (*s_cnt)[FIN_ROW] = reinterpret_cast<uint32_t (*)[FIN_ROW]>(s_u + 2 * FIN_CAP);
    uint32_t* s_tot = s_u + 2 * FIN_CAP + FIN_WAVES * FIN_ROW;
    uint32_t* s_start = s_tot + RADIX;
    // counting path: a byte counter per slot of the range, where each word of four counters starts in the sorted range, the sorted range
    uint32_t* c_cnt = s_u;
    uint16_t* c_ws = reinterpret_cast<uint16_t*>(s_u + CNT_W_MAX / 4);
    uint32_t* c_fin = s_u + CNT_W_MAX / 4 + CNT_W_MAX / 8;
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_part[2 * FIN_WAVES];
    __shared__ uint32_t s_pos[FIN_THREADS + 1];    // where tile t's piece of my range starts among the range's elements
    __shared__ uint32_t s_st[FIN_THREADS];         // ... and inside the tile
    __shared__ uint32_t s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = blockIdx.x;
    if (r >= (uint32_t)NRANGE) {
        // Round 6, the rank form (range_part.hpp, PART_RANK): blocks behind the ranges' scan the hot ids' columns of the table --
        // hot_P[tile * hot_ids + id] = requests of hot id `id` in the tiles before `tile`, hot_n[id] = its requests in the
        // batch: what the evaluation's hot role adds to a request's rank inside its tile.  32 ids x 16 stretches of tiles per
        // block, the table's words of 32 neighbouring ids read and written side by side.
        uint32_t (*s_seg)[32] = reinterpret_cast<uint32_t (*)[32]>(s_idx);
        const uint32_t hl = threadIdx.x & 31u, seg = threadIdx.x >> 5, h = (r - (uint32_t)NRANGE) * 32u + hl;
        const uint32_t per = (tiles + 15u) / 16u, t0 = min(seg * per, tiles), t1 = min(t0 + per, tiles);
        uint32_t sum = 0;
        if (h < hot_ids) {
#pragma unroll 8
            for (uint32_t t = t0; t < t1; ++t) sum += table[(size_t)t * stride + NRANGE + h] >> 16;
        }
        s_seg[seg][hl] = sum;
        __syncthreads();
        uint32_t at = 0, total = 0;
        for (uint32_t q = 0; q < 16u; ++q) {
            const uint32_t v = s_seg[q][hl];
            if (q < seg) at += v;
            total += v;
        }
        if (h < hot_ids) {
#pragma unroll 8
            for (uint32_t t = t0; t < t1; ++t) {
                hot_P[(size_t)t * hot_ids + h] = at;
                at += table[(size_t)t * stride + NRANGE + h] >> 16;
            }
            if (seg == 0u) hot_n[h] = total;
        }
        return;
    }
    RS_STAMP(1, 0, 128);
    const uint32_t lo = ILV ? 0u : range_lo(r, msd_mul);
    const uint32_t width = ILV ? ilv_width(msd_mul) : range_lo(r + 1u, msd_mul) - lo; // slots of my range (every offset inside it is below this)
    auto sub_of = [&](uint32_t slot) -> uint32_t { return ILV ? ilv_sub(slot) : slot - lo; };
    auto slot_of = [&](uint32_t sub) -> uint32_t { return ILV ? ilv_slot(r, sub) : lo + sub; };
    // my range's pieces: one table word per tile; exclusive scan of the counts over the tiles
    uint32_t c, tot_early = 0;
    {
        uint32_t w = 0;
        if (threadIdx.x < tiles) w = table[(size_t)threadIdx.x * stride + r];
        tot_early = totals[threadIdx.x]; // (NRANGE == FIN_THREADS: a range's size per thread)
        // (while the two words are on their way: the counting path's counters)
        if (width <= CNT_W_MAX)
            for (uint32_t i = threadIdx.x; i < (width + 3u) / 4u; i += FIN_THREADS) c_cnt[i] = 0;
        const uint32_t cnt = w >> 16;
        RS_STAMP(1, 1, 128);
        if (cnt == 0xFFFFu) s_part[0] = 1; // (RS_TIMING: the table word has landed)
        RS_STAMP(1, 2, 128);
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t carry = 0, total = 0;
        for (int q = 0; q < FIN_WAVES; ++q) {
            if (q < wave) carry += s_part[q];
            total += s_part[q];
        }
        s_pos[threadIdx.x] = carry + incl - cnt;
        s_st[threadIdx.x] = w & 0xFFFFu;
        if (threadIdx.x == 0) {
            s_pos[FIN_THREADS] = total;
            for (uint32_t i = r; i < totals_words; i += NRANGE) totals_next[i] = 0; // (round 6: the hot buckets' sizes lie behind the ranges')
        }
        c = total;
        __syncthreads();
    }
    RS_STAMP(1, 3, 128);
    // where my range goes in the sorted batch: the sizes of the ranges before mine (and, for the last block, the largest one)
    auto place = [&]() -> uint32_t {
        uint32_t mine = 0, big = 0;
        big = tot_early;
        mine = threadIdx.x < r ? big : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            mine += __shfl_xor(mine, off, 64);
            big = max(big, __shfl_xor(big, off, 64));
        }
        __syncthreads(); // (s_part is free again)
        if (lane == 0) {
            s_part[wave] = mine;
            s_part[FIN_WAVES + wave] = big;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t b0 = 0, m0 = 0;
            for (int q = 0; q < FIN_WAVES; ++q) {
                b0 += s_part[q];
                m0 = max(m0, s_part[FIN_WAVES + q]);
            }
            s_base = b0;
            if (r == NRANGE - 1 && hot_n != nullptr) {
                hot_n[hot_ids] = b0 + c; // (rank form: the requests the ranges hold = where the evaluation's sorted part ends)
                // ... mirrored for the host, which sizes the sorted part's grid of LATER batches by it (never waited for)
                if (cold_hint != nullptr) __hip_atomic_store(cold_hint, ((unsigned long long)n << 32) | (b0 + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (r == NRANGE - 1 && range_hint != nullptr)
                __hip_atomic_store(range_hint, ((unsigned long long)n << 32) | m0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        return s_base;
    };
    if (c == 0u) {
        if (r == NRANGE - 1) (void)place(); // (the hint)
        return;
    }
    if (c <= FIN_CAP) {
        // which tile owns position p of the range
        auto expand = [&](uint32_t* s_owner) {
            // tile t's piece is usually a handful of elements (n / 256 / tiles: 16 of a 1 Mi batch): its thread writes them;
            // a long piece (the range's requests came in a burst) is left to the whole wave
            constexpr uint32_t LONG_PIECE = 48;
            const uint32_t t = threadIdx.x;
            uint32_t p0 = 0, p1 = 0;
            if (t < tiles) {
                p0 = s_pos[t];
                p1 = s_pos[t + 1];
            }
            const bool is_long = p1 - p0 > LONG_PIECE;
            if (!is_long)
                for (uint32_t p = p0; p < p1; ++p) s_owner[p] = t;
            unsigned long long lm = __ballot(is_long);
            while (lm) {
                const int src = __builtin_ctzll(lm);
                lm &= lm - 1ull;
                const uint32_t q0 = __shfl(p0, src, 64), q1 = __shfl(p1, src, 64), tt = (uint32_t)wave * 64u + (uint32_t)src;
                for (uint32_t p = q0 + (uint32_t)lane; p < q1; p += 64u) s_owner[p] = tt;
            }
        };
        if (width <= CNT_W_MAX) {
            // COUNTING: one byte counter per slot, bumped with one LDS atomic per element whose return value is the
            // element's arrival number among the requests of its slot; a scan over the counters gives every slot's place.
            // Requests of one slot must end up in index order (= position order p): the few slots with more than one
            // request sort their members by p (each counts the members before it).  A slot with more than CNT_DUP_MAX
            // requests (the counter would not even hold 256) sends the whole range to the ballot passes below.
            const uint32_t nw = (width + 3u) / 4u; // (the counters were cleared while the table word was on its way)
            if (threadIdx.x == 0) s_flag = 0;
            uint32_t* s_owner = c_fin; // (free until the sorted range is written)
            expand(s_owner);
            __syncthreads();
            RS_STAMP(1, 4, 128);
            const uint32_t strip = ((c + FIN_THREADS - 1) / FIN_THREADS) * 64u;
            uint32_t kv[FIN_ITEMS], arr[FIN_ITEMS];
            bool valid[FIN_ITEMS], over = false;
#pragma unroll
            for (int j = 0; j < FIN_ITEMS; ++j) {
                const uint32_t p = (uint32_t)wave * strip + (uint32_t)j * 64u + (uint32_t)lane;
                valid[j] = (uint32_t)j * 64u < strip && p < c;
                kv[j] = 0;
                arr[j] = 0;
                if (valid[j]) {
                    const uint32_t t = s_owner[p];
                    const uint64_t e = tiled[(size_t)t * tile_len + s_st[t] + (p - s_pos[t])];
                    s_idx[p] = (uint32_t)e;
                    const uint32_t sub = sub_of((uint32_t)(e >> 32));
                    kv[j] = (sub << FIN_POS_BITS) | p;
                    const uint32_t sh = 8u * (sub & 3u);
                    arr[j] = (atomicAdd(&c_cnt[sub >> 2], 1u << sh) >> sh) & 255u;
                    over |= arr[j] >= CNT_DUP_MAX;
                }
            }
            if (over) s_flag = 1u;
            __syncthreads();
            RS_STAMP(1, 5, 128);
            if (s_flag == 0u) { // (block-uniform)
                // where every word of four counters starts: each thread sums a run of words, the block scans the sums
                const uint32_t per = (nw + FIN_THREADS - 1) / FIN_THREADS;
                const uint32_t w0 = min(threadIdx.x * per, nw), w1 = min(w0 + per, nw);
                uint32_t mine = 0;
                for (uint32_t i = w0; i < w1; ++i) mine += __builtin_amdgcn_sad_u8(c_cnt[i], 0u, 0u); // (the sum of the word's four counters)
                uint32_t incl = mine;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                if (lane == 63) s_part[wave] = incl;
                __syncthreads();
                uint32_t run = incl - mine;
                for (int q = 0; q < wave; ++q) run += s_part[q];
                for (uint32_t i = w0; i < w1; ++i) {
                    c_ws[i] = (uint16_t)run;
                    run += __builtin_amdgcn_sad_u8(c_cnt[i], 0u, 0u);
                }
                __syncthreads();
                RS_STAMP(1, 6, 128);
                // place: slots with one request know their position; the members of the others meet in the slot's places first
                uint32_t q_of[FIN_ITEMS], len[FIN_ITEMS];
                uint32_t* s_grp = c_fin;
#pragma unroll
                for (int j = 0; j < FIN_ITEMS; ++j) {
                    q_of[j] = 0;
                    len[j] = 0;
                    if (valid[j]) {
                        const uint32_t sub = kv[j] >> FIN_POS_BITS, sh = 8u * (sub & 3u);
                        const uint32_t w = c_cnt[sub >> 2];
                        len[j] = (w >> sh) & 255u;
                        q_of[j] = (uint32_t)c_ws[sub >> 2] + __builtin_amdgcn_sad_u8(w & ((1u << sh) - 1u), 0u, 0u);
                        if (len[j] > 1u) s_grp[q_of[j] + arr[j]] = kv[j] & (FIN_CAP - 1u);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < FIN_ITEMS; ++j) {
                    if (valid[j] && len[j] > 1u) {
                        const uint32_t p = kv[j] & (FIN_CAP - 1u), g0 = q_of[j];
                        uint32_t before = 0;
                        for (uint32_t i = 0; i < len[j]; ++i) before += s_grp[g0 + i] < p ? 1u : 0u;
                        q_of[j] = g0 + before;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < FIN_ITEMS; ++j)
                    if (valid[j]) c_fin[q_of[j]] = kv[j];
                __syncthreads();
                RS_STAMP(1, 12, 128);
                const uint32_t base = place();
                RS_STAMP(1, 13, 128);
                uint64_t* out = elem_out + base;
                for (uint32_t q = threadIdx.x; q < c; q += FIN_THREADS) {
                    const uint32_t v = c_fin[q];
                    out[q] = ((uint64_t)slot_of(v >> FIN_POS_BITS) << 32) | s_idx[v & (FIN_CAP - 1u)];
                }
                RS_STAMP(1, 14, 128);
                return;
            }
            __syncthreads(); // (a slot too popular for its counter: everybody has left the counters before they are reused)
        }
        uint32_t* s_owner = s_kv[1]; // (free until the second pass writes it)
        expand(s_owner);
        __syncthreads();
        // every wave takes an equal strip of whole 64-position rows, so that all 16 waves work on a half-full range too
        const uint32_t strip = ((c + FIN_THREADS - 1) / FIN_THREADS) * 64u; // <= 64 * FIN_ITEMS
        uint32_t kv[FIN_ITEMS], dig[FIN_ITEMS], rank[FIN_ITEMS];
        bool valid[FIN_ITEMS];
#pragma unroll
        for (int j = 0; j < FIN_ITEMS; ++j) {
            const uint32_t p = (uint32_t)wave * strip + (uint32_t)j * 64u + (uint32_t)lane;
            valid[j] = (uint32_t)j * 64u < strip && p < c;
            kv[j] = 0;
            if (valid[j]) {
                const uint32_t t = s_owner[p];
                const uint64_t e = tiled[(size_t)t * tile_len + s_st[t] + (p - s_pos[t])];
                s_idx[p] = (uint32_t)e;
                kv[j] = (sub_of((uint32_t)(e >> 32)) << FIN_POS_BITS) | p;
            }
        }
        RS_STAMP(1, 5, 128);
        for (int ps = 0; ps < sub_passes; ++ps) {
            RS_STAMP(1, 6 + 3 * ps, 128);
            if (ps != 0) {
#pragma unroll
                for (int j = 0; j < FIN_ITEMS; ++j) {
                    const uint32_t p = (uint32_t)wave * strip + (uint32_t)j * 64u + (uint32_t)lane;
                    if (valid[j]) kv[j] = s_kv[(ps - 1) & 1][p];
                }
                __syncthreads(); // (everybody has read s_owner / the previous order before s_kv[ps & 1] is rewritten)
            }
#pragma unroll
            for (int j = 0; j < FIN_ITEMS; ++j) dig[j] = (kv[j] >> (FIN_POS_BITS + 8u * (uint32_t)ps)) & 255u;
            fin_rank<FIN_ITEMS>(dig, valid, rank, s_cnt, s_tot);
            RS_STAMP(1, 7 + 3 * ps, 128);
            fin_scan_digits(s_tot, s_start, s_part);
            RS_STAMP(1, 8 + 3 * ps, 128);
#pragma unroll
            for (int j = 0; j < FIN_ITEMS; ++j)
                if (valid[j]) s_kv[ps & 1][s_start[dig[j]] + s_cnt[wave][dig[j]] + rank[j]] = kv[j];
            __syncthreads();
        }
        RS_STAMP(1, 12, 128);
        const uint32_t base = place();
        RS_STAMP(1, 13, 128);
        uint64_t* out = elem_out + base;
        const uint32_t* fin = s_kv[(sub_passes - 1) & 1];
        for (uint32_t q = threadIdx.x; q < c; q += FIN_THREADS) {
            const uint32_t v = fin[q];
            out[q] = ((uint64_t)slot_of(v >> FIN_POS_BITS) << 32) | s_idx[v & (FIN_CAP - 1u)];
        }
        RS_STAMP(1, 14, 128);
        return;
    }

    // A range that does not fit: collect it, then stable LSD passes over the offset, chunk by chunk through global memory,
    // by this block alone (out -> scratch -> out; one pass: copied back).  Slow, exact, and rare: the host leaves skewed
    // streams on the LSD passes.
    const uint32_t base = place();
    uint64_t* b = elem_out + base;
    uint64_t* a = scratch + base;
    for (uint32_t q = threadIdx.x; q < c; q += FIN_THREADS) {
        uint32_t lo_t = 0, hi_t = tiles; // the last tile whose piece starts at or before q
        while (hi_t - lo_t > 1u) {
            const uint32_t mid = (lo_t + hi_t) >> 1;
            if (s_pos[mid] <= q) lo_t = mid;
            else hi_t = mid;
        }
        b[q] = tiled[(size_t)lo_t * tile_len + s_st[lo_t] + (q - s_pos[lo_t])];
    }
    __threadfence();
    __syncthreads();
    uint32_t* s_off = s_kv[0]; // [RADIX] running start of every digit in the destination
    for (int ps = 0; ps < sub_passes; ++ps) {
        const uint64_t* src = (ps & 1) ? a : b;
        uint64_t* dst = (ps & 1) ? b : a;
        const uint32_t shift = 8u * (uint32_t)ps;
        if (threadIdx.x < RADIX) s_off[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t q0 = 0; q0 < c; q0 += FIN_THREADS) { // histogram of the digit over the whole range
            const uint32_t q = q0 + threadIdx.x;
            const bool v = q < c;
            const uint32_t d = v ? ((sub_of((uint32_t)(ld_l2(&src[q]) >> 32)) >> shift) & 255u) : 0u;
            unsigned long long m = __ballot(v);
#pragma unroll
            for (int bt = 0; bt < 8; ++bt) {
                const unsigned long long bb = __ballot((d >> bt) & 1u);
                m &= ((d >> bt) & 1u) ? bb : ~bb;
            }
            if (v && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&s_off[d], (uint32_t)__popcll(m)); // one add per distinct digit of the wave
        }
        __syncthreads();
        if (threadIdx.x < RADIX) s_tot[threadIdx.x] = s_off[threadIdx.x];
        __syncthreads();
        fin_scan_digits(s_tot, s_off, s_part);
        for (uint32_t c0 = 0; c0 < c; c0 += FIN_CAP) { // chunks in order: the pass is stable
            uint64_t el[FIN_ITEMS];
            uint32_t dig[FIN_ITEMS], rank[FIN_ITEMS];
            bool valid[FIN_ITEMS];
#pragma unroll
            for (int j = 0; j < FIN_ITEMS; ++j) {
                const uint32_t p = c0 + (uint32_t)wave * (64u * FIN_ITEMS) + (uint32_t)j * 64u + (uint32_t)lane;
                valid[j] = p < c;
                el[j] = valid[j] ? ld_l2(&src[p]) : 0ull;
                dig[j] = (sub_of((uint32_t)(el[j] >> 32)) >> shift) & 255u;
            }
            fin_rank<FIN_ITEMS>(dig, valid, rank, s_cnt, s_tot);
#pragma unroll
            for (int j = 0; j < FIN_ITEMS; ++j)
                if (valid[j]) dst[s_off[dig[j]] + s_cnt[wave][dig[j]] + rank[j]] = el[j];
            __syncthreads();
            if (threadIdx.x < RADIX) s_off[threadIdx.x] += s_tot[threadIdx.x];
            __syncthreads();
        }
        __threadfence(); // this block's stores have reached the L2 before its next pass reads them back (ld_l2)
        __syncthreads();
    }
    if ((sub_passes & 1) != 0)
        for (uint32_t q = threadIdx.x; q < c; q += FIN_THREADS) b[q] = ld_l2(&a[q]);
}

} // namespace rs
