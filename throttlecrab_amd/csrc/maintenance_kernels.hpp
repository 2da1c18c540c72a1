// maintenance_kernels.hpp -- everything that is not a decision: expiry sweep (== AdaptiveStore::cleanup),
// top-denied-key selection, rate-plan id fills, the single-key `trait Store` operations.
// Kernels have internal linkage: every translation unit (engine.hpp) compiles the ones it launches.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tcgpu.h"
#include "eval_kernels.hpp"
#include "gcra_math.hpp"
#include "key_table.hpp"

namespace mk {

using ev::BLOCK;
using tc::Cell;

// ---------------------------------------------------------------------------
// K4: expiry sweep == AdaptiveStore::cleanup (adaptive_cleanup.rs:173-203)
// ---------------------------------------------------------------------------
// (later in round 4: four cells per thread in flight, and a block leaves its two counts in part[] for k_sweep_fold instead
// of adding them to the counters itself -- 2 048 blocks that finish together on three words, ~12.8 ns each on one word)
constexpr uint32_t SWEEP_GRID = 2048; // blocks of a sweep kernel at most
constexpr int SWEEP_UNROLL = 4;
__device__ __forceinline__ void sweep_block_counts(uint32_t removed, uint32_t live, uint32_t* __restrict__ part) {
    __shared__ uint32_t s_r[BLOCK / 64], s_l[BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) {
        removed += __shfl_down(removed, off, 64);
        live += __shfl_down(live, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_r[threadIdx.x >> 6] = removed;
        s_l[threadIdx.x >> 6] = live;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0, l = 0;
        for (int w = 0; w < BLOCK / 64; ++w) {
            r += s_r[w];
            l += s_l[w];
        }
        part[SWEEP_GRID + blockIdx.x] = r;
        part[2 * SWEEP_GRID + blockIdx.x] = l;
    }
}
static __global__ __launch_bounds__(BLOCK) void k_sweep(Cell* __restrict__ cells, uint64_t capacity, int64_t now, uint32_t* __restrict__ part) {
    uint32_t removed = 0, live = 0;
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < capacity; i0 += stride * SWEEP_UNROLL) {
        uint64_t ex[SWEEP_UNROLL];
#pragma unroll
        for (int j = 0; j < SWEEP_UNROLL; ++j) ex[j] = i0 + j * stride < capacity ? cells[i0 + j * stride].expiry : 0ull;
#pragma unroll
        for (int j = 0; j < SWEEP_UNROLL; ++j) {
            if (ex[j] == 0) continue;
            if (!(ex[j] > (uint64_t)now)) { // retain(|exp| *exp > now)
                Cell c;
                c.tat = 0;
                c.expiry = 0;
                cells[i0 + j * stride] = c;
                removed++;
            } else {
                live++;
            }
        }
    }
    sweep_block_counts(removed, live, part);
}

// TC_CFG_FIXED_PARAMS layout: expiry == tat + dvt of the key's plan (tc::fixed_cell); vacant == TAT_VACANT
static __global__ __launch_bounds__(BLOCK) void k_sweep_fixed(int64_t* __restrict__ tat8, const uint16_t* __restrict__ rate_id,
                                                       const tc::RateClass* __restrict__ classes, uint32_t uniform_class, uint64_t capacity,
                                                       int64_t now, uint32_t* __restrict__ part) {
    uint32_t removed = 0, live = 0;
    const int64_t dvt_all = classes[uniform_class].dvt;
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i0 = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i0 < capacity; i0 += stride * SWEEP_UNROLL) {
        int64_t tt[SWEEP_UNROLL];
#pragma unroll
        for (int j = 0; j < SWEEP_UNROLL; ++j) tt[j] = i0 + j * stride < capacity ? tat8[i0 + j * stride] : tc::TAT_VACANT;
#pragma unroll
        for (int j = 0; j < SWEEP_UNROLL; ++j) {
            if (tt[j] == tc::TAT_VACANT) continue;
            const int64_t dvt = uniform_class ? dvt_all : classes[rate_id[i0 + j * stride]].dvt;
            if (!(tc::fixed_cell(tt[j], dvt).expiry > (uint64_t)now)) { // retain(|exp| *exp > now)
                tat8[i0 + j * stride] = tc::TAT_VACANT;
                removed++;
            } else {
                live++;
            }
        }
    }
    sweep_block_counts(removed, live, part);
}
// the blocks' counts -> the counters (one block behind a slot-mode sweep; key mode: k_sweep_decide)
static __global__ __launch_bounds__(1024) void k_sweep_fold(const uint32_t* __restrict__ part, uint32_t blocks, unsigned long long* __restrict__ removed_out,
                                                            unsigned long long* counters) {
    __shared__ uint32_t s_w[16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t removed = 0, live = 0;
    for (uint32_t b = threadIdx.x; b < blocks; b += 1024) {
        removed += part[SWEEP_GRID + b];
        live += part[2 * SWEEP_GRID + b];
    }
    for (int off = 32; off > 0; off >>= 1) {
        removed += __shfl_down(removed, off, 64);
        live += __shfl_down(live, off, 64);
    }
    if (lane == 0) {
        s_w[wave][0] = removed;
        s_w[wave][1] = live;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0, l = 0;
        for (int w = 0; w < 16; ++w) {
            r += s_w[w][0];
            l += s_w[w][1];
        }
        *removed_out = r;
        counters[TC_CNT_SWEPT] += r;
        counters[TC_CNT_LIVE_SLOTS] = l;
    }
}

// Key-mode sweep: same retain rule, and an expired (or never written) bound slot also loses its key: slot back on the free
// stack, tombstone in the hash table.  Three kernels without a single contended atomic (later in round 4: every block of the
// old k_sweep_keys reserved its stretch of the free stack with a returning atomicAdd on the stack pointer and added to three
// counters when it was done -- 2 048 blocks that finish together, ~12.8 ns per block on one word: a quarter of the kernel,
// measured by its grid size, gpurun_out/r04_v51/sweepgrid.txt):
//   k_sweep_keys        every block owns a contiguous range of slots; the slots it unbinds go into ITS stretch of a list that
//                       is as long as the table (work.list[first ...]), its counts into work.part[]
//   k_sweep_decide      one block: scans the blocks' counts (-> where each stretch goes on the free stack), moves the stack
//                       pointer, sums the counters, latches "the table will be rebuilt" (the tombstones it has + the keys just
//                       unbound would fill more than 1/4 of it) and the overflow arena's compaction
//   k_sweep_tombstones  block b copies stretch b onto the stack and -- UNLESS the rebuild is due, which clears the whole table
//                       and re-enters the bound keys anyway -- turns the unbound keys' entries into tombstones: four keys per
//                       thread in flight, the entry's position from the compact column pos_col[] (not from the key's 128-byte
//                       record), a 4-byte store into the binding word's low half (nobody reads a tombstone's tag).
//                       (The first sweep of configs[4] unbinds 9 M of 10.5 M keys: 9 M entries written only to be cleared by the
//                       rebuild that followed were a third of that sweep's 1 GB.)
struct SweepWork {
    uint32_t* list; // [capacity]
    uint32_t* part; // [3][SWEEP_GRID]: unbound, removed, live per block
    uint32_t* off;  // [SWEEP_GRID]: unbound slots of the blocks before
};
__host__ __device__ __forceinline__ uint64_t sweep_per_block(uint64_t capacity, uint32_t blocks) {
    return ((capacity + blocks - 1) / blocks + BLOCK - 1) / BLOCK * BLOCK;
}

// `touched` (round 6, the sweep BESIDE the newest batch's evaluation -- keys.hip: sweep_keys_device): a byte per slot, non-zero for
// the slots that batch asked for.  Its evaluation may still be writing their cells, so they are left alone here (counted live,
// which they are unless their request was an error or denied on a key that had expired) and looked at by k_sweep_fixup once that
// evaluation is done.  nullptr: the sweep runs behind every evaluation, as before.
static __global__ __launch_bounds__(BLOCK) void k_sweep_keys(Cell* __restrict__ cells, kt::Table t, int64_t now, SweepWork work,
                                                      uint32_t* __restrict__ denied, const uint8_t* __restrict__ touched) {
    uint32_t removed = 0, live = 0, fill = 0; // fill is uniform over the block
    const uint64_t per_block = sweep_per_block(t.capacity, gridDim.x);
    const uint64_t first = (uint64_t)blockIdx.x * per_block;
    const uint64_t last = first + per_block < t.capacity ? first + per_block : t.capacity;
    uint32_t* __restrict__ mine_list = work.list + first;
    // eight slots per thread and round: their `bound` bytes and their cells' expiry words are all requested before anything is
    // looked at, and the block ranks the slots it unbinds once per round
    constexpr int SWEEP_ITEMS = 8;
    __shared__ uint32_t s_cnt[BLOCK / 64];
    for (uint64_t base = first; base < last; base += (uint64_t)BLOCK * SWEEP_ITEMS) {
        uint8_t bnd[SWEEP_ITEMS];
        uint64_t exp_[SWEEP_ITEMS];
        bool unbind[SWEEP_ITEMS];
#pragma unroll
        for (int j = 0; j < SWEEP_ITEMS; ++j) {
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            bnd[j] = i < last ? t.bound[i] : (uint8_t)0;
        }
        uint8_t tch[SWEEP_ITEMS];
#pragma unroll
        for (int j = 0; j < SWEEP_ITEMS; ++j) {
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            tch[j] = (touched != nullptr && i < last) ? touched[i] : (uint8_t)0; // (wave-uniform branch)
        }
        // (only the bound slots' cells: between sweeps a quarter of configs[4]'s slots are bound, and free slots come in stretches --
        // the others all ask for the block's first cell, one line)
#pragma unroll
        for (int j = 0; j < SWEEP_ITEMS; ++j) {
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            exp_[j] = cells[(bnd[j] && !tch[j]) ? i : first].expiry;
        }
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < SWEEP_ITEMS; ++j) {
            unbind[j] = false;
            if (bnd[j] && tch[j]) {
                live++; // (k_sweep_fixup takes it back if the evaluation left the cell expired)
            } else if (bnd[j]) {
                if (!(exp_[j] > (uint64_t)now)) {
                    if (exp_[j] != 0) removed++; // the reference's map only ever held written entries
                    unbind[j] = true;
                    mine++;
                } else {
                    live++;
                }
            }
        }
        // rank of my unbound slots among the block's (any order will do: the free stack is a pool)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_cnt[wave] = incl;
        __syncthreads();
        uint32_t before = incl - mine, total = 0;
        for (int w = 0; w < BLOCK / 64; ++w) {
            if (w < wave) before += s_cnt[w];
            total += s_cnt[w];
        }
#pragma unroll
        for (int j = 0; j < SWEEP_ITEMS; ++j) {
            if (!unbind[j]) continue;
            const uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
            Cell z;
            z.tat = 0;
            z.expiry = 0;
            cells[i] = z;
            t.bound[i] = 0; // (the key's table entry is dealt with by k_sweep_tombstones -- or by the rebuild, all at once)
            if (denied) {
                // the slot will serve another key; its denials stay with the key it served (kt::RetiredRec)
                const uint32_t dc = denied[i];
                if (dc) {
                    const uint32_t klen = t.rec[i].len;
                    if (klen != kt::NO_SLOT) kt::retire_denials(t, t.rec[i].hash, kt::stored_key(t, (uint32_t)i, klen), klen, dc);
                    denied[i] = 0;
                }
            }
            mine_list[fill + before] = (uint32_t)i; // (fill + before < the slots looked at so far <= per_block)
            ++before;
        }
        fill += total;
        __syncthreads(); // s_cnt is reused next round
    }
    sweep_block_counts(removed, live, work.part);
    if (threadIdx.x == 0) work.part[blockIdx.x] = fill;
}

// A batch's slots, marked for the sweep that runs beside its evaluation (value != 0: mark, 0: not used -- the marks are
// taken off by k_sweep_fixup).  Unresolved keys (slot >= capacity) have no slot to mark.
static __global__ __launch_bounds__(BLOCK) void k_touch_mark(const uint32_t* __restrict__ slots, uint32_t n, uint32_t capacity, uint8_t* __restrict__ touched) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t sl = slots[i];
    if (sl < capacity) touched[sl] = (uint8_t)1;
}

// ... and, once the newest evaluation is done, the marked slots: the mark column is read sixteen slots per lane (most words are zero),
// every marked slot's cell is requested, the marks are taken off.  A cell still expired at `now` -- the request was an error, or was
// denied on a key that had expired, or the caller sweeps at a later time than the batch's -- and the key loses its slot exactly as in
// k_sweep_keys / k_sweep_tombstones: cell cleared, unbound, its denials retired, the entry a tombstone, the slot on the free stack (an
// atomic on the stack pointer: these are a handful unless the sweep's time is far ahead), the counters corrected (k_sweep_decide has
// run).  A marked slot need not be bound any more: of two marked batches the older one's slots may have gone in a sweep between the
// two (and may or may not have been handed out again since) -- the `bound` column is read beside the marks.
static __global__ __launch_bounds__(BLOCK) void k_sweep_fixup(Cell* __restrict__ cells, kt::Table t, int64_t now, uint8_t* __restrict__ touched,
                                                       uint32_t* __restrict__ denied, unsigned long long* __restrict__ removed_out,
                                                       unsigned long long* __restrict__ counters) {
    uint32_t removed = 0, unbound = 0;
    const uint64_t vecs = ((uint64_t)t.capacity + 15u) / 16u; // (the column is allocated and zero up to a multiple of 16)
    uint4* __restrict__ marks = reinterpret_cast<uint4*>(touched);
    for (uint64_t v = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; v < vecs; v += (uint64_t)gridDim.x * BLOCK) {
        const uint4 m = marks[v];
        if ((m.x | m.y | m.z | m.w) == 0u) continue;
        const uint4 bd = *reinterpret_cast<const uint4*>(t.bound + v * 16u); // (hipMalloc aligns `bound`; its last vector may reach past the
                                                                              // capacity into the block's padding: no marks there)
        const uint32_t w[4] = {m.x, m.y, m.z, m.w}, bw[4] = {bd.x, bd.y, bd.z, bd.w};
        uint64_t ex[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool on = ((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) != 0u && ((bw[j >> 2] >> (8 * (j & 3))) & 0xFFu) != 0u;
            ex[j] = on ? cells[v * 16u + (uint64_t)j].expiry : ~0ull;
        }
        marks[v] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (ex[j] > (uint64_t)now) continue; // (not marked: ~0)
            const uint32_t sl = (uint32_t)(v * 16u) + (uint32_t)j;
            if (ex[j] != 0) removed++; // the reference's map only ever held written entries
            unbound++;
            Cell z;
            z.tat = 0;
            z.expiry = 0;
            cells[sl] = z;
            t.bound[sl] = 0;
            if (denied) {
                const uint32_t dc = denied[sl];
                if (dc) {
                    const uint32_t klen = t.rec[sl].len;
                    if (klen != kt::NO_SLOT) kt::retire_denials(t, t.rec[sl].hash, kt::stored_key(t, sl, klen), klen, dc);
                    denied[sl] = 0;
                }
            }
            const uint32_t pos = t.pos_col[sl];
            *reinterpret_cast<uint32_t*>(&t.ktab[pos].w) = kt::VAL_TOMB; // (little-endian: the `val` half)
            atomicAdd(&t.tombs[pos % kt::TOMB_SHARDS], 1u);
            const int at = atomicAdd(t.free_top, 1);
            t.free_slots[at] = sl;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        removed += __shfl_down(removed, off, 64);
        unbound += __shfl_down(unbound, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && unbound != 0u) {
        if (removed) {
            atomicAdd(removed_out, (unsigned long long)removed);
            atomicAdd(&counters[TC_CNT_SWEPT], (unsigned long long)removed);
        }
        atomicAdd(&counters[TC_CNT_LIVE_SLOTS], 0ull - (unsigned long long)unbound);
    }
}

constexpr int DECIDE_THREADS = 1024;
static __global__ __launch_bounds__(DECIDE_THREADS) void k_sweep_decide(kt::Table t, SweepWork work, uint32_t blocks, int* __restrict__ top_save,
                                                                        uint32_t* __restrict__ flag, unsigned long long* __restrict__ oflag,
                                                                        unsigned long long* __restrict__ removed_out, unsigned long long* counters) {
    static_assert(SWEEP_GRID == 2 * DECIDE_THREADS, "two blocks of k_sweep_keys per thread");
    __shared__ uint32_t s_w[DECIDE_THREADS / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b0 = 2u * threadIdx.x, b1 = b0 + 1u;
    const uint32_t c0 = b0 < blocks ? work.part[b0] : 0u, c1 = b1 < blocks ? work.part[b1] : 0u;
    uint32_t removed = (b0 < blocks ? work.part[SWEEP_GRID + b0] : 0u) + (b1 < blocks ? work.part[SWEEP_GRID + b1] : 0u);
    uint32_t live = (b0 < blocks ? work.part[2 * SWEEP_GRID + b0] : 0u) + (b1 < blocks ? work.part[2 * SWEEP_GRID + b1] : 0u);
    uint32_t incl = c0 + c1;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    for (int off = 32; off > 0; off >>= 1) {
        removed += __shfl_down(removed, off, 64);
        live += __shfl_down(live, off, 64);
    }
    if (lane == 63) s_w[wave][0] = incl;
    if (lane == 0) {
        s_w[wave][1] = removed;
        s_w[wave][2] = live;
    }
    __syncthreads();
    uint32_t before = incl - (c0 + c1), total = 0, r = 0, l = 0;
    for (int w = 0; w < DECIDE_THREADS / 64; ++w) {
        if (w < wave) before += s_w[w][0];
        total += s_w[w][0];
        r += s_w[w][1];
        l += s_w[w][2];
    }
    if (b0 < blocks) work.off[b0] = before;
    if (b1 < blocks) work.off[b1] = before + c0;
    if (threadIdx.x == 0) {
        const int top = *t.free_top;
        *top_save = top; // (k_sweep_tombstones pushes above it)
        *t.free_top = top + (int)total;
        uint32_t tombs = 0; // (the shards wrap around individually; their sum is the count)
        for (uint32_t s = 0; s < kt::TOMB_SHARDS; ++s) tombs += t.tombs[s];
        *flag = (uint64_t)tombs + (uint64_t)total > (t.nb_mask + 1) / 4 ? 1u : 0u;
        kt::overflow_decide(t, oflag);
        *removed_out = r;
        counters[TC_CNT_SWEPT] += r;
        counters[TC_CNT_LIVE_SLOTS] = l;
    }
}

constexpr int TOMB_ITEMS = 4;
static __global__ __launch_bounds__(BLOCK) void k_sweep_tombstones(kt::Table t, SweepWork work, const int* __restrict__ top_save,
                                                                   const uint32_t* __restrict__ flag, uint32_t spread) {
    const uint32_t cnt = work.part[blockIdx.x];
    if (cnt == 0u) return;
    const bool tomb = *flag == 0u; // (else: the rebuild that follows drops every entry of an unbound key by itself)
    const uint32_t* __restrict__ src = work.list + (uint64_t)blockIdx.x * sweep_per_block(t.capacity, gridDim.x);
    // the sweep's m freed slots, in slot order, are the stack entries [top_save, top_save + m); spread (kt::spread_position):
    // the g-th of them is the one popped spread_position(g)-th, i.e. entry top_save + m - 1 - that
    const uint32_t first = work.off[blockIdx.x];
    const uint32_t m = (uint32_t)(*t.free_top - *top_save), cols = kt::spread_cols(m);
    uint32_t* __restrict__ base = t.free_slots + *top_save;
    for (uint32_t k0 = threadIdx.x; k0 < cnt; k0 += BLOCK * TOMB_ITEMS) {
        uint32_t slot[TOMB_ITEMS], pos[TOMB_ITEMS];
#pragma unroll
        for (int j = 0; j < TOMB_ITEMS; ++j) slot[j] = k0 + j * BLOCK < cnt ? src[k0 + j * BLOCK] : kt::NO_SLOT;
#pragma unroll
        for (int j = 0; j < TOMB_ITEMS; ++j) pos[j] = tomb && slot[j] != kt::NO_SLOT ? t.pos_col[slot[j]] : 0u;
#pragma unroll
        for (int j = 0; j < TOMB_ITEMS; ++j)
            if (slot[j] != kt::NO_SLOT) {
                const uint32_t g = first + k0 + j * BLOCK;
                base[spread ? m - 1u - kt::spread_position(g, m, cols) : g] = slot[j];
                if (tomb) *reinterpret_cast<uint32_t*>(&t.ktab[pos[j]].w) = kt::VAL_TOMB; // (little-endian: the `val` half)
            }
    }
    if (tomb && threadIdx.x == 0) atomicAdd(&t.tombs[blockIdx.x % kt::TOMB_SHARDS], cnt);
}

// tc_debug_check_keys: the table seen from the slots (pass 0) and from the entries (pass 1); bad[0] = inconsistencies,
// bad[1] = bound slots
static __global__ __launch_bounds__(BLOCK) void k_check_keys(kt::Table t, int pass, unsigned long long* __restrict__ bad) {
    uint32_t wrong = 0, bound = 0;
    if (pass == 0) {
        for (uint64_t s = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; s < t.capacity; s += (uint64_t)gridDim.x * BLOCK) {
            if (!t.bound[s]) continue;
            ++bound;
            const uint32_t pos = t.pos_col[s];
            const kt::KeyRec& kr = t.rec[s];
            if (kr.len == kt::NO_SLOT || pos > t.nb_mask) {
                ++wrong;
                continue;
            }
            const kt::Entry& en = t.ktab[pos];
            if ((uint32_t)en.w != (uint32_t)s + 2u || en.hash != kr.hash || (en.w & 0xFFFFFFFF00000000ull) != kt::entry_meta(kr.hash, kr.len)) ++wrong;
        }
    } else {
        for (uint64_t p = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; p <= t.nb_mask; p += (uint64_t)gridDim.x * BLOCK) {
            const uint32_t val = (uint32_t)t.ktab[p].w;
            if (val == kt::VAL_EMPTY || val == kt::VAL_TOMB) continue;
            if (val & kt::VAL_PENDING) {
                ++wrong; // a claim nobody bound
                continue;
            }
            const uint32_t s = val - 2u;
            if (s >= t.capacity || !t.bound[s] || t.pos_col[s] != (uint32_t)p) ++wrong;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        wrong += __shfl_down(wrong, off, 64);
        bound += __shfl_down(bound, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (wrong) atomicAdd(&bad[0], (unsigned long long)wrong);
        if (bound) atomicAdd(&bad[1], (unsigned long long)bound);
    }
}

// ---------------------------------------------------------------------------
// top denied keys (metrics.rs:24-76 keeps a capped HashMap on the host; here the counts are
// exact, one u32 per slot, and the top K are selected on demand: radix select of the K-th
// largest count, 8 bits per pass, then one compaction pass)
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(BLOCK) void k_denied_hist(const uint32_t* __restrict__ denied, uint64_t capacity,
                                                       uint32_t prefix, uint32_t prefix_mask, uint32_t shift,
                                                       uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0; // BLOCK == 256
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t c = denied[i];
        if (c != 0u && (c & prefix_mask) == prefix) atomicAdd(&s_h[(c >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// entries with count > T go to list 0, entries with count == T to list 1 (each capped at `cap_out`)
static __global__ __launch_bounds__(BLOCK) void k_denied_collect(const uint32_t* __restrict__ denied, uint64_t capacity, uint32_t T,
                                                          uint32_t* __restrict__ n_out /*[2]*/, uint32_t* __restrict__ lists,
                                                          uint32_t cap_out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t c = denied[i];
        if (c == 0u || c < T) continue;
        const int which = c > T ? 0 : 1;
        const uint32_t at = atomicAdd(&n_out[which], 1u);
        if (at < cap_out) {
            uint32_t* l = lists + (size_t)which * 2 * cap_out;
            l[2 * at] = (uint32_t)i;
            l[2 * at + 1] = c;
        }
    }
}

// key mode: copy the KeyRec of each listed slot into a dense array
static __global__ __launch_bounds__(BLOCK) void k_gather_keyrecs(kt::Table t, const uint32_t* __restrict__ slots, uint32_t n,
                                                          kt::KeyRec* __restrict__ out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slots[i];
    // (eight 16-byte words, no record on the stack: this was the library's last kernel with scratch -- see k_eval_lean_hot)
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    static_assert(sizeof(kt::KeyRec) == 128, "eight 16-byte words");
    const bool have = s < t.capacity && t.bound[s];
    const v2u64* __restrict__ src = reinterpret_cast<const v2u64*>(&t.rec[have ? s : 0u]);
    v2u64* __restrict__ dst = reinterpret_cast<v2u64*>(&out[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v2u64 v = src[j];
        if (!have) v = j == 0 ? v2u64{0ull, (unsigned long long)kt::NO_SLOT} : v2u64{0ull, 0ull}; // hash 0, len NO_SLOT, pos 0
        dst[j] = v;
    }
}

static __global__ __launch_bounds__(BLOCK) void k_fill_i64(int64_t* __restrict__ a, uint64_t n, int64_t v) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) a[i] = v;
}

static __global__ __launch_bounds__(BLOCK) void k_fill_rate_id(uint16_t* __restrict__ rate_id, uint64_t capacity, uint16_t id) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK)
        rate_id[i] = id;
}

static __global__ __launch_bounds__(BLOCK) void k_scatter_rate_id(uint16_t* __restrict__ rate_id, const uint32_t* __restrict__ slots,
                                                           const uint16_t* __restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) rate_id[slots ? slots[i] : i] = src[i];
}

// `trait Store` shims on one resolved slot (store/mod.rs:85-133,
// adaptive_cleanup.rs:221-279).  op: 0 get, 1 cas, 2 set_nx
struct StoreOpResult {
    int64_t value;
    int32_t flag;
    int32_t pad;
};
static __global__ void k_store_op(Cell* cells, uint64_t slot, int op, int64_t a, int64_t b, uint64_t ttl, int64_t now,
                           StoreOpResult* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Cell c = cells[slot];
    const bool live = c.expiry > (uint64_t)now;
    StoreOpResult r;
    r.value = 0;
    r.flag = 0;
    r.pad = 0;
    uint64_t e = (uint64_t)now + ttl;
    if (e < ttl) e = UINT64_MAX;
    if (op == 0) {
        if (live) {
            r.value = c.tat;
            r.flag = 1;
        }
    } else if (op == 1) {
        if (live && c.tat == a) {
            c.tat = b;
            c.expiry = e;
            cells[slot] = c;
            r.flag = 1;
        }
    } else {
        if (!live) {
            c.tat = a;
            c.expiry = e;
            cells[slot] = c;
            r.flag = 1;
        }
    }
    *out = r;
}

// ---------------------------------------------------------------------------
// Do two streams run concurrently?  HIP multiplexes streams onto a few hardware queues (round robin
// at creation); two streams on one queue execute in order.  k_probe_spin (on stream A) waits for a
// flag that only k_probe_set (on stream B) raises, or gives up after `timeout` ticks of the 100 MHz
// wall clock: if it saw the flag, B ran while A was running.
// ---------------------------------------------------------------------------
static __global__ void k_probe_spin(uint32_t* flag, uint32_t* saw, long long timeout) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long t0 = wall_clock64();
    uint32_t v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < timeout)
        __builtin_amdgcn_s_sleep(8);
    *saw = v;
}
// Two streams on different hardware queues can still share a dispatch PIPE: a kernel on the one then is not handed out while
// a kernel on the other still has blocks to hand out (tools/pipeprobe.hip: with 8 queues, the stream created four after
// another one; intermittently).  k_probe_occupy (a grid four times what the chip holds, every block staying `ticks`) stamps
// its start; a one-block k_probe_stamp on the other stream stamps when it ran: one round of blocks later if the two do not
// collide, four rounds later if they do.
static __global__ void k_probe_occupy(long long ticks, long long* start) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) *start = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static __global__ void k_probe_stamp(long long* out) {
    if (threadIdx.x == 0) *out = wall_clock64();
}
static __global__ void k_probe_set(uint32_t* flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// A slot column that arrives in pieces (tc_batch.seg_slot: one piece per source GPU of a sharded deployment) gathered
// into one column: block b copies entries [b * PER_BLOCK, ...) of the concatenation, finding its piece by a walk over
// the (at most 64) prefix sums.
// ---------------------------------------------------------------------------
struct Segments {
    const uint32_t* ptr[64];
    uint32_t start[65]; // prefix sums; start[n] = total
    uint32_t n;
};
static __global__ __launch_bounds__(BLOCK) void k_concat(Segments sg, uint32_t* __restrict__ out) {
    const uint32_t total = sg.start[sg.n];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < total; i += gridDim.x * BLOCK) {
        uint32_t s = 0;
        while (s + 1 < sg.n && i >= sg.start[s + 1]) ++s;
        out[i] = sg.ptr[s][i - sg.start[s]];
    }
}

// the other direction: one column (the router's segments, one after the other) scattered to one destination per segment
struct Destinations {
    uint32_t* ptr[64];
    uint32_t start[65];
    uint32_t n;
};
static __global__ __launch_bounds__(BLOCK) void k_forward(const uint32_t* __restrict__ src, Destinations ds) {
    const uint32_t total = ds.start[ds.n];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < total; i += gridDim.x * BLOCK) {
        uint32_t s = 0;
        while (s + 1 < ds.n && i >= ds.start[s + 1]) ++s;
        ds.ptr[s][i - ds.start[s]] = src[i];
    }
}

// ---------------------------------------------------------------------------
// Plain copy, used for device memory -> PINNED host memory (which the GPU addresses directly): the result
// transfers of TC_B_ASYNC batches are ordinary kernels on the engine's stream (see copy_back_async).
// 16 bytes per lane per step when both ends and the size allow, else bytes.
// ---------------------------------------------------------------------------
static __global__ __launch_bounds__(BLOCK) void k_copy(void* __restrict__ dst, const void* __restrict__ src, size_t bytes) {
    const size_t tid = (size_t)blockIdx.x * BLOCK + threadIdx.x, nthreads = (size_t)gridDim.x * BLOCK;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const size_t n16 = bytes / 16;
        const uint4* s = static_cast<const uint4*>(src);
        uint4* d = static_cast<uint4*>(dst);
        size_t i = tid;
        for (; i + 3 * nthreads < n16; i += 4 * nthreads) { // four loads in flight per lane: PCIe latency, few waves
            const uint4 a = s[i], b = s[i + nthreads], c = s[i + 2 * nthreads], e = s[i + 3 * nthreads];
            d[i] = a;
            d[i + nthreads] = b;
            d[i + 2 * nthreads] = c;
            d[i + 3 * nthreads] = e;
        }
        for (; i < n16; i += nthreads) d[i] = s[i];
        for (i = n16 * 16 + tid; i < bytes; i += nthreads) static_cast<uint8_t*>(dst)[i] = static_cast<const uint8_t*>(src)[i];
    } else {
        for (size_t i = tid; i < bytes; i += nthreads) static_cast<uint8_t*>(dst)[i] = static_cast<const uint8_t*>(src)[i];
    }
}

// Round 6: the evaluation's notes on heavy runs (ev::heavy_note) -> pinned host memory, the sequence word of the copy behind them;
// the table is cleared for the evaluations to come.  Enqueued now and then (slots.hip), never waited for.  1 024 words per block;
// the last block to finish writes the sequence word.  (At first ONE block on the engine's stream, between two evaluations: 11-16 us
// there, and a kernel boundary more on the stream every step waits for -- three times in the driver's twenty batches of a skewed
// stream.  Now on the grouping stream of the next pipelined batch, beside the evaluations: a note that lands while the copy
// runs is in this copy or the next one.)
static __global__ __launch_bounds__(1024) void k_heavy_publish(unsigned long long* __restrict__ dst, unsigned long long* __restrict__ src,
                                                               uint32_t words, unsigned long long seq, uint32_t* __restrict__ done) {
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    if (i < words) {
        dst[i] = __hip_atomic_exchange(&src[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&dst[words], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Round 6, TC_B_PLAN_DICT: the batch's dictionary-coded plan column (and its u32 quantities) -> the wide staging columns every
// evaluation kernel reads.  What the compact form saves is PCIe bytes (6 per request instead of 32); on the device the
// expansion is 32 B of HBM writes per request, ~10 us per 1 Mi.  An index beyond the dictionary decodes to (0, 0, 0): the
// request's status becomes TC_INVALID_RATE_LIMIT (rate_limiter.rs:114-117), like any other non-positive triple.
static __global__ __launch_bounds__(BLOCK) void k_expand_plans(const int64_t* __restrict__ dict, uint32_t n_plans, const uint16_t* __restrict__ id,
                                                              const uint32_t* __restrict__ q32, uint32_t n, int64_t* __restrict__ burst,
                                                              int64_t* __restrict__ count, int64_t* __restrict__ period, int64_t* __restrict__ q) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t k = id[i];
        const bool ok = k < n_plans;
        burst[i] = ok ? dict[3 * (size_t)k] : 0;
        count[i] = ok ? dict[3 * (size_t)k + 1] : 0;
        period[i] = ok ? dict[3 * (size_t)k + 2] : 0;
        if (q32 != nullptr) q[i] = (int64_t)q32[i];
    }
}

// Several such copies in ONE launch (blockIdx.y = the segment): the input columns of a synchronous host batch from PINNED memory.
// Seven hipMemcpyAsync calls on one stream cost ~10-18 us each before the first byte moves (a 4 Ki-request reference-shaped
// call spent 70 of its 200 us there); one launch reads them all over PCIe side by side.
struct CopySegs {
    const void* src[8];
    void* dst[8];
    size_t bytes[8];
};
static __global__ __launch_bounds__(BLOCK) void k_copy_multi(CopySegs sg) {
    const void* __restrict__ src = sg.src[blockIdx.y];
    void* __restrict__ dst = sg.dst[blockIdx.y];
    const size_t bytes = sg.bytes[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * BLOCK + threadIdx.x, nthreads = (size_t)gridDim.x * BLOCK;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0) {
        const size_t n16 = bytes / 16;
        const uint4* s = static_cast<const uint4*>(src);
        uint4* d = static_cast<uint4*>(dst);
        size_t i = tid;
        for (; i + 3 * nthreads < n16; i += 4 * nthreads) {
            const uint4 a = s[i], b = s[i + nthreads], c = s[i + 2 * nthreads], e = s[i + 3 * nthreads];
            d[i] = a;
            d[i + nthreads] = b;
            d[i + 2 * nthreads] = c;
            d[i + 3 * nthreads] = e;
        }
        for (; i < n16; i += nthreads) d[i] = s[i];
        for (i = n16 * 16 + tid; i < bytes; i += nthreads) static_cast<uint8_t*>(dst)[i] = static_cast<const uint8_t*>(src)[i];
    } else {
        for (size_t i = tid; i < bytes; i += nthreads) static_cast<uint8_t*>(dst)[i] = static_cast<const uint8_t*>(src)[i];
    }
}

// ---------------------------------------------------------------------------
// The policy feed (autosweep.hip): what the host-side cleanup policy (AdaptiveStore::should_clean, adaptive_cleanup.rs:138-171)
// needs to know about the device -- how many requests were allowed so far (the reference counts one operation per mutating
// store call), how many entries the store holds, how many a sweep removed, the newest timestamp of a device column -- written
// into PINNED host memory behind every mutating call while a policy is set.  The host reads it without ever waiting: the
// record is bracketed by two copies of its sequence number (begin first, end last, system-scope fences between), a reader that
// finds them different reads again later.
// ---------------------------------------------------------------------------
static __global__ void k_counter_add(unsigned long long* word, unsigned long long delta) { atomicAdd(word, delta); }

struct PolicyFeed {
    unsigned long long seq_end;   // written last
    unsigned long long allowed;   // TC_CNT_ALLOWED, folded from the shards
    unsigned long long swept;     // TC_CNT_SWEPT
    unsigned long long entries;   // string mode: bound keys (capacity - free slots); slot mode: TC_CNT_LIVE_SLOTS (as of the last sweep)
    unsigned long long free_slots; // string mode
    long long last_now;           // the call's last timestamp
    unsigned long long seq_begin; // written first
    unsigned long long inserted;  // TC_CNT_KEYS_INSERTED: keys bound so far (string mode)
};
static __global__ __launch_bounds__(ev::NSHARD) void k_policy_feed(const unsigned long long* __restrict__ counters, PolicyFeed* __restrict__ feed,
                                                                   unsigned long long seq, const int64_t* __restrict__ now_last, int64_t now_scalar,
                                                                   const int* __restrict__ free_top, uint64_t capacity) {
    __shared__ unsigned long long s[ev::NSHARD / 64];
    const unsigned long long* shard = counters + (TC_CNT_COUNT + 1) + threadIdx.x * ev::SHARD_WORDS;
    unsigned long long v = __hip_atomic_load(shard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long allowed = 0;
    for (int w = 0; w < ev::NSHARD / 64; ++w) allowed += s[w];
    volatile PolicyFeed* f = feed;
    f->seq_begin = seq;
    __threadfence_system();
    f->allowed = allowed;
    f->swept = counters[TC_CNT_SWEPT];
    if (free_top != nullptr) {
        const int top = *free_top;
        const unsigned long long fr = top < 0 ? 0ull : (unsigned long long)top;
        f->free_slots = fr;
        f->entries = capacity - (fr < capacity ? fr : capacity);
    } else {
        f->free_slots = 0;
        f->entries = counters[TC_CNT_LIVE_SLOTS];
    }
    f->last_now = now_last != nullptr ? (long long)*now_last : (long long)now_scalar;
    f->inserted = __hip_atomic_load(&counters[TC_CNT_KEYS_INSERTED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    f->seq_end = seq;
}

} // namespace mk
