// route_kernels.hpp -- routing of a GLOBAL request stream to the GPU that owns each key (gfx950).
//
// The reference has no distributed mode ("use client-side sharding by key", README.md:247-249); here keys are
// independent units, so a key space of `world * keys_per_shard` global ids is sharded with no data-path
// collective: every global id maps to (owner, shard-local slot) by a BIJECTION of [0, world * keys_per_shard)
//     x = (id * mul + add) mod (world * keys_per_shard),   owner = x mod world,   slot = x div world
// (mul coprime with the modulus: every shard gets exactly keys_per_shard dense slots, consecutive ids land
// on different owners, and no routing table is needed).  A GPU is handed the global batch -- or its part of
// it -- and keeps what it owns:
//   k_route_count    requests per destination, per 4096-request tile
//   k_route_scan     prefix of those counts over the tiles + destination totals            (one block)
//   k_route_scatter  stable compaction per destination (wave ballots): shard-local slots and, optionally, the
//                    requests' positions in the global batch, in request order
// Stable: the requests of a key keep their order, which is all the sequence semantics need.
// `only` >= 0 writes that destination's requests alone (what one rank of bench.py --gpus N does); -1 writes every
// destination's segment one after the other (a front end that forwards segments over xGMI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

constexpr int THREADS = 256, ITEMS = 16;
constexpr uint32_t TILE = THREADS * ITEMS;
constexpr uint32_t MAX_WORLD = 64;

struct Map {
    uint64_t modulus; // world * keys_per_shard
    uint64_t mul, add;
    uint32_t world;
};
// mul < 2^24 and modulus < 2^39: the product stays below 2^63
__host__ __device__ inline uint64_t permute(const Map& m, uint64_t id) { return (id % m.modulus * m.mul + m.add) % m.modulus; }

inline uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
// the map of a (world, keys_per_shard) pair: the same on every rank, the host mirror and the device
inline bool make_map(uint32_t world, uint64_t keys_per_shard, Map* out) {
    if (world == 0 || world > MAX_WORLD || keys_per_shard == 0 || keys_per_shard > ((uint64_t)1 << 32)) return false;
    const uint64_t modulus = (uint64_t)world * keys_per_shard;
    if (modulus >= ((uint64_t)1 << 39)) return false;
    static const uint64_t primes[] = {10000019ull, 9999991ull, 8388617ull, 7654321ull, 6700417ull, 5000011ull, 4999999ull, 3999971ull};
    for (uint64_t p : primes)
        if (gcd64(p, modulus) == 1) {
            out->modulus = modulus;
            out->mul = p;
            out->add = 0x5bd1e995ull % modulus;
            out->world = world;
            return true;
        }
    return false;
}

struct Work {
    uint32_t* tile_cnt; // [tiles][world] -> exclusive prefix over the tiles
    uint32_t* totals;   // [world]
    uint32_t* starts;   // [world + 1] segment starts in the output (only < 0), else {0, totals[only]}
};

__global__ __launch_bounds__(THREADS) void k_route_count(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w) {
    __shared__ uint32_t s_c[MAX_WORLD];
    if (threadIdx.x < MAX_WORLD) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * TILE + threadIdx.x;
    uint32_t v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) v[j] = base + j * THREADS < n ? id[base + j * THREADS] : 0u;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (base + j * THREADS < n) atomicAdd(&s_c[permute(m, v[j]) % m.world], 1u);
    __syncthreads();
    if (threadIdx.x < m.world) w.tile_cnt[(size_t)blockIdx.x * m.world + threadIdx.x] = s_c[threadIdx.x];
}

// one block: thread d < world walks the tiles of destination d
__global__ __launch_bounds__(MAX_WORLD) void k_route_scan(Work w, uint32_t tiles, uint32_t world, int only) {
    __shared__ uint32_t s_tot[MAX_WORLD];
    const uint32_t d = threadIdx.x;
    uint32_t run = 0;
    if (d < world)
        for (uint32_t t = 0; t < tiles; ++t) {
            const uint32_t c = w.tile_cnt[(size_t)t * world + d];
            w.tile_cnt[(size_t)t * world + d] = run;
            run += c;
        }
    s_tot[d] = d < world ? run : 0u;
    if (d < world) w.totals[d] = run;
    __syncthreads();
    if (d == 0) {
        uint32_t at = 0;
        for (uint32_t k = 0; k < world; ++k) {
            w.starts[k] = only < 0 ? at : 0u;
            at += s_tot[k];
        }
        w.starts[world] = only < 0 ? at : s_tot[only];
    }
}

__global__ __launch_bounds__(THREADS) void k_route_scatter(const uint32_t* __restrict__ id, uint32_t n, Map m, Work w, int only,
                                                            uint32_t* __restrict__ out_slot, uint32_t* __restrict__ out_pos) {
    __shared__ uint32_t s_wave[THREADS / 64][MAX_WORLD]; // per-wave counts -> exclusive prefix over the waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (THREADS / 64) * MAX_WORLD; i += THREADS) (&s_wave[0][0])[i] = 0;
    __syncthreads();
    // wave-striped: step j of wave k covers requests tile*TILE + k*1024 + j*64 + lane (request order inside a wave)
    const uint32_t first = blockIdx.x * TILE + wave * (64 * ITEMS) + lane;
    uint32_t dest[ITEMS], slot[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = first + j * 64;
        const uint64_t x = pos < n ? permute(m, id[pos]) : 0ull;
        dest[j] = pos < n ? (uint32_t)(x % m.world) : 0xFFFFFFFFu;
        slot[j] = (uint32_t)(x / m.world);
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
            const uint32_t at = w.starts[d] + w.tile_cnt[(size_t)blockIdx.x * m.world + d] + s_wave[wave][d] + rank[j];
            out_slot[at] = slot[j];
            if (out_pos) out_pos[at] = pos;
        }
    }
}

} // namespace rt
