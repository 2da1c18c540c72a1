// partbench.hip -- round 6: the range path (rp::k_tile_part, 1 024 threads per tile of 4 096 requests, 512 ranges + rs::k_finish,
// 512-thread blocks) and its RANK form (rp::k_tile_part<PART_RANK> + rs::k_finish with its scan blocks) against the LSD passes, on
// uniform and skewed batches -- each result checked on the host:  (round 4's first half, rs::k_tile_ranges with 256 ranges, is gone: its
// last figures beside the new kernel's are in profiles/r06_v1_partbench.txt) the plain form against std::sort, the hot form against
// [the cold requests sorted by (slot, index)]; every hot request's hot_P[tile][id] + rank inside its tile == its rank among its slot's requests
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/partbench.hip -o tools/bin/partbench && tools/bin/partbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>
#include "../throttlecrab_amd/csrc/radix_sort.hpp"
#include "../throttlecrab_amd/csrc/range_part.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static std::vector<uint32_t> make(const std::string& d, uint32_t n, uint32_t cap, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<uint32_t> h(n);
    std::vector<double> cdf;
    std::vector<uint32_t> perm;
    if (d == "zipf") { // Zipf(1.1) over the key space, ranks scattered (throttlecrab_amd/workload.py)
        const uint32_t K = 1u << 20; // the head by table, the tail uniform over the rest (its mass is spread thin anyway)
        cdf.resize(K);
        double acc = 0;
        for (uint32_t k = 0; k < K; ++k) acc += std::pow((double)(k + 1), -1.1), cdf[k] = acc;
        const double tail = (std::pow((double)K, -0.1) - std::pow((double)cap, -0.1)) / 0.1;
        for (auto& c : cdf) c /= (acc + tail);
    }
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t r = rng();
        if (d == "uniform") h[i] = r % cap;
        else if (d == "zipf") {
            const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
            const auto it = std::lower_bound(cdf.begin(), cdf.end(), u);
            const uint64_t rank = it == cdf.end() ? (uint64_t)cdf.size() + rng() % (cap - cdf.size()) : (uint64_t)(it - cdf.begin());
            h[i] = (uint32_t)((rank * 2654435761ull + 12345ull) % cap); // (an odd multiplier: a bijection mod 10^7? no -- close enough to scattered; collisions only merge keys)
        } else if (d == "hot") h[i] = (r % 100 < 40) ? 1234567u % cap : (uint32_t)((r >> 8) % cap);
        else if (d == "same") h[i] = cap - 1;
        else if (d == "few") h[i] = (uint32_t)((r % 7) * (cap / 7));
        else if (d == "dups") h[i] = (r % 1000 < 3) ? (uint32_t)(777777u + (r >> 30) % 5u) : (uint32_t)((r >> 8) % cap);
        else if (d == "edges") h[i] = (r & 1) ? (uint32_t)(r % 4 == 1 ? cap + 5 : cap - 1) : (uint32_t)((r >> 8) % 3);
        else if (d == "mid") h[i] = (r % 100 < 50) ? (uint32_t)((r >> 20) % 3000u * 3001u % cap) : (uint32_t)((r >> 8) % cap); // 3 000 slots with ~175 requests each
        else h[i] = r % cap;
    }
    return h;
}

int main(int argc, char** argv) {
    const uint32_t cap = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    const uint32_t mul = rs::range_mul(cap), width = rs::range_width(mul);
    const int sub_passes = width <= 256 ? 1 : 2;
    const uint32_t NMAX = 1u << 21, OLD_ITEMS = 16, old_tile = rs::THREADS * OLD_ITEMS;
    uint32_t* d_slot; uint64_t *a, *b, *c3; uint32_t *wsmem, *table, *totals; unsigned long long* hint; rp::HotDev* hotd; uint32_t *d_info, *d_P, *d_hn; uint32_t tpar = 0, parity = 0;
    const uint32_t max_tiles_old = (NMAX + old_tile - 1) / old_tile, max_tiles = (NMAX + rp::PT_TILE - 1) / rp::PT_TILE;
    const size_t words = rs::workspace_words(max_tiles_old);
    CK(hipMalloc(&d_slot, NMAX * 4)); CK(hipMalloc(&a, NMAX * 8)); CK(hipMalloc(&b, NMAX * 8)); CK(hipMalloc(&c3, NMAX * 8)); CK(hipMalloc(&wsmem, words * 4));
    CK(hipMalloc(&table, (size_t)max_tiles * rp::NB_HOT * 4)); CK(hipMalloc(&totals, 2 * rp::NB_HOT * 4)); CK(hipMemset(totals, 0, 2 * rp::NB_HOT * 4));
    CK(hipMalloc(&hotd, sizeof(rp::HotDev))); CK(hipMemset(hotd, 0, sizeof(rp::HotDev)));
    CK(hipMalloc(&d_info, NMAX * 4)); CK(hipMalloc(&d_P, (size_t)max_tiles * rp::HOT_MAX * 4)); CK(hipMalloc(&d_hn, (rp::HOT_MAX + 8) * 4));
    CK(hipHostMalloc((void**)&hint, 64, hipHostMallocDefault));
    CK(hipMemset(wsmem, 0, words * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t ev[6]; for (auto& e : ev) CK(hipEventCreate(&e));
    int bad = 0;
    struct Case { const char* dist; uint32_t n; uint32_t hot_min; }; // hot_min: slots with at least this many requests are on the hot list (0: plain forms only)
    const Case cases[] = {{"uniform", 1u << 20, 0}, {"uniform", 1u << 20, 4}, {"zipf", 1u << 20, 32}, {"zipf", 1u << 20, 64}, {"zipf", 1u << 18, 32}, {"zipf", 1572864, 32},
                          {"hot", 1u << 20, 32}, {"same", 200000, 32}, {"same", 1u << 20, 32}, {"few", 1u << 20, 32}, {"dups", 1u << 20, 32}, {"mid", 1u << 20, 32},
                          {"edges", 100000, 32}, {"uniform", 5000, 0}, {"uniform", 4097, 2}, {"zipf", 20000, 8}, {"uniform", 1, 0}};
    for (const Case& cs : cases) {
        const uint32_t n = cs.n;
        std::vector<uint32_t> h = make(cs.dist, n, cap, 1 + n);
        CK(hipMemcpy(d_slot, h.data(), n * 4, hipMemcpyHostToDevice));
        // the hot list: true counts, heaviest first
        rp::HotList hl;
        hl.count = 0;
        std::unordered_map<uint32_t, uint32_t> hot_id;
        if (cs.hot_min) {
            std::unordered_map<uint32_t, uint32_t> cnt;
            for (uint32_t s : h) if (s < cap) cnt[s]++;
            std::vector<std::pair<uint32_t, uint32_t>> v;
            for (auto& kv : cnt) if (kv.second >= cs.hot_min) v.emplace_back(kv.second, kv.first);
            std::sort(v.begin(), v.end(), [](auto& x, auto& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; });
            if (v.size() > rp::HOT_MAX) v.resize(rp::HOT_MAX);
            for (auto& kv : v) hot_id[kv.second] = hl.count, hl.slot[hl.count++] = kv.second;
        }
        std::vector<uint64_t> ref(n), ref_hot;
        for (uint32_t i = 0; i < n; ++i) ref[i] = ((uint64_t)std::min(h[i], cap) << 32) | i;
        if (hl.count) {
            std::vector<uint64_t> cold;
            std::vector<std::vector<uint64_t>> runs(hl.count);
            for (uint32_t i = 0; i < n; ++i) {
                const auto it = hot_id.find(h[i]);
                if (it != hot_id.end()) runs[it->second].push_back(ref[i]);
                else cold.push_back(ref[i]);
            }
            std::sort(cold.begin(), cold.end());
            ref_hot = cold; // (the rank form's sorted part)
        }
        std::sort(ref.begin(), ref.end());
        const uint32_t tiles_old = (n + old_tile - 1) / old_tile, tiles = (n + rp::PT_TILE - 1) / rp::PT_TILE;
        std::vector<uint64_t> out(n);
        const int iters = 12;
        for (int mode = 0; mode < 4; ++mode) { // 0: LSD passes, 1: round 4's range path, 2: k_tile_part + k_finish, 3: the hot form
            if (mode == 3 && !hl.count) continue;
            if (mode == 1) continue; // (round 4's kernels: removed)
            if (mode >= 2 && tiles > (uint32_t)rs::FIN_THREADS) continue;
            float acc[4] = {0, 0, 0, 0};
            if (mode == 3) hipLaunchKernelGGL(rp::k_hot_install, dim3(1), dim3(rp::PT_THREADS), 0, 0, hl, hotd);
            for (int it = 0; it < iters + 2; ++it) {
                const rs::Workspace ws = rs::carve(wsmem, parity, max_tiles_old);
                uint32_t* tot = totals + rp::NB_HOT * tpar;
                uint32_t* tot_next = totals + rp::NB_HOT * (tpar ^ 1u);
                CK(hipEventRecord(ev[0]));
                if (mode == 0) {
                    parity ^= 1;
                    hipLaunchKernelGGL(rs::k_hist<rs::HIST_THREADS>, dim3(rs::HIST_BLOCKS), dim3(rs::HIST_THREADS), 0, 0, d_slot, n, cap, 3, ws, tiles_old, nullptr, 0u, (uint8_t*)nullptr, 0u, mul);
                    CK(hipEventRecord(ev[1]));
                    hipLaunchKernelGGL((rs::k_onesweep<OLD_ITEMS, true>), dim3(tiles_old), dim3(rs::THREADS), 0, 0, d_slot, (const uint64_t*)nullptr, a, n, cap, 0, ws, nullptr, 0u, hint);
                    hipLaunchKernelGGL((rs::k_onesweep<OLD_ITEMS, false>), dim3(tiles_old), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, a, b, n, cap, 1, ws, nullptr, 0u, (unsigned long long*)nullptr);
                    CK(hipEventRecord(ev[2]));
                    hipLaunchKernelGGL((rs::k_onesweep<OLD_ITEMS, false>), dim3(tiles_old), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, b, a, n, cap, 2, ws, nullptr, 0u, (unsigned long long*)nullptr);
                    CK(hipEventRecord(ev[3]));
                } else {
                    const bool hotm = mode == 3;
                    const uint32_t stride = hotm ? rp::NB_HOT : rp::NR;
                    if (hotm) hipLaunchKernelGGL((rp::k_tile_part<rp::PART_RANK>), dim3(tiles), dim3(rp::PT_THREADS), 0, 0, d_slot, b, table, stride, tot, n, cap, mul, (uint8_t*)nullptr, 0u, (const rp::HotDev*)hotd, d_info);
                    else hipLaunchKernelGGL((rp::k_tile_part<rp::PART_PLAIN>), dim3(tiles), dim3(rp::PT_THREADS), 0, 0, d_slot, b, table, stride, tot, n, cap, mul, (uint8_t*)nullptr, 0u, (const rp::HotDev*)nullptr, (uint32_t*)nullptr);
                    CK(hipEventRecord(ev[1]));
                    CK(hipEventRecord(ev[2]));
                    hipLaunchKernelGGL(rs::k_finish<false>, dim3(rs::NRANGE + (hotm ? rp::HOT_MAX / 32u : 0u)), dim3(rs::FIN_THREADS), 0, 0, (const uint64_t*)b, (const uint32_t*)table, a, c3, (const uint32_t*)tot, tot_next, n, tiles,
                                       rp::PT_TILE, mul, sub_passes, hint, stride, rp::NB_HOT, hotm ? d_P : (uint32_t*)nullptr, hotm ? d_hn : (uint32_t*)nullptr, rp::HOT_MAX, (unsigned long long*)nullptr);
                    tpar ^= 1u;
                    CK(hipEventRecord(ev[3]));
                }
                CK(hipEventSynchronize(ev[3]));
                CK(hipGetLastError());
                if (it >= 2) for (int k = 0; k < 3; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
            }
            CK(hipMemcpy(out.data(), a, (size_t)n * 8, hipMemcpyDeviceToHost));
            bool ok;
            if (mode == 3) {
                // the sorted part: the cold requests; every hot request: its id and its rank among its slot's requests
                const size_t n_cold = ref_hot.size();
                std::vector<uint32_t> info(n), P((size_t)tiles * rp::HOT_MAX), hn(rp::HOT_MAX + 1);
                CK(hipMemcpy(info.data(), d_info, (size_t)n * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(P.data(), d_P, P.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hn.data(), d_hn, hn.size() * 4, hipMemcpyDeviceToHost));
                ok = hn[rp::HOT_MAX] == n_cold && std::equal(ref_hot.begin(), ref_hot.end(), out.begin());
                std::vector<uint32_t> seen(hl.count, 0);
                for (uint32_t i = 0; i < n && ok; ++i) {
                    const auto it = hot_id.find(h[i]);
                    if (it == hot_id.end()) ok = info[i] == rp::HOT_NONE;
                    else ok = (info[i] >> 16) == it->second && P[(size_t)(i / rp::PT_TILE) * rp::HOT_MAX + it->second] + (info[i] & 0xFFFFu) == seen[it->second]++;
                }
                for (uint32_t k = 0; k < hl.count && ok; ++k) ok = hn[k] == seen[k];
            } else {
                ok = out == ref;
            }
            const std::vector<uint64_t>& want = ref;
            const unsigned long long hv = *(volatile unsigned long long*)hint;
            static const char* names[4] = {"lsd", "range4", "part", "rank"};
            static const char* cols[4][3] = {{"hist", "p0+p1", "p2"}, {"tiles", "-", "finish"}, {"part", "-", "finish"}, {"part", "-", "finish"}};
            printf("%-8s n=%8u hot %4u %-7s %-6s %5.1f us  %-6s %5.1f  %-6s %5.1f  total %6.1f   largest range %u  %s\n", cs.dist, n, mode == 3 ? hl.count : 0u, names[mode], cols[mode][0],
                   1e3 * acc[0] / iters, cols[mode][1], 1e3 * acc[1] / iters, cols[mode][2], 1e3 * acc[2] / iters, 1e3 * (acc[0] + acc[1] + acc[2]) / iters, (unsigned)(hv & 0xFFFFFFFFu),
                   ok ? "ok" : "WRONG");
            if (!ok && mode != 3) {
                uint32_t shown = 0;
                for (uint32_t i = 0; i < n && shown < 4; ++i)
                    if (out[i] != want[i]) { printf("   first differences at %u: got %016llx want %016llx\n", i, (unsigned long long)out[i], (unsigned long long)want[i]); ++shown; }
            }
        }
    }
    printf(bad ? "FAILED: %d wrong results\n" : "all grouped correctly (%d)\n", bad);
    return bad ? 1 : 0;
}
