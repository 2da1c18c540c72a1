// sortbench.hip -- the grouping kernels alone: rs::k_hist + three rs::k_onesweep LSD passes against the range path
// (rs::k_hist counting ranges + one partition pass + rs::k_finish), each checked against std::sort and timed with events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSB_ITEMS=16] tools/sortbench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <string>
#include <vector>
#define RS_TIMING 1
#include "../throttlecrab_amd/csrc/radix_sort.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#ifndef SB_ITEMS
#define SB_ITEMS 16
#endif

static std::vector<uint32_t> make(const char* dist, uint32_t n, uint32_t cap, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<uint32_t> h(n);
    const std::string d(dist);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t r = rng();
        if (d == "uniform") h[i] = r % cap;
        else if (d == "hot") h[i] = (r % 100 < 40) ? 1234567u % cap : (uint32_t)((r >> 8) % cap);          // one key with 40 % of the batch
        else if (d == "hot_ranges") h[i] = (r % 100 < 70) ? (uint32_t)(cap / 3 + (r >> 20) % 3000) : (uint32_t)((r >> 8) % cap);  // one RANGE with 70 %, many keys in it
        else if (d == "dups") h[i] = (r % 1000 < 3) ? (uint32_t)(777777u + (r >> 30) % 5u) : (uint32_t)((r >> 8) % cap);   // five slots with ~630 requests each among uniform ones
        else if (d == "pairs") h[i] = (uint32_t)(((r >> 8) % cap) & ~7u);                                                     // every slot a multiple of 8: ~8x the duplicates
        else if (d == "same") h[i] = cap - 1;
        else if (d == "few") h[i] = (uint32_t)((r % 7) * (cap / 7));
        else if (d == "edges") h[i] = (r & 1) ? (uint32_t)(r % 4 == 1 ? cap + 5 : cap - 1) : (uint32_t)((r >> 8) % 3);             // out-of-range slots (clamped to cap), both ends
        else h[i] = r % cap;
    }
    return h;
}

int main(int argc, char** argv) {
    const uint32_t cap = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000u;
    const int passes = 3;
    const uint32_t mul = rs::range_mul(cap);
    const uint32_t width = rs::range_width(mul);
    const int sub_passes = width <= 256 ? 1 : 2;
    printf("cap %u  range mul %u  widest range %u slots  sub passes %d  lo(1)=%u lo(255)=%u\n", cap, mul, width, sub_passes, rs::range_lo(1, mul), rs::range_lo(255, mul));
    if (width > 65536) { printf("key space too wide for the range path\n"); return 1; }
    const uint32_t NMAX = 1u << 21;
    uint32_t* d_slot; uint64_t *a, *b, *c3; uint32_t* wsmem; unsigned long long* hint; uint32_t* totals; uint32_t tpar = 0;
    const uint32_t tile = rs::THREADS * SB_ITEMS, max_tiles = (NMAX + tile - 1) / tile;
    const size_t words = rs::workspace_words(max_tiles);
    CK(hipMalloc(&d_slot, NMAX * 4)); CK(hipMalloc(&a, NMAX * 8)); CK(hipMalloc(&b, NMAX * 8)); CK(hipMalloc(&c3, NMAX * 8)); CK(hipMalloc(&wsmem, words * 4));
    CK(hipMalloc(&totals, 512 * 4)); CK(hipMemset(totals, 0, 512 * 4));
    CK(hipHostMalloc((void**)&hint, 64, hipHostMallocDefault));
    CK(hipMemset(wsmem, 0, words * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t ev[8]; for (auto& evx : ev) CK(hipEventCreate(&evx));
    uint32_t parity = 0;
    int bad = 0;
    struct Case { const char* dist; uint32_t n; };
    const Case cases[] = {{"uniform", 1u << 20}, {"dups", 1u << 20}, {"pairs", 1u << 20}, {"uniform", 5000}, {"uniform", 300000}, {"uniform", 256}, {"uniform", 1}, {"uniform", 4097},
                          {"uniform", 1u << 21}, {"hot", 1u << 20}, {"hot_ranges", 1u << 20}, {"same", 200000}, {"same", 70000}, {"few", 1u << 20},
                          {"edges", 100000}, {"uniform", 1572864}};
    for (const Case& cs : cases) {
        const uint32_t n = cs.n;
        std::vector<uint32_t> h = make(cs.dist, n, cap, 1 + n);
        CK(hipMemcpy(d_slot, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<uint64_t> ref(n);
        for (uint32_t i = 0; i < n; ++i) ref[i] = ((uint64_t)std::min(h[i], cap) << 32) | i;
        std::sort(ref.begin(), ref.end());
        const uint32_t tiles = (n + tile - 1) / tile;
        float acc[8] = {0};
        const int iters = 12;
        std::vector<uint64_t> out(n);
        for (int mode = 0; mode < 2; ++mode) { // 0: three LSD passes, 1: range path
            for (int k = 0; k < 8; ++k) acc[k] = 0;
            for (int it = 0; it < iters + 2; ++it) {
                const rs::Workspace ws = rs::carve(wsmem, parity, max_tiles); parity ^= 1;
                CK(hipEventRecord(ev[0]));
                if (mode == 0) {
                    hipLaunchKernelGGL(rs::k_hist<rs::HIST_THREADS>, dim3(rs::HIST_BLOCKS), dim3(rs::HIST_THREADS), 0, 0, d_slot, n, cap, passes, ws, tiles, nullptr, 0u, (uint8_t*)nullptr, 0u, mul);
                    CK(hipEventRecord(ev[1]));
                    hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, true>), dim3(tiles), dim3(rs::THREADS), 0, 0, d_slot, (const uint64_t*)nullptr, a, n, cap, 0, ws, nullptr, 0u, hint);
                    CK(hipEventRecord(ev[2]));
                    hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, false>), dim3(tiles), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, a, b, n, cap, 1, ws, nullptr, 0u, (unsigned long long*)nullptr);
                    CK(hipEventRecord(ev[3]));
                    hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, false>), dim3(tiles), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, b, a, n, cap, 2, ws, nullptr, 0u, (unsigned long long*)nullptr);
                    CK(hipEventRecord(ev[4]));
                } else {
                    CK(hipEventRecord(ev[1]));
                    if (tiles > (uint32_t)rs::FIN_THREADS) { printf("(n too large for the range path)\n"); break; }
                    hipLaunchKernelGGL((rs::k_tile_ranges<SB_ITEMS>), dim3(tiles), dim3(rs::THREADS), 0, 0, d_slot, b, ws.status, totals + 256 * tpar, n, cap, mul, (uint8_t*)nullptr, 0u);
                    CK(hipEventRecord(ev[2]));
                    hipLaunchKernelGGL(rs::k_finish, dim3(rs::RADIX), dim3(rs::FIN_THREADS), 0, 0, (const uint64_t*)b, (const uint32_t*)ws.status, a, c3, (const uint32_t*)(totals + 256 * tpar), totals + 256 * (tpar ^ 1u), n, tiles, tile, mul, sub_passes, hint);
                    tpar ^= 1u;
                    CK(hipEventRecord(ev[3]));
                    CK(hipEventRecord(ev[4]));
                }
                CK(hipEventSynchronize(ev[4]));
                CK(hipGetLastError());
                if (it >= 2) for (int k = 0; k < 4; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
            }
            CK(hipMemcpy(out.data(), a, (size_t)n * 8, hipMemcpyDeviceToHost));
            const bool ok = out == ref;
            bad += !ok;
            const unsigned long long hv = *(volatile unsigned long long*)hint;
            printf("%-10s n=%8u %-6s %s %5.1f us  %s %5.1f  %s %5.1f  %s %5.1f  total %6.1f   largest range %u (n %u)  %s\n", cs.dist, n, mode ? "range" : "lsd", mode ? "-   " : "hist",
                   1e3 * acc[0] / iters, mode ? "tiles" : "p0", 1e3 * acc[1] / iters, mode ? "finish" : "p1", 1e3 * acc[2] / iters, mode ? "-" : "p2", 1e3 * acc[3] / iters,
                   1e3 * (acc[0] + acc[1] + acc[2] + acc[3]) / iters, (unsigned)(hv & 0xFFFFFFFFu), (unsigned)(hv >> 32), ok ? "sorted" : "WRONG");
            if (mode == 1 && n == (1u << 20) && std::string(cs.dist) == "uniform") {
                long long st[2][16];
                CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_rs_stamps), sizeof st));
                printf("   k_tile_ranges phases (us, one block): zero+issue loads+fill %.2f | loads land %.2f | rank %.2f | barrier %.2f | prefix+table %.2f | stage %.2f | write %.2f\n",
                       (st[0][1] - st[0][0]) / 100.0, (st[0][2] - st[0][1]) / 100.0, (st[0][3] - st[0][2]) / 100.0, (st[0][4] - st[0][3]) / 100.0,
                       (st[0][5] - st[0][4]) / 100.0, (st[0][6] - st[0][5]) / 100.0, (st[0][7] - st[0][6]) / 100.0);
                printf("   k_finish phases, counting path (us, block 128): issue table %.2f | table lands %.2f | scan+publish %.2f | zero+expand %.2f | gather+count %.2f | scan counters %.2f | order+place in LDS %.2f | place %.2f | write %.2f\n",
                       (st[1][1] - st[1][0]) / 100.0, (st[1][2] - st[1][1]) / 100.0, (st[1][3] - st[1][2]) / 100.0, (st[1][4] - st[1][3]) / 100.0, (st[1][5] - st[1][4]) / 100.0,
                       (st[1][6] - st[1][5]) / 100.0, (st[1][12] - st[1][6]) / 100.0, (st[1][13] - st[1][12]) / 100.0, (st[1][14] - st[1][13]) / 100.0);
            }
            if (!ok) {
                uint32_t shown = 0;
                for (uint32_t i = 0; i < n && shown < 4; ++i)
                    if (out[i] != ref[i]) { printf("   first differences at %u: got %016llx want %016llx\n", i, (unsigned long long)out[i], (unsigned long long)ref[i]); ++shown; }
            }
        }
    }
    printf(bad ? "FAILED: %d wrong results\n" : "all sorted correctly (%d)\n", bad);
    return bad ? 1 : 0;
}
