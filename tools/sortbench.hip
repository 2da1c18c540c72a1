// sortbench.hip -- times rs::k_hist / rs::k_onesweep alone (with phase ablations via -DRS_ABLATE=N)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "../throttlecrab_amd/csrc/radix_sort.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#ifndef SB_ITEMS
#define SB_ITEMS 16
#endif
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 20), cap = 10000000;
    const int passes = 3;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> h(n);
    for (auto& x : h) x = rng() % cap;
    uint32_t* d_slot; uint64_t *a, *b; uint32_t* wsmem;
    const uint32_t tile = rs::THREADS * SB_ITEMS, tiles = (n + tile - 1) / tile;
    const size_t words = rs::workspace_words(tiles);
    CK(hipMalloc(&d_slot, n * 4)); CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&wsmem, words * 4));
    CK(hipMemcpy(d_slot, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(wsmem, 0, words * 4));
    hipEvent_t ev[8]; for (auto& evx : ev) CK(hipEventCreate(&evx));
    float acc[4] = {0, 0, 0, 0}; const int iters = 20; uint32_t parity = 0;
    for (int it = 0; it < iters + 3; ++it) {
        const rs::Workspace ws = rs::carve(wsmem, parity, tiles); parity ^= 1;
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(rs::k_hist<rs::HIST_THREADS>, dim3(rs::HIST_BLOCKS), dim3(rs::HIST_THREADS), 0, 0, d_slot, n, cap, passes, ws, tiles, nullptr, 0u, (uint8_t*)nullptr, 0u);
        CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, true>), dim3(tiles), dim3(rs::THREADS), 0, 0, d_slot, (const uint64_t*)nullptr, a, n, cap, 0, ws, nullptr, 0u);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, false>), dim3(tiles), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, a, b, n, cap, 1, ws, nullptr, 0u);
        CK(hipEventRecord(ev[3]));
        hipLaunchKernelGGL((rs::k_onesweep<SB_ITEMS, false>), dim3(tiles), dim3(rs::THREADS), 0, 0, (const uint32_t*)nullptr, b, a, n, cap, 2, ws, nullptr, 0u);
        CK(hipEventRecord(ev[4]));
        CK(hipEventSynchronize(ev[4]));
        if (it >= 3) for (int k = 0; k < 4; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
    }
    printf("n=%u ABLATE=%d ITEMS=%d tiles=%u  hist %.1f us  pass0 %.1f  pass1 %.1f  pass2 %.1f\n", n, RS_ABLATE, SB_ITEMS, tiles,
           1e3 * acc[0] / iters, 1e3 * acc[1] / iters, 1e3 * acc[2] / iters, 1e3 * acc[3] / iters);
    if (RS_ABLATE == 0) {
        std::vector<uint64_t> out(n); CK(hipMemcpy(out.data(), a, n * 8, hipMemcpyDeviceToHost));
        std::vector<uint64_t> ref(n); for (uint32_t i = 0; i < n; ++i) ref[i] = ((uint64_t)h[i] << 32) | i;
        std::sort(ref.begin(), ref.end());
        printf("sorted correctly: %s\n", out == ref ? "yes" : "NO");
    }
    return 0;
}
