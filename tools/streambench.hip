// streambench.hip -- would the evaluation gain from STREAMING its window of the TAT column through LDS instead of gathering cells?
// A lean-evaluation block holds 512 sorted positions whose slots span ~4 900 consecutive slots (39 KB of the 8-byte column, 81 % of
// its 128-byte lines touched).  gather: every thread loads its 2 cells (8 B each) from global memory, stores them back -- the
// memory pattern of k_eval_sorted_lean.  stream: the block first copies its whole window into LDS with 16-byte coalesced loads,
// takes its cells from LDS, stores back to global memory.  Same stores, same grid (2 048 x 256).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/streambench.hip -o tools/bin/streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
constexpr int BS = 256, ITEMS = 2;
__global__ __launch_bounds__(BS, 8) void k_gather(long long* __restrict__ tab, const unsigned* __restrict__ slots, unsigned n, long long add) {
    const unsigned base = blockIdx.x * BS * ITEMS;
    unsigned s[ITEMS];
    long long v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) s[j] = slots[min(base + j * BS + threadIdx.x, n - 1)];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) v[j] = tab[s[j]];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) tab[s[j]] = v[j] + add;
}
// round 6: the same gathers and stores over a 4-BYTE column (a TAT kept as a 32-bit delta against a per-engine
// epoch would make the column 40 MB for 10 M keys instead of 80): how much of the bare pattern's time is bytes?
__global__ __launch_bounds__(BS, 8) void k_gather4(unsigned* __restrict__ tab, const unsigned* __restrict__ slots, unsigned n, unsigned add) {
    const unsigned base = blockIdx.x * BS * ITEMS;
    unsigned s[ITEMS], v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) s[j] = slots[min(base + j * BS + threadIdx.x, n - 1)];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) v[j] = tab[s[j]];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) tab[s[j]] = v[j] + add;
}
template <int LDS_WORDS>
__global__ __launch_bounds__(BS) void k_stream(long long* __restrict__ tab, const unsigned* __restrict__ slots, unsigned n, long long add, unsigned cap) {
    __shared__ long long win[LDS_WORDS];
    const unsigned base = blockIdx.x * BS * ITEMS;
    unsigned s[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) s[j] = slots[min(base + j * BS + threadIdx.x, n - 1)];
    // window: [first slot of the block rounded down to 2, last slot]
    __shared__ unsigned s_lo, s_hi;
    if (threadIdx.x == 0) s_lo = slots[min(base, n - 1)] & ~15u;
    if (threadIdx.x == BS - 1) s_hi = slots[min(base + BS * ITEMS - 1, n - 1)];
    __syncthreads();
    const unsigned lo = s_lo, cnt = min(s_hi - lo + 1u, (unsigned)LDS_WORDS);
    const longlong2* src = reinterpret_cast<const longlong2*>(tab + lo);
    longlong2* dst = reinterpret_cast<longlong2*>(win);
    for (unsigned i = threadIdx.x; i < (cnt + 1) / 2; i += BS) dst[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned o = s[j] - lo;
        const long long v = o < (unsigned)LDS_WORDS ? win[o] : tab[s[j]];
        tab[s[j]] = v + add;
    }
}
int main(int argc, char** argv) {
    const unsigned cap = 10000000, n = 1u << 20;
    std::vector<unsigned> h(n);
    std::mt19937_64 rng(1);
    for (auto& x : h) x = (unsigned)(rng() % cap);
    std::sort(h.begin(), h.end());
    long long* tab;
    unsigned* sl;
    hipMalloc(&tab, (size_t)cap * 8 + 1024);
    hipMalloc(&sl, n * 4);
    hipMemset(tab, 0, (size_t)cap * 8 + 1024);
    hipMemcpy(sl, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const dim3 grid(n / (BS * ITEMS));
    unsigned* tab4;
    hipMalloc(&tab4, (size_t)cap * 4 + 1024);
    hipMemset(tab4, 0, (size_t)cap * 4 + 1024);
    for (int var = 0; var < 4; ++var) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(a);
            if (var == 0) hipLaunchKernelGGL(k_gather, grid, dim3(BS), 0, 0, tab, sl, n, 1LL);
            else if (var == 1) hipLaunchKernelGGL(k_stream<6144>, grid, dim3(BS), 0, 0, tab, sl, n, 1LL, cap);
            else if (var == 2) hipLaunchKernelGGL(k_stream<5120>, grid, dim3(BS), 0, 0, tab, sl, n, 1LL, cap);
            else hipLaunchKernelGGL(k_gather4, grid, dim3(BS), 0, 0, tab4, sl, n, 1u);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep >= 2) best = std::min(best, ms), sum += ms;
        }
        printf("%s: best %.1f us, mean %.1f us\n", var == 0 ? "gather, 8-byte column (8 blocks/CU)" : (var == 1 ? "stream via 48 KB LDS (3 blocks/CU)" : (var == 2 ? "stream via 40 KB LDS (4 blocks/CU)" : "gather, 4-BYTE column (8 blocks/CU)")), best * 1e3,
               sum / 10 * 1e3);
    }
    return 0;
}
