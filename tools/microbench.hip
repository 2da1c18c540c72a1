// microbench.hip -- what do the memory patterns of the eval kernel cost on MI355X?
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct __attribute__((aligned(32))) Rec { long long a, b, c, d; };
constexpr int NB = 16; // distinct batches rotated through (fresh random slots per launch, like the real stream)

// mode bits: 1 = gather 32B record, 2 = store 16B back, 4 = scattered byte store by idx, 8 = coalesced byte store
template <int MODE>
__global__ __launch_bounds__(256) void k_pat(const uint64_t* __restrict__ sorted, uint32_t n, Rec* __restrict__ table,
                                             uint8_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint64_t e = sorted[k];
    const uint32_t slot = (uint32_t)(e >> 32), idx = (uint32_t)e;
    long long v = slot;
    if (MODE & 1) {
        const Rec r = table[slot];
        v = r.a + r.b + r.c + r.d;
    }
    if (MODE & 2) {
        long long* p = &table[slot].a;
        p[0] = v + 1;
        p[1] = v + 2;
    }
    if (MODE & 4) out[idx] = (uint8_t)(v & 1);
    if (MODE & 8) out[k] = (uint8_t)(v & 1);
}

// 16-byte records (cell only; 160 MB for 10 M keys -> fits the 256 MiB Infinity Cache)
struct __attribute__((aligned(16))) Rec16 { long long a, b; };
template <int MODE>
__global__ __launch_bounds__(256) void k_pat16(const uint64_t* __restrict__ sorted, uint32_t n, Rec16* __restrict__ table,
                                               uint8_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint64_t e = sorted[k];
    const uint32_t slot = (uint32_t)(e >> 32), idx = (uint32_t)e;
    long long v = slot;
    typedef long long v2ll __attribute__((ext_vector_type(2)));
    v2ll* cell = reinterpret_cast<v2ll*>(table + slot);
    if (MODE & 1) {
        const v2ll r = (MODE & 32) ? __builtin_nontemporal_load(cell) : *cell;
        v = r.x + r.y;
    }
    if (MODE & 2) {
        v2ll w;
        w.x = v + 1;
        w.y = v + 2;
        if (MODE & 64) *reinterpret_cast<long long*>(cell) = w.x; // 8 of the 16 bytes
        else if (MODE & 16) __builtin_nontemporal_store(w, cell);
        else *cell = w;
    }
    if (MODE & 4) out[idx] = (uint8_t)(v & 1);
    if (MODE & 8) out[k] = (uint8_t)(v & 1);
}
 // distinct batches rotated through (d_sorted holds NB batches back to back)
template <int MODE>
float run16(const uint64_t* d_sorted, uint32_t n, Rec16* table, uint8_t* out, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_pat16<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, d_sorted, n, table, out);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_pat16<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, d_sorted + (size_t)(i % NB) * n, n, table, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / iters;
}

template <int MODE>
float run(const uint64_t* d_sorted, uint32_t n, Rec* table, uint8_t* out, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_pat<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, d_sorted, n, table, out);
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_pat<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, d_sorted + (size_t)(i % NB) * n, n, table, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / iters;
}

int main(int argc, char** argv) {
    const uint32_t n = 1 << 20, cap = argc > 1 ? (uint32_t)atol(argv[1]) : 10000000;
    printf("== keys %u\n", cap);
    std::mt19937_64 rng(1);
    std::vector<uint64_t> h((size_t)n * NB);
    for (int b = 0; b < NB; ++b)
        for (uint32_t i = 0; i < n; ++i) h[(size_t)b * n + i] = ((uint64_t)(rng() % cap) << 32) | i;
    std::vector<uint64_t> hs = h;
    for (int b = 0; b < NB; ++b) std::sort(hs.begin() + (size_t)b * n, hs.begin() + (size_t)(b + 1) * n);
    uint64_t *d_sorted, *d_unsorted;
    Rec* table;
    uint8_t* out;
    CK(hipMalloc(&d_sorted, (size_t)n * NB * 8));
    CK(hipMalloc(&d_unsorted, (size_t)n * NB * 8));
    CK(hipMalloc(&table, (size_t)cap * sizeof(Rec)));
    CK(hipMalloc(&out, n));
    CK(hipMemset(table, 0, (size_t)cap * sizeof(Rec)));
    CK(hipMemcpy(d_sorted, hs.data(), (size_t)n * NB * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_unsorted, h.data(), (size_t)n * NB * 8, hipMemcpyHostToDevice));
    printf("pattern (1Mi requests over 10M x 32B records)             sorted_us  unsorted_us\n");
#define ROW(M, name) printf("%-58s %9.1f  %9.1f\n", name, run<M>(d_sorted, n, table, out, 32), run<M>(d_unsorted, n, table, out, 32));
    ROW(0, "read elems only");
    ROW(8, "read elems + coalesced byte store");
    ROW(4, "read elems + byte store scattered by idx");
    ROW(1 | 8, "gather 32B + coalesced byte store");
    ROW(1 | 4, "gather 32B + scattered byte store");
    ROW(1 | 2 | 8, "gather 32B + store 16B + coalesced byte store");
    ROW(1 | 2 | 4, "gather 32B + store 16B + scattered byte store");
    Rec16* t16;
    CK(hipMalloc(&t16, (size_t)cap * sizeof(Rec16)));
    CK(hipMemset(t16, 0, (size_t)cap * sizeof(Rec16)));
    printf("-- 16-byte records (160 MB table)\n");
#define ROW16(M, name) printf("%-58s %9.1f  %9.1f\n", name, run16<M>(d_sorted, n, t16, out, 32), run16<M>(d_unsorted, n, t16, out, 32));
    ROW16(1 | 8, "gather 16B + coalesced byte store");
    ROW16(1 | 2 | 8, "gather 16B + store 16B + coalesced byte store");
    ROW16(1 | 2 | 4, "gather 16B + store 16B + scattered byte store");
    ROW16(2 | 8, "store 16B only (no gather) + coalesced byte store");
    ROW16(1 | 2 | 8 | 16, "gather 16B + NT store 16B + coalesced byte store");
    ROW16(1 | 2 | 8 | 32, "NT gather 16B + store 16B + coalesced byte store");
    ROW16(1 | 2 | 8 | 16 | 32, "NT gather 16B + NT store 16B + coalesced byte store");
    ROW16(1 | 2 | 8 | 64, "gather 16B + store 8B + coalesced byte store");
    return 0;
}
