// cumask_probe.hip -- what a hipExtStreamCreateWithCUMask mask does on this chip: which (XCC, SE, CU) the blocks of
// a masked stream land on, and how a latency-bound kernel's time changes with the number of CUs.
// build: hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__global__ void k_where(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) << 16 | (hw & 0xFFFF);
}

int main() {
    uint32_t* d;
    const int blocks = 2048;
    CK(hipMalloc(&d, blocks * 4));
    std::vector<uint32_t> h(blocks);
    struct M {
        const char* name;
        uint32_t w[8];
    };
    M masks[] = {{"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
                 {"w0", {~0u, 0, 0, 0, 0, 0, 0, 0}},
                 {"w0-w3", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
                 {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
                 {"low byte of every word", {0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu}},
                 {"bits 0-7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}},
                 {"bits 0-63", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
                 {"3/4: w0-w5", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0, 0}}};
    for (const M& m : masks) {
        hipStream_t s;
        hipError_t rc = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (rc != hipSuccess) {
            printf("%-24s create failed: %s\n", m.name, hipGetErrorString(rc));
            (void)hipGetLastError();
            continue;
        }
        CK(hipMemsetAsync(d, 0xFF, blocks * 4, s));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 200); // warm
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 500); // 5 us per block
        CK(hipEventRecord(b, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
        std::set<uint32_t> cus;
        int per_xcc[16] = {0};
        for (uint32_t v : h) {
            const uint32_t xcc = v >> 16, hw = v & 0xFFFF;
            const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert(xcc << 12 | se << 8 | sh << 4 | cu);
            per_xcc[xcc & 15]++;
        }
        printf("%-24s %7.1f us  distinct (xcc,se,sh,cu) = %3zu   blocks per xcc:", m.name, ms * 1e3, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
        printf("\n");
        CK(hipStreamDestroy(s));
    }
    return 0;
}
