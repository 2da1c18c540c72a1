// pipeprobe.hip -- do two HIP streams share a dispatch pipe?  Kernel A (a grid four times what the chip holds, every block
// staying for 10 us) runs on stream `a`; a one-block kernel B on stream `b` stamps the clock.  If B's block is dispatched
// while A is still handing out blocks, B finishes a few us after A started; if B has to wait until A's last block has been
// dispatched, it finishes ~30 us later.  Streams are created in order, so the table shows which creation distances collide.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pipeprobe.hip -o tools/bin/pipeprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void k_occupy(long long ticks, long long* start) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) *start = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_stamp(long long* out) { *out = wall_clock64(); }
int main(int argc, char** argv) {
    const int NS = 10;
    const int main_idx = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t s[NS];
    long long* d;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));          // (the null stream takes a queue first, as it does at engine creation)
    CK(hipDeviceSynchronize());
    for (int i = 0; i < NS; ++i) {
        CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s[i], d + 2); // first use: the stream gets its queue now, in creation order
        CK(hipStreamSynchronize(s[i]));
    }
    printf("main = stream %d; delay of a one-block kernel on stream i behind the start of an 8192-block kernel on main (us):\n", main_idx);
    for (int i = 0; i < NS; ++i) {
        if (i == main_idx) continue;
        double worst = 0, best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k_occupy, dim3(8192), dim3(256), 0, s[main_idx], 1000LL, d);
            hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s[i], d + 1);
            CK(hipDeviceSynchronize());
            long long h[2];
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            const double us = (h[1] - h[0]) / 100.0;
            if (us > worst) worst = us;
            if (us < best) best = us;
        }
        printf("  stream %d: %7.1f .. %7.1f\n", i, best, worst);
    }
    return 0;
}
