// anyorder.hip -- does hipExtAnyOrderLaunch let a kernel start while the previous kernel of the SAME stream is still running on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx.)  And: what does a back-to-back hand-over of two 2048-block kernels cost with an
// in-kernel dependency instead of the stream's barrier?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/anyorder.hip -o tools/bin/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void k_long(long long ticks, long long* stamps) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void k_stamp(long long* stamps) {
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2] = wall_clock64();
}
int main() {
    long long* d;
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int any = 0; any < 2; ++any) {
        long long h[3];
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 64);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_long, dim3(2048), dim3(256), 0, s, 2000LL, d); // 20 us per block
            hipExtLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, d);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("anyorder=%d: long kernel %.1f us; second kernel started %.1f us after the first STARTED (%.1f us relative to its end)\n", any,
                   (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[2] - h[1]) / 100.0);
        }
    }
    return 0;
}
