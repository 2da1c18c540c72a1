// gapbench.hip -- what one cross-stream dependency in front of a kernel costs on the critical stream.
// The pipelined engine's main stream is: [wait for batch k's grouping (another stream)] [evaluation k], repeated; the rocprofv3
// timeline shows ~10 us between the end of one evaluation and the start of the next although the grouping finished long
// before.  This program times the same shape with a dummy 20 us kernel (grid 2048 x 256 like the evaluation) and several
// ways of expressing the dependency.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gapbench.hip -o tools/bin/gapbench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_busy(long long ticks, unsigned* sink) { // ~ticks of the 100 MHz wall clock
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (v == 0xFFFFFFFFu) *sink = v;
}
__global__ void k_side(long long ticks, unsigned* sink, unsigned* flag, unsigned value) { // a "grouping" kernel: busy, then publishes a word
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (v == 0xFFFFFFFFu) *sink = v;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the evaluation's grid with the dependency INSIDE the kernel: one lane per block polls the word
__global__ void k_busy_polling(long long ticks, unsigned* sink, const unsigned* flag, unsigned want) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < ticks) v += 1;
    if (v == 0xFFFFFFFFu) *sink = v;
}

__global__ void k_check(const unsigned* flag, unsigned want, unsigned* viol) {
    if (threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) atomicAdd(viol, 1u);
}

int main() {
    const int N = 200, AHEAD = 3;
    hipStream_t main_s, side[3];
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *sink, *flags;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&flags, 4096));
    CK(hipMemset(flags, 0, 4096));
    std::vector<hipEvent_t> ev(N + AHEAD + 1);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<hipEvent_t> mev(16);
    for (auto& e : mev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const long long EVAL = 2000, SIDE = 3000; // 20 us / 30 us
    const dim3 grid(2048), block(256), sgrid(256), sblock(256);
    auto run = [&](const char* name, int mode) {
        CK(hipDeviceSynchronize());
        CK(hipMemset(flags, 0, 4096));
        CK(hipDeviceSynchronize());
        // side work of step i runs on side[i % 3]; the main kernel of step i depends on it; the host is AHEAD steps ahead with the side work
        auto side_launch = [&](int i) {
            hipStream_t s = side[i % 3];
            if (mode == 1) { hipLaunchKernelGGL(k_side, sgrid, sblock, 0, s, SIDE, sink, (unsigned*)nullptr, 0u); CK(hipEventRecord(ev[i], s)); }
            else if (mode == 2 || mode == 6 || mode == 7) hipExtLaunchKernelGGL(k_side, sgrid, sblock, 0, s, nullptr, ev[i], 0, SIDE, sink, (unsigned*)nullptr, 0u);
            else if (mode == 3 || mode == 4 || mode == 5) hipLaunchKernelGGL(k_side, sgrid, sblock, 0, s, SIDE, sink, flags + (i % 64), (unsigned)(i / 64 + 1));
            else hipLaunchKernelGGL(k_side, sgrid, sblock, 0, s, SIDE, sink, (unsigned*)nullptr, 0u);
        };
        for (int i = 0; i < AHEAD; ++i) side_launch(i);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            side_launch(i + AHEAD);
            if (mode == 1 || mode == 2 || mode == 6 || mode == 7) CK(hipStreamWaitEvent(main_s, ev[i], 0));
            if (mode == 6 || mode == 7) { // the main kernel carries a stop event too (the engine's `consumed`), and the side stream of 3 steps later waits for it
                if (mode == 7 && i >= 3) CK(hipStreamWaitEvent(side[(i + AHEAD) % 3], mev[(i - 3) % 16], 0));
                hipExtLaunchKernelGGL(k_busy, grid, block, 0, main_s, nullptr, mev[i % 16], 0, EVAL, sink);
                continue;
            }
            if (mode == 3) CK(hipStreamWaitValue32(main_s, flags + (i % 64), (unsigned)(i / 64 + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
            if (mode == 4) hipLaunchKernelGGL(k_busy_polling, grid, block, 0, main_s, EVAL, sink, flags + (i % 64), (unsigned)(i / 64 + 1));
            else hipLaunchKernelGGL(k_busy, grid, block, 0, main_s, EVAL, sink);
        }
        CK(hipStreamSynchronize(main_s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        CK(hipDeviceSynchronize());
        printf("%-64s %7.1f us per step  (kernel 20 us: %5.1f us of hand-over)\n", name, us, us - 20.0);
    };
    // the base cost of a kernel boundary against the shape of the grid (no dependency at all)
    for (int shape = 0; shape < 6; ++shape) {
        const dim3 g[6] = {dim3(2048), dim3(4096), dim3(1024), dim3(512), dim3(256), dim3(512)}, b[6] = {dim3(256), dim3(256), dim3(256), dim3(1024), dim3(1024), dim3(256)};
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_busy, g[shape], b[shape], 0, main_s, EVAL, sink);
        CK(hipStreamSynchronize(main_s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("back to back, grid %4u x %4u threads: %7.1f us per step (%5.1f us of hand-over)\n", g[shape].x, b[shape].x, us, us - 20.0);
    }
    {   // the same 2048 x 256 launches replayed from a graph (one graph of 50 kernel nodes in a chain)
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_busy, grid, block, 0, main_s, EVAL, sink);
        CK(hipStreamEndCapture(main_s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, main_s));
        CK(hipStreamSynchronize(main_s));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(exec, main_s));
        CK(hipStreamSynchronize(main_s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
        printf("hipGraph of 50 chained 2048 x 256 kernels:  %7.1f us per step (%5.1f us of hand-over)\n", us, us - 20.0);
    }
    {   // does hipStreamWaitValue32 on plain device memory really hold the stream back?  A 300 us side kernel publishes the
        // word (from the kernel, then through hipStreamWriteValue32); the main stream waits for it and checks.
        unsigned* viol;
        CK(hipMalloc(&viol, 4));
        for (int how = 0; how < 2; ++how) {
            CK(hipMemset(viol, 0, 4));
            CK(hipMemset(flags, 0, 4096));
            CK(hipDeviceSynchronize());
            for (int i = 0; i < 20; ++i) {
                if (how == 0) hipLaunchKernelGGL(k_side, sgrid, sblock, 0, side[i % 3], 30000LL, sink, flags + i, 7u);
                else {
                    hipLaunchKernelGGL(k_side, sgrid, sblock, 0, side[i % 3], 30000LL, sink, (unsigned*)nullptr, 0u);
                    CK(hipStreamWriteValue32(side[i % 3], flags + i, 7u, 0));
                }
                CK(hipStreamWaitValue32(main_s, flags + i, 7u, hipStreamWaitValueEq, 0xFFFFFFFFu));
                hipLaunchKernelGGL(k_check, dim3(64), dim3(64), 0, main_s, flags + i, 7u, viol);
            }
            CK(hipDeviceSynchronize());
            unsigned v = 0;
            CK(hipMemcpy(&v, viol, 4, hipMemcpyDeviceToHost));
            printf("hipStreamWaitValue32 (Eq) on hipMalloc memory, word written by %s: %u of %u checks ran too early\n", how ? "hipStreamWriteValue32" : "the kernel", v, 20u * 64u);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        run("0 no dependency (side work runs beside)", 0);
        run("1 hipEventRecord on the side stream + hipStreamWaitEvent", 1);
        run("2 stop event on the side kernel's packet + hipStreamWaitEvent", 2);
        run("3 side kernel stores a word + hipStreamWaitValue32 (>=)", 3);
        run("4 side kernel stores a word + the main kernel polls it itself", 4);
        run("5 side kernel stores a word, main stream not ordered (reference)", 5);
        run("6 as 2 + a stop event on the main kernel's packet", 6);
        run("7 as 6 + the side stream waits for the main kernel of 3 steps ago", 7);
    }
    return 0;
}
