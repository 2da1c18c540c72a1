// actor_bench.cpp -- requests/s through the batch-draining actor (include/throttlecrab_actor.hpp):
// P producer threads keep W requests outstanding each (throttle_async), string keys "user:<id>".
// The reference's own figures for this layer: 12.5 M req/s for the bare library on an M3 Max, 185 k req/s
// through its fastest transport; its actor answers one request per loop turn.
// build: g++ -O2 -std=c++17 -pthread -Iinclude tools/actor_bench.cpp -Lthrottlecrab_amd -ltcgpu -Wl,-rpath,$PWD/throttlecrab_amd -o tools/actor_bench.bin
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <thread>

#include "throttlecrab_actor.hpp"

using namespace throttlecrab;
using namespace throttlecrab::server;

int main(int argc, char** argv) {
    const int producers = argc > 1 ? atoi(argv[1]) : 16;
    const int per = argc > 2 ? atoi(argv[2]) : 400000;
    const int window = argc > 3 ? atoi(argv[3]) : 4096;
    const int group = argc > 4 ? atoi(argv[4]) : 0; // > 0: throttle_many with this many requests per message
    const uint64_t keys = 1000000;
    const size_t max_batch = 1 << 18;
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(1 << 20, GpuStore(2 * keys, max_batch), max_batch);
    const SystemTime t0(std::chrono::nanoseconds(1700000000LL * 1000000000LL));
    std::atomic<uint64_t> allowed{0}, failed{0};
    auto run = [&](int per_thread) {
        std::vector<std::thread> th;
        for (int p = 0; p < producers; ++p)
            th.emplace_back([&, p] {
                RateLimiterHandle h = handle;
                uint64_t x = 88172645463325252ULL + p;
                std::deque<std::future<Result<ThrottleResponse>>> q;
                uint64_t ok = 0, bad = 0;
                auto reap = [&] {
                    auto r = q.front().get();
                    q.pop_front();
                    if (is_ok(r)) ok += std::get<0>(r).allowed; else ++bad;
                };
                if (group > 0) { // a connection thread handing over whole pipelined buffers
                    std::deque<std::future<std::vector<Result<ThrottleResponse>>>> gq;
                    auto greap = [&] {
                        for (auto& r : gq.front().get()) {
                            if (is_ok(r)) ok += std::get<0>(r).allowed; else ++bad;
                        }
                        gq.pop_front();
                    };
                    for (int i = 0; i < per_thread; i += group) {
                        std::vector<ThrottleRequest> g;
                        g.reserve(group);
                        for (int j = 0; j < group && i + j < per_thread; ++j) {
                            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                            g.push_back(ThrottleRequest{"user:" + std::to_string(x % keys), 100, 1000, 3600, 1, t0 + std::chrono::microseconds(i + j)});
                        }
                        if ((int)gq.size() * group >= window) greap();
                        gq.push_back(h.throttle_many_async(std::move(g)));
                    }
                    while (!gq.empty()) greap();
                    allowed += ok, failed += bad;
                    return;
                }
                for (int i = 0; i < per_thread; ++i) {
                    x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                    if ((int)q.size() >= window) reap();
                    q.push_back(h.throttle_async(ThrottleRequest{"user:" + std::to_string(x % keys), 100, 1000, 3600, 1,
                                                                 t0 + std::chrono::microseconds(i)}));
                }
                while (!q.empty()) reap();
                allowed += ok, failed += bad;
            });
        for (auto& t : th) t.join();
    };
    run(per / 10); // warm-up: key inserts, stream probing, pinned buffers
    const auto [b0, r0, l0] = handle.drain_stats();
    const auto a = std::chrono::steady_clock::now();
    run(per);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    const auto [b1, r1, l1] = handle.drain_stats();
    std::printf("actor: %d producers x %d requests, window %d, group %d: %.2f M requests/s  (%llu batches, avg %.0f, largest %llu; failed %llu)\n",
                producers, per, window, group, producers * (double)per / dt / 1e6, (unsigned long long)(b1 - b0),
                (double)(r1 - r0) / (double)(b1 - b0 ? b1 - b0 : 1), (unsigned long long)l1, (unsigned long long)failed.load());
    const auto [ns_s, ns_a, ns_c, ns_r] = handle.loop_ns();
    std::printf("  actor thread, ns per request (warm-up included): submit %.1f, answer %.1f (of which collect_batch = waiting for the GPU + the outcomes: %.1f), "
                "release %.1f; wall %.1f\n",
                (double)ns_s / (double)r1, (double)ns_a / (double)r1, (double)ns_c / (double)r1, (double)ns_r / (double)r1, 1e9 * dt / (producers * (double)per));
    return 0;
}
