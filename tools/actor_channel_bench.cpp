// actor_channel_bench.cpp -- what the actor's channel and reply path cost by themselves (no GPU: the limiter is
// a stand-in that allows everything).  usage: actor_channel_bench [producers] [requests per producer] [window] [group]
// build: g++ -O2 -std=c++17 -pthread -Iinclude tools/actor_channel_bench.cpp -o tools/actor_channel_bench.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <thread>

#include "throttlecrab_actor.hpp"

using namespace throttlecrab;
using namespace throttlecrab::server;

struct NullLimiter {
    static constexpr size_t FLIGHTS = 3;
    std::deque<std::vector<RateLimitOutcome>> flights;
    void submit_batch(const std::vector<Request>& reqs) {
        std::vector<RateLimitOutcome> out;
        out.reserve(reqs.size());
        for (const Request& r : reqs) out.push_back(std::make_pair(true, RateLimitResult{r.max_burst, 1, Duration(0), Duration(0)}));
        flights.push_back(std::move(out));
    }
    std::vector<RateLimitOutcome> collect_batch() {
        auto out = std::move(flights.front());
        flights.pop_front();
        return out;
    }
};

int main(int argc, char** argv) {
    const int producers = argc > 1 ? atoi(argv[1]) : 8;
    const int per = argc > 2 ? atoi(argv[2]) : 500000;
    const int window = argc > 3 ? atoi(argv[3]) : 4096;
    const int group = argc > 4 ? atoi(argv[4]) : 0;
    RateLimiterHandle handle = BasicRateLimiterActor<NullLimiter>::spawn(1 << 20, std::make_shared<NullLimiter>(), 1 << 18);
    const SystemTime t0(std::chrono::nanoseconds(1700000000LL * 1000000000LL));
    const auto a = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int p = 0; p < producers; ++p)
        th.emplace_back([&, p] {
            RateLimiterHandle h = handle;
            uint64_t x = 88172645463325252ULL + p;
            if (group > 0) {
                std::deque<std::future<std::vector<Result<ThrottleResponse>>>> gq;
                for (int i = 0; i < per; i += group) {
                    std::vector<ThrottleRequest> g;
                    for (int j = 0; j < group && i + j < per; ++j) {
                        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                        g.push_back(ThrottleRequest{"user:" + std::to_string(x % 1000000), 100, 1000, 3600, 1, t0});
                    }
                    if ((int)gq.size() * group >= window) gq.front().get(), gq.pop_front();
                    gq.push_back(h.throttle_many_async(std::move(g)));
                }
                while (!gq.empty()) gq.front().get(), gq.pop_front();
                return;
            }
            std::deque<std::future<Result<ThrottleResponse>>> q;
            for (int i = 0; i < per; ++i) {
                x ^= x << 13, x ^= x >> 7, x ^= x << 17;
                if ((int)q.size() >= window) q.front().get(), q.pop_front();
                q.push_back(h.throttle_async(ThrottleRequest{"user:" + std::to_string(x % 1000000), 100, 1000, 3600, 1, t0}));
            }
            while (!q.empty()) q.front().get(), q.pop_front();
        });
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    const auto [b, r, l] = handle.drain_stats();
    std::printf("channel only: %d producers, window %d, group %d: %.2f M requests/s (%llu batches, avg %.0f)\n", producers, window, group,
                producers * (double)per / dt / 1e6, (unsigned long long)b, (double)r / (double)(b ? b : 1));
    return 0;
}
