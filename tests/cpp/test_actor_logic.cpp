// The actor's channel and pipelining logic (include/throttlecrab_actor.hpp) over a stand-in limiter: no GPU.
// What is checked is the host logic around the hot path: queue order == evaluation order, replies matched to
// requests, batches formed from what is queued, at most FLIGHTS batches in flight, back-pressure, single and
// throttle_many messages interleaved, errors from the limiter, shutdown with work in flight.
// build: g++ -std=c++17 -pthread -Iinclude tests/cpp/test_actor_logic.cpp -o tests/cpp/test_actor_logic
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <thread>

#include "throttlecrab_actor.hpp"

using namespace throttlecrab;
using namespace throttlecrab::server;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

// A limiter with the interface the actor needs.  "allowed" while the key has seen fewer than max_burst
// requests; `remaining` carries the global sequence number of the evaluation, so the test can see the order.
struct FakeLimiter {
    static constexpr size_t FLIGHTS = 3;
    std::map<std::string, int64_t> seen;
    std::deque<std::vector<RateLimitOutcome>> flights;
    int64_t seq = 0;
    size_t max_in_flight = 0, submits = 0, largest = 0;
    std::chrono::microseconds collect_delay{0};
    int fail_submit_every = 0; // > 0: every n-th submit throws
    void submit_batch(const std::vector<Request>& reqs) {
        ++submits;
        if (fail_submit_every && submits % (size_t)fail_submit_every == 0) throw std::runtime_error("injected failure");
        CHECK(flights.size() < FLIGHTS);
        std::vector<RateLimitOutcome> out;
        for (const Request& r : reqs) {
            if (r.quantity < 0) {
                out.push_back(CellError{CellError::NegativeQuantity, r.quantity, {}});
                continue;
            }
            int64_t& c = seen[std::string(r.key)];
            const bool ok = c < r.max_burst;
            if (ok) ++c;
            out.push_back(std::make_pair(ok, RateLimitResult{r.max_burst, seq++, Duration(0), Duration(0)}));
        }
        if (reqs.size() > largest) largest = reqs.size();
        flights.push_back(std::move(out));
        if (flights.size() > max_in_flight) max_in_flight = flights.size();
    }
    std::vector<RateLimitOutcome> collect_batch() {
        CHECK(!flights.empty());
        if (collect_delay.count()) std::this_thread::sleep_for(collect_delay); // "the GPU is busy"
        auto out = std::move(flights.front());
        flights.pop_front();
        return out;
    }
};
using FakeActor = BasicRateLimiterActor<FakeLimiter>;
static SystemTime now0() { return SystemTime(std::chrono::nanoseconds(1700000000LL * 1000000000LL)); }

static void test_order_and_matching() {
    auto lim = std::make_shared<FakeLimiter>();
    lim->collect_delay = std::chrono::microseconds(200);
    RateLimiterHandle h = FakeActor::spawn(1 << 16, lim, 4096);
    // one sender: its requests are evaluated in the order it sent them, whatever the batching
    std::vector<std::future<Result<ThrottleResponse>>> fs;
    for (int i = 0; i < 20000; ++i) fs.push_back(h.throttle_async(ThrottleRequest{"k" + std::to_string(i % 50), 1 << 30, 1, 1, 1, now0()}));
    int64_t last = -1;
    for (auto& f : fs) {
        auto r = f.get();
        CHECK(is_ok(r));
        CHECK(std::get<0>(r).remaining == last + 1); // the limiter's sequence number
        last = std::get<0>(r).remaining;
    }
    auto [batches, requests, largest] = h.drain_stats();
    CHECK(requests == 20000 && batches < 20000 && largest <= 4096);
    // a group message keeps its order too and is answered as a whole; errors come back per request
    std::vector<ThrottleRequest> g;
    for (int i = 0; i < 1000; ++i) g.push_back(ThrottleRequest{"g", 1 << 30, 1, 1, i % 100 == 7 ? -1 : 1, now0()});
    auto rs = h.throttle_many(std::move(g));
    CHECK(rs.size() == 1000);
    for (int i = 0; i < 1000; ++i) {
        if (i % 100 == 7) {
            CHECK(!is_ok(rs[i]) && std::get<1>(rs[i]) == "Rate limit check failed: negative quantity: -1");
        } else {
            CHECK(is_ok(rs[i]) && std::get<0>(rs[i]).remaining > last);
            last = std::get<0>(rs[i]).remaining;
        }
    }
}

static void test_pipeline_depth_and_backpressure() {
    auto lim = std::make_shared<FakeLimiter>();
    lim->collect_delay = std::chrono::microseconds(300);
    RateLimiterHandle h = FakeActor::spawn(256, lim, 64); // small buffer: senders must wait, nothing may be lost
    std::atomic<int> allowed{0}, answered{0};
    std::vector<std::thread> th;
    for (int p = 0; p < 8; ++p)
        th.emplace_back([&, p] {
            RateLimiterHandle mine = h;
            for (int i = 0; i < 2000; ++i) {
                if (i % 10 == 0) {
                    std::vector<ThrottleRequest> g(17, ThrottleRequest{"shared" + std::to_string(i % 4), 100, 1, 1, 1, now0()});
                    for (auto& r : mine.throttle_many(std::move(g))) {
                        CHECK(is_ok(r));
                        allowed += std::get<0>(r).allowed;
                        ++answered;
                    }
                } else {
                    auto r = mine.throttle(ThrottleRequest{"shared" + std::to_string((i + p) % 4), 100, 1, 1, 1, now0()});
                    CHECK(is_ok(r));
                    allowed += std::get<0>(r).allowed;
                    ++answered;
                }
            }
        });
    for (auto& t : th) t.join();
    CHECK(answered.load() == 8 * (1800 + 200 * 17));
    CHECK(allowed.load() == 4 * 100);                       // every key granted exactly its burst
    CHECK(lim->max_in_flight <= FakeLimiter::FLIGHTS && lim->max_in_flight >= 2); // the loop really overlaps batches
    CHECK(lim->largest <= 64 + 16);                         // max_batch by requests (a group may overshoot by its own size - 1)
}

static void test_limiter_errors_and_shutdown() {
    auto lim = std::make_shared<FakeLimiter>();
    lim->fail_submit_every = 3;
    std::vector<std::future<Result<ThrottleResponse>>> fs;
    {
        RateLimiterHandle h = FakeActor::spawn(1 << 12, lim, 8);
        for (int i = 0; i < 500; ++i) fs.push_back(h.throttle_async(ThrottleRequest{"e", 1 << 30, 1, 1, 1, now0()}));
        // the handle dies here with requests queued and batches in flight: all of them are still answered
    }
    int ok = 0, failed = 0;
    for (auto& f : fs) {
        auto r = f.get();
        if (is_ok(r)) ++ok;
        else {
            CHECK(std::get<1>(r) == "Rate limit check failed: internal error: injected failure");
            ++failed;
        }
    }
    CHECK(ok + failed == 500 && failed > 0 && ok > 0);
    RateLimiterHandle dead;
    auto r = dead.throttle(ThrottleRequest{"x", 1, 1, 1, 1, now0()});
    CHECK(!is_ok(r) && std::get<1>(r) == "Rate limiter actor has shut down");
    auto many = dead.throttle_many(std::vector<ThrottleRequest>(3, ThrottleRequest{"x", 1, 1, 1, 1, now0()}));
    CHECK(many.size() == 3 && !is_ok(many[0]));
}

static void test_linger_collects_a_batch() {
    auto lim = std::make_shared<FakeLimiter>();
    RateLimiterHandle h = FakeActor::spawn(1 << 12, lim, 1024, std::chrono::microseconds(20000), 64);
    std::vector<std::future<Result<ThrottleResponse>>> fs;
    for (int i = 0; i < 64; ++i) {
        fs.push_back(h.throttle_async(ThrottleRequest{"l", 1 << 30, 1, 1, 1, now0()}));
        if (i % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
    for (auto& f : fs) CHECK(is_ok(f.get()));
    auto [batches, requests, largest] = h.drain_stats();
    CHECK(requests == 64 && batches <= 3 && largest >= 32); // the loop waited for the queue to fill instead of taking 1, 1, 1 ...
}

// Round 5: the channel is sharded by handle.  40 handles (more than there are shards: some share one) send numbered requests for a key
// of their own from their own threads, singly and in groups, through a small buffer: every handle's requests must be evaluated in the
// order it sent them (the limiter's sequence numbers grow along each handle's stream), nothing lost, nothing answered twice.
static void test_per_handle_order_across_shards() {
    auto lim = std::make_shared<FakeLimiter>();
    lim->collect_delay = std::chrono::microseconds(50);
    RateLimiterHandle h = FakeActor::spawn(512, lim, 256);
    std::atomic<int> answered{0};
    std::vector<std::thread> th;
    for (int p = 0; p < 40; ++p)
        th.emplace_back([&, p] {
            RateLimiterHandle mine = h; // (a copy: its own shard)
            const std::string key = "h" + std::to_string(p);
            int64_t last = -1;
            std::deque<std::future<Result<ThrottleResponse>>> q;
            auto reap = [&] {
                auto r = q.front().get();
                q.pop_front();
                CHECK(is_ok(r) && std::get<0>(r).remaining > last);
                last = std::get<0>(r).remaining;
                ++answered;
            };
            for (int i = 0; i < 600; ++i) {
                if (i % 50 == 49) { // a group in between: it must fall into place behind what was sent before it
                    while (!q.empty()) reap();
                    std::vector<ThrottleRequest> g(7, ThrottleRequest{key, 1 << 30, 1, 1, 1, now0()});
                    for (auto& r : mine.throttle_many(std::move(g))) {
                        CHECK(is_ok(r) && std::get<0>(r).remaining > last);
                        last = std::get<0>(r).remaining;
                        ++answered;
                    }
                    continue;
                }
                if (q.size() >= 32) reap();
                q.push_back(mine.throttle_async(ThrottleRequest{key, 1 << 30, 1, 1, 1, now0()}));
            }
            while (!q.empty()) reap();
        });
    for (auto& t : th) t.join();
    CHECK(answered.load() == 40 * (588 + 12 * 7));
    auto [batches, requests, largest] = h.drain_stats();
    CHECK(requests == (uint64_t)answered.load() && batches > 0 && largest <= 256 + 6);
    CHECK(lim->max_in_flight <= FakeLimiter::FLIGHTS);
}

int main() {
    test_order_and_matching();
    test_pipeline_depth_and_backpressure();
    test_limiter_errors_and_shutdown();
    test_linger_collects_a_batch();
    test_per_handle_order_across_shards();
    std::puts("all tests passed");
    return 0;
}
