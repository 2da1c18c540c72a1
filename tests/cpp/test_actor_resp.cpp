// Batch-draining actor + RESP pipeline against libtcgpu.so on the GPU.  Written after
// throttlecrab-server/src/actor_tests.rs:9-70 and transport/redis_test.rs (reply arrays).
// build: g++ -std=c++17 -pthread -Iinclude tests/cpp/test_actor_resp.cpp -Lthrottlecrab_amd -ltcgpu
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>

#include "throttlecrab_actor.hpp"
#include "throttlecrab_metrics.hpp"
#include "throttlecrab_resp.hpp"

using namespace throttlecrab;
using namespace throttlecrab::server;
using std::chrono::seconds;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static SystemTime now0() { return SystemTime(std::chrono::nanoseconds(1700000000LL * 1000000000LL)); }

// actor_tests.rs:9-31
static void test_basic_rate_limiting() {
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(100, GpuStore(1000));
    ThrottleRequest req{"test", 5, 10, 60, 1, now0()};
    auto resp = handle.throttle(req);
    CHECK(is_ok(resp));
    const ThrottleResponse& r = std::get<0>(resp);
    CHECK(r.allowed && r.limit == 5 && r.remaining == 4);
}

// actor_tests.rs:34-70: 20 concurrent identical requests, burst 10 -> exactly 10 allowed
static void test_concurrent_requests() {
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(100, GpuStore(1000));
    ThrottleRequest req{"concurrent_test", 10, 10, 60, 1, now0()};
    std::atomic<int> allowed{0};
    std::vector<std::thread> ts;
    for (int i = 0; i < 20; ++i)
        ts.emplace_back([&, h = handle] () mutable {
            auto r = h.throttle(req);
            CHECK(is_ok(r));
            if (std::get<0>(r).allowed) allowed++;
        });
    for (auto& t : ts) t.join();
    CHECK(allowed.load() == 10);
}

// errors reach the caller as "Rate limit check failed: <CellError>" (actor.rs:252)
static void test_errors_and_truncation() {
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(16, GpuStore(1000));
    auto neg = handle.throttle(ThrottleRequest{"e", 5, 10, 60, -1, now0()});
    CHECK(!is_ok(neg) && std::get<1>(neg) == "Rate limit check failed: negative quantity: -1");
    auto inv = handle.throttle(ThrottleRequest{"e", 0, 10, 60, 1, now0()});
    CHECK(!is_ok(inv) && std::get<1>(inv) == "Rate limit check failed: invalid rate limit parameters");
    // (10,100,60): reset_after 5.4 s -> 5, q=5: 7.8 s -> 7 (redis_test.rs:117-144)
    auto a = handle.throttle(ThrottleRequest{"secs", 10, 100, 60, 1, now0()});
    CHECK(is_ok(a) && std::get<0>(a).reset_after == 5 && std::get<0>(a).retry_after == 0 && std::get<0>(a).remaining == 9);
}

// many producers, one queue: every request answered, per key exactly `burst` allowed, and the
// queue really is drained in batches
static void test_many_producers_are_batched() {
    const int producers = 16, per = 4000, keys = 50, burst = 30;
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(4096, GpuStore(10000, 1 << 16), 1 << 16);
    std::vector<std::atomic<int>> allowed(keys);
    for (auto& a : allowed) a = 0;
    std::atomic<int> answered{0};
    std::vector<std::thread> ts;
    for (int p = 0; p < producers; ++p)
        ts.emplace_back([&, p, h = handle]() mutable {
            std::vector<std::pair<int, std::future<Result<ThrottleResponse>>>> inflight;
            for (int i = 0; i < per; ++i) {
                const int k = (p * 7 + i) % keys;
                inflight.emplace_back(k, h.throttle_async(ThrottleRequest{"k" + std::to_string(k), burst, 1, 3600, 1, now0()}));
                if (inflight.size() == 256 || i + 1 == per) {
                    for (auto& f : inflight) {
                        auto r = f.second.get();
                        CHECK(is_ok(r));
                        if (std::get<0>(r).allowed) allowed[f.first]++;
                        answered++;
                    }
                    inflight.clear();
                }
            }
        });
    for (auto& t : ts) t.join();
    CHECK(answered.load() == producers * per);
    for (int k = 0; k < keys; ++k) CHECK(allowed[k].load() == burst);
    auto [batches, requests, largest] = handle.drain_stats();
    CHECK(requests == (uint64_t)producers * per);
    CHECK(batches < requests / 4 && largest > 16); // the reference's loop would have made `requests` turns
    std::printf("actor: %llu requests in %llu batches (largest %llu)\n", (unsigned long long)requests,
                (unsigned long long)batches, (unsigned long long)largest);
}

static void test_shutdown() {
    RateLimiterHandle moved;
    {
        RateLimiterHandle h = RateLimiterActor::spawn_gpu(8, GpuStore(100));
        CHECK(is_ok(h.throttle(ThrottleRequest{"x", 1, 1, 1, 1, now0()})));
        moved = h; // a copy keeps the actor alive
    }
    CHECK(is_ok(moved.throttle(ThrottleRequest{"x", 1, 1, 1, 1, now0() + seconds(5)})));
    RateLimiterHandle none;
    auto r = none.throttle(ThrottleRequest{"x", 1, 1, 1, 1, now0()});
    CHECK(!is_ok(r) && std::get<1>(r) == "Rate limiter actor has shut down");
}

static std::string cmd(std::initializer_list<std::string> args) {
    std::string s = "*" + std::to_string(args.size()) + "\r\n";
    for (const std::string& a : args) s += "$" + std::to_string(a.size()) + "\r\n" + a + "\r\n";
    return s;
}

// redis_test.rs:117-144, 272-304, 384-395, 492-502, 678-717 through one pipelined buffer
static void test_resp_pipeline() {
    GpuStore store(1000, 4096);
    const int64_t t0 = 1700000000LL * 1000000000LL;
    std::string wire = cmd({"THROTTLE", "a", "10", "100", "60"}) + cmd({"THROTTLE", "b", "10", "100", "60", "5"}) + cmd({"PING"});
    for (int i = 0; i < 4; ++i) wire += cmd({"THROTTLE", "c", "3", "100", "60"});
    wire += cmd({"THROTTLE", "d", "10", "100", "60", "15"}) + cmd({"THROTTLE", "d", "10", "100", "60", "0"}) +
            cmd({"THROTTLE", "e", "10", "100", "60", "-1"}) + cmd({"THROTTLE", "e", "0", "100", "60"}) +
            cmd({"THROTTLE", "m", "9223372036854775807", "9223372036854775807", "9223372036854775807"}) +
            cmd({"THROTTLE", "one", "1", "1", "1"}) + cmd({"NOPE"}) + cmd({"QUIT"}) + cmd({"PING"});
    resp::Pipeline p;
    const size_t consumed = p.parse((const uint8_t*)wire.data(), wire.size(), [&] { return t0; });
    CHECK(p.protocol_error.empty() && p.quit && consumed == wire.size() - cmd({"PING"}).size());
    std::string out;
    CHECK(p.run(store.handle(), out) == TC_E_OK);
    // expected numbers: the CPU oracle on the same sequence (and redis_test.rs:117-144 for a / b)
    const std::string expect =
        "*5\r\n:1\r\n:10\r\n:9\r\n:5\r\n:0\r\n"     // [1,10,9,5,0]
        "*5\r\n:1\r\n:10\r\n:5\r\n:7\r\n:0\r\n"     // q=5: [1,10,5,7,0]
        "+PONG\r\n"
        "*5\r\n:1\r\n:3\r\n:2\r\n:1\r\n:0\r\n"      // (3,100,60): remaining 2,1,0 then denied
        "*5\r\n:1\r\n:3\r\n:1\r\n:1\r\n:0\r\n"
        "*5\r\n:1\r\n:3\r\n:0\r\n:2\r\n:0\r\n"
        "*5\r\n:0\r\n:3\r\n:0\r\n:2\r\n:0\r\n"
        "*5\r\n:0\r\n:10\r\n:10\r\n:4\r\n:3\r\n"    // q=15 on burst 10: denied, remaining 10
        "*5\r\n:1\r\n:10\r\n:10\r\n:4\r\n:0\r\n"    // q=0: allowed, remaining 10
        "-ERR Rate limit check failed: negative quantity: -1\r\n"
        "-ERR Rate limit check failed: invalid rate limit parameters\r\n"
        "*5\r\n:1\r\n:9223372036854775807\r\n:4294967294\r\n:4294967294\r\n:0\r\n" // redis_test.rs:678-699
        "*5\r\n:1\r\n:1\r\n:0\r\n:0\r\n:0\r\n"      // (1,1,1): redis_test.rs:702-717
        "-ERR unknown command 'NOPE'\r\n"
        "+OK\r\n";
    if (out != expect) std::fprintf(stderr, "got:\n%s\nwant:\n%s\n", out.c_str(), expect.c_str());
    CHECK(out == expect);
    // the same buffer through the actor (one throttle_many message) against a fresh store: same bytes
    RateLimiterHandle handle = RateLimiterActor::spawn_gpu(1000, GpuStore(1000, 4096), 4096);
    std::string via;
    p.run_via_actor(handle, via);
    if (via != expect) std::fprintf(stderr, "via actor got:\n%s\nwant:\n%s\n", via.c_str(), expect.c_str());
    CHECK(via == expect);
}

// tests/metrics_test.rs + denied_keys_test.rs:37-67 with the decisions counted on the device
static void test_metrics_from_engine() {
    GpuStore store(1000, 4096, 0, /*track_denied=*/true);
    const int64_t t0 = 1700000000LL * 1000000000LL;
    std::string wire;
    for (int i = 0; i < 12; ++i) wire += cmd({"THROTTLE", "top_key", "2", "10", "60"});      // 2 allowed, 10 denied
    for (int i = 0; i < 7; ++i) wire += cmd({"THROTTLE", "medium_key", "2", "10", "60"});    // 5 denied
    for (int i = 0; i < 3; ++i) wire += cmd({"THROTTLE", "low_key", "2", "10", "60"});       // 1 denied
    wire += cmd({"THROTTLE", "user:789", "2", "10", "60"}) + cmd({"THROTTLE", "bad", "0", "10", "60"});  // allowed; error
    resp::Pipeline p;
    p.parse((const uint8_t*)wire.data(), wire.size(), [&] { return t0; });
    std::string out;
    CHECK(p.run(store.handle(), out) == TC_E_OK);
    Metrics m;
    m.record_transport(Transport::Redis, p.commands());
    CHECK(m.snapshot_from_engine(store.handle()) == TC_E_OK);
    const std::string prom = m.export_prometheus();
    CHECK(prom.find("throttlecrab_requests_total 24\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_requests_allowed 7\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_requests_denied 16\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_requests_errors 1\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_requests_by_transport{transport=\"redis\"} 24\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_top_denied_keys{key=\"top_key\",rank=\"1\"} 10\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_top_denied_keys{key=\"medium_key\",rank=\"2\"} 5\n") != std::string::npos);
    CHECK(prom.find("throttlecrab_top_denied_keys{key=\"low_key\",rank=\"3\"} 1\n") != std::string::npos);
    CHECK(prom.find("user:789") == std::string::npos); // never denied
}

// RateLimiter::submit_batch / collect_batch (TC_B_ASYNC behind them) == rate_limit_batch on a second limiter:
// same stream of batches (new keys, hot keys, long keys, errors, tiny batches answered by single calls),
// up to FLIGHTS batches in flight.
static void test_pipelined_submit_collect() {
    RateLimiter seq(GpuStore(20000, 4096)), pip(GpuStore(20000, 4096));
    uint64_t x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13, x ^= x >> 7, x ^= x << 17; return x; };
    std::vector<std::vector<std::string>> keys;     // the requests' keys must outlive the submission
    std::vector<std::vector<Request>> batches;
    const size_t sizes[] = {700, 3, 4096, 1, 2500, 5, 64, 4000, 2, 900, 3100, 4};
    for (size_t bi = 0; bi < sizeof sizes / sizeof sizes[0]; ++bi) {
        keys.emplace_back();
        for (size_t i = 0; i < sizes[bi]; ++i) {
            const uint64_t r = rnd();
            std::string k = (r % 7 == 0) ? "hot" : "user:" + std::to_string(r % (500 + 300 * bi));
            if (r % 97 == 0) k = std::string(70, 'L') + std::to_string(r % 5); // beyond the inline 48 bytes
            keys.back().push_back(std::move(k));
        }
        batches.emplace_back();
        for (size_t i = 0; i < sizes[bi]; ++i) {
            const uint64_t r = rnd();
            batches.back().push_back(Request{keys.back()[i], 5, 10, 60, r % 53 == 0 ? -1 : (int64_t)(r % 3),
                                             now0() + std::chrono::milliseconds(100 * bi) + std::chrono::microseconds(r % 50000)});
        }
    }
    std::vector<std::vector<RateLimitOutcome>> want, got;
    for (const auto& b : batches) want.push_back(seq.rate_limit_batch(b));
    for (const auto& b : batches) {
        if (pip.in_flight() == RateLimiter::FLIGHTS) got.push_back(pip.collect_batch());
        pip.submit_batch(b);
    }
    while (pip.in_flight()) got.push_back(pip.collect_batch());
    CHECK(got.size() == want.size());
    for (size_t bi = 0; bi < want.size(); ++bi) {
        CHECK(got[bi].size() == want[bi].size());
        for (size_t i = 0; i < want[bi].size(); ++i) {
            CHECK(got[bi][i].index() == want[bi][i].index());
            if (is_ok(want[bi][i])) {
                const auto &a = std::get<0>(got[bi][i]), &b = std::get<0>(want[bi][i]);
                CHECK(a.first == b.first && a.second.limit == b.second.limit && a.second.remaining == b.second.remaining &&
                      a.second.reset_after == b.second.reset_after && a.second.retry_after == b.second.retry_after);
            } else {
                CHECK(std::get<1>(got[bi][i]).to_string() == std::get<1>(want[bi][i]).to_string());
            }
        }
    }
}

// a connection's pipelined buffer as one message: evaluated in order between the other senders' requests,
// answered together; same replies as the requests sent one by one
static void test_throttle_many() {
    RateLimiterHandle one = RateLimiterActor::spawn_gpu(1000, GpuStore(1000));
    RateLimiterHandle many = RateLimiterActor::spawn_gpu(1000, GpuStore(1000));
    std::vector<ThrottleRequest> reqs;
    for (int i = 0; i < 300; ++i)
        reqs.push_back(ThrottleRequest{"conn:" + std::to_string(i % 7), 5, 10, 60, i % 31 == 0 ? -1 : 1, now0() + std::chrono::milliseconds(i)});
    auto got = many.throttle_many(reqs);
    CHECK(got.size() == reqs.size());
    for (size_t i = 0; i < reqs.size(); ++i) {
        auto want = one.throttle(reqs[i]);
        CHECK(want.index() == got[i].index());
        if (is_ok(want)) {
            const ThrottleResponse &a = std::get<0>(want), &b = std::get<0>(got[i]);
            CHECK(a.allowed == b.allowed && a.limit == b.limit && a.remaining == b.remaining && a.reset_after == b.reset_after &&
                  a.retry_after == b.retry_after);
        } else {
            CHECK(std::get<1>(want) == std::get<1>(got[i]));
        }
    }
    CHECK(many.throttle_many({}).empty());
    // groups from several threads, mixed with single requests: every key still gets exactly `burst` grants
    std::atomic<int> allowed{0};
    std::vector<std::thread> th;
    for (int p = 0; p < 8; ++p)
        th.emplace_back([&, p] {
            RateLimiterHandle h = many;
            for (int round = 0; round < 20; ++round) {
                std::vector<ThrottleRequest> g;
                for (int i = 0; i < 50; ++i) g.push_back(ThrottleRequest{"shared:" + std::to_string(i % 10), 20, 10, 3600, 1, now0()});
                for (auto& r : h.throttle_many(std::move(g))) {
                    CHECK(is_ok(r));
                    allowed += std::get<0>(r).allowed;
                }
                auto r = h.throttle(ThrottleRequest{"shared:" + std::to_string(p % 10), 20, 10, 3600, 1, now0()});
                CHECK(is_ok(r));
                allowed += std::get<0>(r).allowed;
            }
        });
    for (auto& t : th) t.join();
    CHECK(allowed.load() == 10 * 20);
}

// the drop-in cleans itself.  A store of 4 096 slots behind the actor, 3 x that many distinct keys whose
// entries live 0.1 s ((2, 10, 1 s)), stream time advancing one second per group: the reference's AdaptiveStore cleans inside
// its own set_if_not_exists / compare_and_swap (adaptive_cleanup.rs:205-211,229,262) and never holds more than a few groups;
// here the engine's policy (GpuStore's default: CleanupPolicy::adaptive) has to keep the fixed table from filling up.
// Not one reply may be an internal error; groups go through the pipelined TC_B_ASYNC path (2 000 > 1 024 requests), the
// single requests in between through the one-launch path and tc_rate_limit.
static void test_store_cleans_itself_behind_the_actor() {
    const size_t capacity = 4096, group = 2000, groups = 7; // 14 000 distinct keys = 3.4 x the table
    GpuStore store(capacity, 1 << 12, 0, false, CleanupPolicy::adaptive().created_at(now0()));
    auto limiter = std::make_shared<RateLimiter>(std::move(store));
    RateLimiterHandle h = BasicRateLimiterActor<RateLimiter>::spawn(1 << 14, limiter, 1 << 12);
    size_t serial = 0, allowed = 0;
    for (size_t g = 0; g < groups; ++g) {
        const SystemTime t = now0() + std::chrono::seconds(g);
        std::vector<ThrottleRequest> reqs;
        for (size_t i = 0; i < group; ++i) reqs.push_back(ThrottleRequest{"short:" + std::to_string(serial++), 2, 10, 1, 1, t + std::chrono::microseconds(i)});
        for (auto& r : h.throttle_many(std::move(reqs))) {
            CHECK(is_ok(r)); // "Rate limit check failed: internal error: ..." is what a full table would answer
            allowed += std::get<0>(r).allowed;
        }
        for (int i = 0; i < 5; ++i) { // lightly loaded moments: single requests
            auto r = h.throttle(ThrottleRequest{"short:" + std::to_string(serial++), 2, 10, 1, 1, t + std::chrono::milliseconds(500)});
            CHECK(is_ok(r) && std::get<0>(r).allowed);
        }
    }
    CHECK(allowed == groups * group); // every key was new
    h = RateLimiterHandle();          // the actor exits; the limiter is ours again
    const tc_sweep_info st = limiter->store().cleanup_stats();
    CHECK(st.kind == TC_SWEEP_ADAPTIVE && st.sweeps >= 2 && st.retries == 0);
    CHECK(limiter->store().len() <= capacity);
    uint64_t c[TC_CNT_COUNT];
    CHECK(tc_counters(limiter->store().handle(), c) == TC_E_OK && c[TC_CNT_ERRORS] == 0 && c[TC_CNT_ALLOWED] == groups * (group + 5));
    // a store WITHOUT a policy runs full on the same stream: that is what the policy is for
    RateLimiter bare(GpuStore(capacity, 1 << 12, 0, false, CleanupPolicy::none()));
    size_t internal = 0;
    serial = 0;
    for (size_t g = 0; g < 3; ++g) {
        std::vector<Request> reqs;
        std::vector<std::string> keys;
        for (size_t i = 0; i < group; ++i) keys.push_back("short:" + std::to_string(serial++));
        for (size_t i = 0; i < group; ++i) reqs.push_back(Request{keys[i], 2, 10, 1, 1, now0() + std::chrono::seconds(g)});
        for (auto& o : bare.rate_limit_batch(reqs)) internal += !throttlecrab::is_ok(o);
    }
    CHECK(internal == 3 * group - capacity);
}

int main() {
    test_store_cleans_itself_behind_the_actor();
    test_pipelined_submit_collect();
    test_throttle_many();
    test_basic_rate_limiting();
    test_concurrent_requests();
    test_errors_and_truncation();
    test_many_producers_are_batched();
    test_shutdown();
    test_resp_pipeline();
    test_metrics_from_engine();
    std::puts("all tests passed");
    return 0;
}
