// C++ host-mirror tests written after the reference's own tests
// (throttlecrab/src/core/tests.rs, store/store_test_suite.rs): same calls, same asserts.
// build: g++ -std=c++17 -Iinclude tests/cpp/test_host_mirror.cpp -Lthrottlecrab_amd -ltcgpu
#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "throttlecrab_gpu.hpp"

using namespace throttlecrab;
using std::chrono::milliseconds;
using std::chrono::seconds;

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                     \
        }                                                                     \
    } while (0)

static std::pair<bool, RateLimitResult> unwrap(const RateLimitOutcome& o) {
    CHECK(is_ok(o));
    return std::get<0>(o);
}

static SystemTime now0() { return SystemTime(std::chrono::nanoseconds(1700000000LL * 1000000000LL)); }

// core/tests.rs:17-33
static void test_burst_capacity() {
    RateLimiter limiter{GpuStore()};
    auto now = now0();
    for (int i = 0; i < 5; ++i) {
        auto [allowed, result] = unwrap(limiter.rate_limit("burst_test", 5, 10, 60, 1, now));
        CHECK(allowed);
        CHECK(result.remaining == 5 - (i + 1));
    }
    auto [allowed, result] = unwrap(limiter.rate_limit("burst_test", 5, 10, 60, 1, now));
    CHECK(!allowed);
    CHECK(result.remaining == 0);
    CHECK(std::chrono::duration_cast<seconds>(result.retry_after).count() > 0);
}

// core/tests.rs:36-62
static void test_rate_replenishment() {
    RateLimiter limiter{GpuStore()};
    auto now = now0();
    CHECK(unwrap(limiter.rate_limit("replenish_test", 2, 60, 60, 1, now)).first);
    CHECK(unwrap(limiter.rate_limit("replenish_test", 2, 60, 60, 1, now)).first);
    CHECK(!unwrap(limiter.rate_limit("replenish_test", 2, 60, 60, 1, now)).first);
    CHECK(unwrap(limiter.rate_limit("replenish_test", 2, 60, 60, 1, now + seconds(1))).first);
}

// core/tests.rs:94-118
static void test_quantity_parameter() {
    RateLimiter limiter{GpuStore()};
    auto now = now0();
    auto r1 = unwrap(limiter.rate_limit("quantity_test", 10, 10, 60, 5, now));
    CHECK(r1.first && r1.second.remaining == 5);
    auto r2 = unwrap(limiter.rate_limit("quantity_test", 10, 10, 60, 6, now));
    CHECK(!r2.first && r2.second.remaining == 5);
    auto r3 = unwrap(limiter.rate_limit("quantity_test", 10, 10, 60, 5, now));
    CHECK(r3.first && r3.second.remaining == 0);
}

// core/tests.rs:121-145
static void test_errors() {
    RateLimiter limiter{GpuStore()};
    auto now = now0();
    auto e = limiter.rate_limit("negative_test", 10, 10, 60, -1, now);
    CHECK(!is_ok(e) && std::get<1>(e).kind == CellError::NegativeQuantity && std::get<1>(e).quantity == -1);
    CHECK(std::get<1>(e).to_string() == "negative quantity: -1");
    CHECK(!is_ok(limiter.rate_limit("test", 0, 10, 60, 1, now)));
    CHECK(!is_ok(limiter.rate_limit("test", 10, 0, 60, 1, now)));
    auto e3 = limiter.rate_limit("test", 10, 10, 0, 1, now);
    CHECK(!is_ok(e3) && std::get<1>(e3).kind == CellError::InvalidRateLimit);
}

// store/store_test_suite.rs:113-170
static void test_store_ttl_expiration() {
    GpuStore store(100);
    auto now = now0();
    auto ttl = seconds(60);
    CHECK(store.set_if_not_exists_with_ttl("key1", 100, ttl, now));
    CHECK(store.get("key1", now) == std::optional<int64_t>(100));
    CHECK(store.get("key1", now + seconds(59)) == std::optional<int64_t>(100));
    CHECK(store.get("key1", now + seconds(61)) == std::nullopt);
    CHECK(!store.compare_and_swap_with_ttl("key1", 100, 200, ttl, now + seconds(61)));
    CHECK(store.set_if_not_exists_with_ttl("key1", 300, ttl, now + seconds(61)));
    CHECK(store.get("key1", now + seconds(61)) == std::optional<int64_t>(300));
    CHECK(store.cleanup(now + seconds(200)) == 1);
}

// actor_tests.rs:34-70 as one batch: 20 identical requests, exactly the burst is allowed,
// and it is the FIRST ten in queue order
static void test_rate_limit_batch_order() {
    RateLimiter limiter{GpuStore(1000, 1024)};
    auto now = now0();
    std::vector<Request> reqs(20, Request{"concurrent_test", 10, 10, 60, 1, now});
    reqs.push_back(Request{"other", 5, 10, 60, -3, now});
    auto out = limiter.rate_limit_batch(reqs);
    CHECK(out.size() == 21);
    for (int i = 0; i < 20; ++i) {
        auto [allowed, result] = unwrap(out[i]);
        CHECK(allowed == (i < 10));
        CHECK(result.limit == 10);
        CHECK(result.remaining == (i < 10 ? 9 - i : 0));
    }
    CHECK(!is_ok(out[20]) && std::get<1>(out[20]).kind == CellError::NegativeQuantity);
}

int main() {
    test_burst_capacity();
    test_rate_replenishment();
    test_quantity_parameter();
    test_errors();
    test_store_ttl_expiration();
    test_rate_limit_batch_order();
    std::puts("host mirror: all tests passed");
    return 0;
}
