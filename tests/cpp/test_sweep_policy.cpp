// Sweep schedulers (include/throttlecrab_sweep.hpp) against the reference's cadence rules
// (adaptive_cleanup.rs:138-211, periodic.rs:128-142, probabilistic.rs:110-125).  No GPU.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>

#include "throttlecrab_sweep.hpp"

using namespace throttlecrab::sweep;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

int main() {
    const int64_t T0 = 1700000000LL * NS;
    { // periodic: due exactly every interval after the previous cleanup
        PeriodicSweep p(T0, 60 * NS);
        CHECK(!p.on_batch(1000000, T0 + 59 * NS) && p.on_batch(1, T0 + 60 * NS));
        p.swept(10, 100, T0 + 61 * NS);
        CHECK(p.next_cleanup_ns() == T0 + 121 * NS && !p.on_batch(1, T0 + 120 * NS) && p.on_batch(1, T0 + 121 * NS));
    }
    { // probabilistic: the batch rule == the reference's per-operation rule
        for (uint64_t n : {1000ULL, 7ULL, 64ULL, 2654435761ULL, 12ULL}) {
            ProbabilisticSweep s(n);
            uint64_t ops = 0;
            uint64_t seed = 88172645463325252ULL;
            for (int b = 0; b < 400; ++b) {
                seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17;
                const uint64_t batch = 1 + seed % 5000;
                bool ref = false;
                for (uint64_t k = ops + 1; k <= ops + batch; ++k) ref |= (k * 2654435761ULL) % n == 0; // probabilistic.rs:116-117
                ops += batch;
                CHECK(s.on_batch(batch, T0) == ref);
            }
            CHECK(s.operations() == ops);
        }
    }
    { // adaptive: triggers and interval adaptation
        AdaptiveSweep a(T0, 1000);
        CHECK(a.current_interval_ns() == 5 * NS && a.next_cleanup_ns() == T0 + 5 * NS);
        CHECK(!a.on_batch(10, T0 + 1 * NS, 10));
        CHECK(a.on_batch(10, T0 + 5 * NS, 10));                          // time (:140)
        a.swept(0, 20, T0 + 5 * NS);                                     // nothing expired: interval doubles (:187-190)
        CHECK(a.current_interval_ns() == 10 * NS && a.next_cleanup_ns() == T0 + 15 * NS);
        CHECK(!a.on_batch(99999, T0 + 6 * NS, 20) && a.on_batch(1, T0 + 6 * NS, 20));   // 100 000 operations (:145)
        a.swept(15, 20, T0 + 6 * NS);                                    // removed > half: interval halves (:191-195)
        CHECK(a.current_interval_ns() == 5 * NS);
        // with_capacity(1000) asks the map for 1300 entries; hashbrown gives it 2048 buckets = capacity() 1792; 3/4 = 1344 (:166)
        CHECK(a.map_capacity() == 1792);
        CHECK(!a.on_batch(1, T0 + 7 * NS, 1344) && a.on_batch(1, T0 + 7 * NS, 1345) && a.why() == AdaptiveSweep::BY_SIZE);
        a.swept(600, 976, T0 + 7 * NS);                                  // productive cleanup: threshold 10 %
        CHECK(!a.on_batch(1, T0 + 8 * NS, 400, 40));                     // <= 50 expired seen: no ratio trigger (:150)
        CHECK(a.on_batch(1, T0 + 8 * NS, 400, 20));                      // 60 seen / 400 = 15 % > 10 %
        a.swept(1, 400, T0 + 8 * NS);                                    // unproductive: threshold 25 %
        CHECK(!a.on_batch(1, T0 + 9 * NS, 400, 60) && a.on_batch(1, T0 + 9 * NS, 400, 50)); // 60/400 = 15 % no; 110/400 = 27.5 % yes
        for (int i = 0; i < 12; ++i) a.swept(0, 400, T0 + (10 + i) * NS), a.on_batch(0, T0, 400, 0);
        CHECK(a.current_interval_ns() <= 300 * NS);                      // capped at max_cleanup_interval
    }
    { // the two-step form (an owner that enqueues the sweep and hears its result later) == cleanup()'s bookkeeping
        AdaptiveSweep one(T0, 1000), two(T0, 1000);
        CHECK(one.on_batch(7, T0 + 5 * NS, 100) && two.on_batch(7, T0 + 5 * NS, 100) && one.why() == AdaptiveSweep::BY_TIME);
        one.swept(80, 100, T0 + 5 * NS);
        two.swept_begin(T0 + 5 * NS);
        CHECK(!two.on_batch(3, T0 + 6 * NS, 100) && two.operations() == 3);    // between the two steps the trigger is quiet
        two.swept_result(80, 100);
        CHECK(one.current_interval_ns() == two.current_interval_ns() && one.next_cleanup_ns() == two.next_cleanup_ns());
        CHECK(two.current_interval_ns() == 5 * NS / 2 && two.last_removed() == 80);
    }
    { // hashbrown's capacity() and the virtual growth of the map
        CHECK(AdaptiveSweep::hashbrown_capacity(0) == 0 && AdaptiveSweep::hashbrown_capacity(3) == 3 && AdaptiveSweep::hashbrown_capacity(7) == 7);
        CHECK(AdaptiveSweep::hashbrown_capacity(8) == 14 && AdaptiveSweep::hashbrown_capacity(14) == 14 && AdaptiveSweep::hashbrown_capacity(15) == 28);
        CHECK(AdaptiveSweep::hashbrown_capacity(1300) == 1792 && AdaptiveSweep::hashbrown_capacity(1792) == 1792);
        AdaptiveSweep a(T0, 1000);
        a.grow_map(1792);
        CHECK(a.map_capacity() == 1792);
        a.grow_map(1793);
        CHECK(a.map_capacity() == 3584);
        CHECK(!a.on_batch(1, T0, 2688) && a.on_batch(1, T0, 2689));      // 3/4 of the grown map
    }
    std::puts("all tests passed");
    return 0;
}
