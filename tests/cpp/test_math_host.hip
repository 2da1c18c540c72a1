// Host-side property tests of throttlecrab_amd/csrc/gcra_math.hpp (the very functions the kernels
// inline), no GPU needed.  They pin the three facts the evaluation kernels rest on:
//   (1) closed form == sequence: for a regular run of identical requests, request r sees the cell
//       new0 + (min(r, n_tot) - 1) * inc  (k_eval_sorted);
//   (2) the host's proof obligation for direct stores (all_runs_regular in slots.hip): with
//       ei > 0, dvt > 0, q > 0, ei*q < 2^62, 0 <= now, now + dvt < 2^62, ANY cell whose first request
//       is allowed gives a regular run;
//   (3) late readers are harmless: once the run's allowance is used up, a request evaluated against
//       the FINAL cell gets exactly the decision the closed form gives it from the OLD cell.
// build: hipcc -O2 -std=c++17 tests/cpp/test_math_host.hip -o tests/cpp/test_math_host   (host code only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../throttlecrab_amd/csrc/gcra_math.hpp"
#include "../../throttlecrab_amd/csrc/key_table.hpp"

using tc::Cell;
using tc::Decision;

#define CHECK(c)                                                                \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static std::mt19937_64 rng(12345);
static int64_t pick(const int64_t* v, int n) { return v[rng() % n]; }
static int64_t uni(int64_t lo, int64_t hi) { return lo + (int64_t)(rng() % (uint64_t)(hi - lo + 1)); }

static bool same(const Decision& a, const Decision& b) {
    return a.allowed == b.allowed && a.remaining == b.remaining && a.reset_after == b.reset_after && a.retry_after == b.retry_after;
}

int main() {
    const int64_t T0 = 1700000000LL * 1000000000LL;
    const int64_t EXT[] = {0, 1, -1, INT64_MAX, INT64_MIN, T0, T0 + 5000000000LL, T0 - 7000000000LL, -T0, (int64_t)1 << 62};
    const int64_t BURST[] = {1, 2, 3, 5, 10, 100, 100000, ((int64_t)1 << 32) + 1, INT64_MAX, INT64_MAX / 1000};
    const int64_t COUNT[] = {1, 6, 7, 10, 100, 1000, 120, (int64_t)1 << 62, INT64_MAX};
    const int64_t PERIOD[] = {1, 60, 3600, 86400, INT64_MAX};
    const int64_t QTY[] = {0, 1, 1, 1, 2, 3, 7, 1000, (int64_t)1 << 62};
    uint64_t regular_runs = 0, irregular_runs = 0, direct_cases = 0, late_checks = 0;
    for (int it = 0; it < 400000; ++it) {
        int64_t ei, dvt;
        const int64_t burst = pick(BURST, 10), count = pick(COUNT, 9), period = pick(PERIOD, 5);
        if (tc::derive_rate(burst, count, period, ei, dvt) != tc::ST_OK) continue;
        const int64_t q = pick(QTY, 9);
        const int64_t now = (rng() & 7) ? T0 + uni(0, 100000000000LL) : (int64_t)(rng() >> 2);
        if (tc::check_request(q, now, dvt) != tc::ST_OK) continue;
        Cell c0;
        switch (rng() % 5) {
            case 0: c0.tat = 0, c0.expiry = 0; break;                                              // vacant
            case 1: c0.tat = pick(EXT, 10), c0.expiry = (uint64_t)pick(EXT, 10); break;             // planted by store ops
            case 2: c0.tat = now + uni(-200000000000LL, 200000000000LL), c0.expiry = (uint64_t)now + (uint64_t)uni(0, 400000000000LL); break;
            case 3: c0.tat = now - uni(0, 10) * (ei > 0 ? (ei < ((int64_t)1 << 40) ? ei : 1) : 1), c0.expiry = UINT64_MAX; break;
            default: c0.tat = now + uni(0, 3) * (dvt > 0 && dvt < ((int64_t)1 << 50) ? dvt : 1), c0.expiry = (uint64_t)now + 1; break;
        }
        const int len = 1 + (int)(rng() % 40);
        // the sequence
        Cell before[48];
        Decision seq[48];
        Cell c = c0;
        for (int r = 0; r < len; ++r) {
            before[r] = c;
            seq[r] = tc::gcra_step<true>(c, ei, dvt, q, now);
        }
        const Cell c_final = c;
        if (!seq[0].allowed) { // request 0 denied => nothing changes => all equal request 0
            for (int r = 1; r < len; ++r) CHECK(same(seq[r], seq[0]) && before[r].tat == c0.tat && before[r].expiry == c0.expiry);
            continue;
        }
        Cell after0 = c0;
        (void)tc::gcra_step<true>(after0, ei, dvt, q, now);
        const tc::RunForm f = tc::run_form(after0, ei, dvt, q, now);
        // (2) the direct-store proof obligation
        const int64_t LIM = (int64_t)1 << 62;
        int64_t inc, lim;
        const bool host_regular = ei > 0 && dvt > 0 && q > 0 && !__builtin_mul_overflow(ei, q, &inc) && inc < LIM && now >= 0 &&
                                  !__builtin_add_overflow(now, dvt, &lim) && lim < LIM;
        if (host_regular) {
            ++direct_cases;
            CHECK(f.regular);
        }
        if (!f.regular) {
            ++irregular_runs;
            continue;
        }
        ++regular_runs;
        // (1) closed form == sequence
        for (int r = 1; r < len; ++r) {
            const int64_t j = (int64_t)r < f.n_tot ? (int64_t)r : f.n_tot;
            Cell v;
            v.tat = f.new0 + (j - 1) * f.inc;
            v.expiry = UINT64_MAX;
            CHECK(v.tat == before[r].tat);
            CHECK(before[r].expiry > (uint64_t)now); // the real cell is live, as the closed form assumes
            Cell w = v;
            const Decision d = tc::gcra_step<true>(w, ei, dvt, q, now);
            CHECK(same(d, seq[r]));
            CHECK(d.allowed == ((int64_t)r < f.n_tot));
            if (d.allowed) {
                Cell real = before[r];
                (void)tc::gcra_step<true>(real, ei, dvt, q, now);
                CHECK(w.tat == real.tat && w.expiry == real.expiry); // what the owner lane stores
            }
        }
        // (3) late readers: requests beyond the allowance against the FINAL cell
        if ((int64_t)len >= f.n_tot) {
            for (int r = (int)f.n_tot; r < len; ++r) {
                Cell w = c_final;
                const Decision d = tc::gcra_step<true>(w, ei, dvt, q, now);
                CHECK(!d.allowed && same(d, seq[r]) && w.tat == c_final.tat && w.expiry == c_final.expiry);
                ++late_checks;
            }
        }
    }
    CHECK(regular_runs > 20000 && irregular_runs > 2000 && direct_cases > 20000 && late_checks > 50000);
    std::printf("regular %llu irregular %llu direct-proof cases %llu late-reader checks %llu\n", (unsigned long long)regular_runs,
                (unsigned long long)irregular_runs, (unsigned long long)direct_cases, (unsigned long long)late_checks);
    // key_table.hpp: the hash of a short key computed from its two padded words == the byte-wise hash,
    // for every length 0..16 (k_probe takes this path; the table and snapshots hold hash_key values)
    {
        unsigned long long short_keys = 0;
        uint8_t buf[32];
        for (int it = 0; it < 200000; ++it) {
            const uint32_t len = (uint32_t)(rng() % 17);
            for (uint8_t& b : buf) b = (uint8_t)rng();
            uint64_t k0, k1, a, b;
            kt::short_key_words(buf, len, k0, k1);
            memcpy(&a, buf, 8);
            memcpy(&b, buf + 8, 8);
            CHECK(kt::keep_bytes(a, len) == k0 && (len > 8 ? kt::keep_bytes(b, len - 8) : 0ull) == k1);
            CHECK(kt::hash_short(k0, k1, len) == kt::hash_key(buf, len));
            ++short_keys;
        }
        CHECK(short_keys == 200000);
    }
    std::puts("all tests passed");
    return 0;
}
