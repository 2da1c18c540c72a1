// bucketbench.hip -- the bucket path's kernels alone (csrc/bucket_path.hpp), timed one by one with HIP events
// and checked against the sequence (tc::gcra_step on the host, request by request).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/bucketbench.hip -o tools/bucketbench.bin
// run:   tools/bucketbench.bin [keys=10000000] [batch=1048576] [iters=20] [zipf=0] [fixed=0] [full=0] [lb] [skew]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../throttlecrab_amd/csrc/bucket_path.hpp"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static uint64_t splitmix(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const uint64_t keys = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const uint32_t n = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : (1u << 20);
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int zipf = argc > 4 ? atoi(argv[4]) : 0;
    const int fixed = argc > 5 ? atoi(argv[5]) : 0;
    const int full = argc > 6 ? atoi(argv[6]) : 0;
    const int lb = argc > 7 ? atoi(argv[7]) : bp::pick_lb(keys, n);
    const uint32_t skew = argc > 8 ? (uint32_t)strtoul(argv[8], 0, 10) : bp::MAX_SKEW; // (the harness has no sort path: longer buckets are skipped)
    const int NB = 8, CHECK = 3;
    const int64_t T0 = 1700000000ll * 1000000000ll;
    printf("keys %llu batch %u iters %d zipf %d fixed %d full %d lb %d\n", (unsigned long long)keys, n, iters, zipf, fixed, full, lb);

    // request streams
    std::vector<std::vector<uint32_t>> hb(NB, std::vector<uint32_t>(n));
    std::vector<double> cdf;
    std::vector<uint32_t> perm;
    if (zipf) {
        cdf.resize(keys);
        double acc = 0;
        for (uint64_t k = 0; k < keys; ++k) cdf[k] = (acc += std::pow((double)(k + 1), -1.1));
        for (auto& v : cdf) v /= acc;
        perm.resize(keys);
        for (uint64_t k = 0; k < keys; ++k) perm[k] = (uint32_t)k;
        for (uint64_t k = keys - 1; k > 0; --k) std::swap(perm[k], perm[splitmix(k ^ 33) % (k + 1)]);
    }
    for (int b = 0; b < NB; ++b)
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t r = splitmix((uint64_t)b * n + i + 2ull * 0xD1B54A32D192ED03ull);
            if (zipf) {
                const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
                uint64_t rank = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
                hb[b][i] = perm[std::min<uint64_t>(rank, keys - 1)];
            } else {
                hb[b][i] = (uint32_t)(((r >> 32) * keys) >> 32);
            }
        }

    // device state
    tc::Cell* d_cells = nullptr;
    int64_t* d_tat8 = nullptr;
    if (fixed) {
        CK(hipMalloc(&d_tat8, keys * 8));
        std::vector<int64_t> init(keys, tc::TAT_VACANT);
        CK(hipMemcpy(d_tat8, init.data(), keys * 8, hipMemcpyHostToDevice));
    } else {
        CK(hipMalloc(&d_cells, keys * sizeof(tc::Cell)));
        CK(hipMemset(d_cells, 0, keys * sizeof(tc::Cell)));
    }
    tc::RateClass hcls[2] = {{0, 0, 0, 0}, {0, 0, 100, 0}};
    if (tc::derive_rate(100, 1000, 3600, hcls[1].ei, hcls[1].dvt) != tc::ST_OK) return 1;
    tc::RateClass* d_cls;
    CK(hipMalloc(&d_cls, sizeof hcls));
    CK(hipMemcpy(d_cls, hcls, sizeof hcls, hipMemcpyHostToDevice));
    const size_t cnt_words = (TC_CNT_COUNT + 1) + (size_t)ev::NSHARD * ev::SHARD_WORDS;
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_cnt, cnt_words * 8));
    CK(hipMemset(d_cnt, 0, cnt_words * 8));
    uint32_t* d_slots;
    CK(hipMalloc(&d_slots, (size_t)NB * n * 4));
    for (int b = 0; b < NB; ++b) CK(hipMemcpy(d_slots + (size_t)b * n, hb[b].data(), (size_t)n * 4, hipMemcpyHostToDevice));
    uint8_t* d_allowed;
    CK(hipMalloc(&d_allowed, n));
    int64_t* d_full[3] = {nullptr, nullptr, nullptr};
    if (full)
        for (auto& p : d_full) CK(hipMalloc(&p, (size_t)n * 8));
    const uint32_t nbk = bp::buckets_of(keys, lb);
    void* d_work;
    CK(hipMalloc(&d_work, bp::work_bytes(n, nbk)));
    bp::Work w = bp::carve(d_work, n, nbk, lb);
    w.skew = skew;
    ev::PendEntry* d_park;
    CK(hipMalloc(&d_park, (size_t)n * sizeof(ev::PendEntry)));
    const uint32_t tiles = bp::tiles_of(n);
    printf("buckets %u tiles %u scatter LDS %zu B\n", nbk, tiles, bp::scatter_lds_bytes(nbk));

    ev::Params p;
    memset(&p, 0, sizeof p);
    p.n = n;
    p.flags = ev::F_REGISTERED | ev::F_UNIFORM_CLASS | (fixed ? ev::F_FIXED : 0u);
    p.q_s = 1;
    p.allowed = d_allowed;
    if (full) {
        p.remaining = d_full[0];
        p.reset = d_full[1];
        p.retry = d_full[2];
    }
    p.cells = d_cells;
    p.tat8 = d_tat8;
    p.classes = d_cls;
    p.uniform_class = 1;
    p.capacity = keys;
    p.counters = d_cnt;

    hipEvent_t ev[5];
    for (auto& e : ev) CK(hipEventCreate(&e));
    double sum[4] = {0, 0, 0, 0};
    int timed = 0;
    std::vector<tc::Cell> model(keys, tc::Cell{0, 0});
    std::vector<uint8_t> got(n), want(n);
    std::vector<int64_t> gfull(n);
    uint32_t maxb = 0;
    long bad = 0;
    const int total = CHECK + 3 + 2 * iters;
    const size_t elds = bp::eval_lds_bytes(lb);
    auto batch = [&](int it, bool events) {
        const uint32_t* slots = d_slots + (size_t)(it % NB) * n;
        p.slot = slots;
        p.now_s = T0 + (int64_t)it * 1000000;
        if (events) CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(bp::k_tile_hist, dim3(tiles), dim3(bp::TILE_THREADS), nbk * 4, 0, slots, n, (uint32_t)keys, w);
        if (events) CK(hipEventRecord(ev[1]));
        hipLaunchKernelGGL(bp::k_bucket_scan, dim3(bp::scan_blocks(nbk)), dim3(bp::SCAN_THREADS), 0, 0, w, tiles);
        if (events) CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(bp::k_scatter, dim3(tiles), dim3(bp::TILE_THREADS), bp::scatter_lds_bytes(nbk), 0, slots, n, (uint32_t)keys, w);
        if (events) CK(hipEventRecord(ev[3]));
        if (fixed) {
            if (full) hipLaunchKernelGGL((bp::k_bucket_eval<true, true, false>), dim3(nbk), dim3(64), elds, 0, p, w, d_park);
            else hipLaunchKernelGGL((bp::k_bucket_eval<false, true, false>), dim3(nbk), dim3(64), elds, 0, p, w, d_park);
        } else {
            if (full) hipLaunchKernelGGL((bp::k_bucket_eval<true, false, false>), dim3(nbk), dim3(64), elds, 0, p, w, d_park);
            else hipLaunchKernelGGL((bp::k_bucket_eval<false, false, false>), dim3(nbk), dim3(64), elds, 0, p, w, d_park);
        }
        if (events) CK(hipEventRecord(ev[4]));
    };
    double plain_us = 0;
    for (int it = 0; it < total; ++it) {
        if (it == CHECK + 3 + iters) { // second half: no events between the kernels
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev[0]));
            for (; it < total; ++it) batch(it, false);
            CK(hipEventRecord(ev[4]));
            CK(hipEventSynchronize(ev[4]));
            float ms;
            CK(hipEventElapsedTime(&ms, ev[0], ev[4]));
            plain_us = 1e3 * ms / iters;
            break;
        }
        batch(it, true);
        CK(hipGetLastError());
        CK(hipEventSynchronize(ev[4]));
        if (it >= CHECK + 3) {
            for (int k = 0; k < 4; ++k) {
                float ms;
                CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
                sum[k] += ms;
            }
            ++timed;
        }
        if (it < CHECK) { // the sequence, request by request
            CK(hipMemcpy(got.data(), d_allowed, n, hipMemcpyDeviceToHost));
            if (full) CK(hipMemcpy(gfull.data(), d_full[0], (size_t)n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&maxb, w.maxb, 4, hipMemcpyDeviceToHost));
            const std::vector<uint32_t>& s = hb[it % NB];
            long bad_here = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const tc::Decision d = tc::gcra_step<true>(model[s[i]], hcls[1].ei, hcls[1].dvt, 1, p.now_s);
                if ((uint8_t)d.allowed != got[i] || (full && d.remaining != gfull[i])) {
                    if (bad_here < 5) printf("  batch %d request %u slot %u: allowed %d want %d\n", it, i, s[i], got[i], (int)d.allowed);
                    ++bad_here;
                }
            }
            printf("batch %d: %ld mismatches, largest bucket %u\n", it, bad_here, maxb);
            bad += bad_here;
        }
    }
    // resident state against the model (the timed batches too)
    for (int it = CHECK; it < total; ++it) {
        const std::vector<uint32_t>& s = hb[it % NB];
        const int64_t now = T0 + (int64_t)it * 1000000;
        for (uint32_t i = 0; i < n; ++i) tc::gcra_step<false>(model[s[i]], hcls[1].ei, hcls[1].dvt, 1, now);
    }
    long bad_state = 0;
    if (fixed) {
        std::vector<int64_t> h(keys);
        CK(hipMemcpy(h.data(), d_tat8, keys * 8, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < keys; ++k) {
            const tc::Cell c = tc::fixed_cell(h[k], hcls[1].dvt);
            bad_state += c.tat != model[k].tat || c.expiry != model[k].expiry;
        }
    } else {
        std::vector<tc::Cell> h(keys);
        CK(hipMemcpy(h.data(), d_cells, keys * sizeof(tc::Cell), hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k < keys; ++k) bad_state += h[k].tat != model[k].tat || h[k].expiry != model[k].expiry;
    }
    unsigned long long viol = 0;
    CK(hipMemcpy(&viol, d_cnt + (TC_CNT_COUNT + 1) + 3, 8, hipMemcpyDeviceToHost));
    printf("decision mismatches %ld, state mismatches %ld, selfcheck %llu\n", bad, bad_state, viol);
    const char* names[4] = {"k_tile_hist", "k_bucket_scan", "k_scatter", "k_bucket_eval"};
    double tot = 0;
    for (int k = 0; k < 4; ++k) {
        printf("%-14s %8.2f us\n", names[k], 1e3 * sum[k] / timed);
        tot += 1e3 * sum[k] / timed;
    }
    printf("in order total %8.2f us per batch with an event pair per kernel; %.2f us back to back (%.2f G decisions/s)\n", tot, plain_us,
           n / plain_us / 1e3);
    return bad || bad_state || viol ? 2 : 0;
}
