// maint.hip -- maintenance: AdaptiveStore::cleanup (tc_sweep_expired), top denied keys, raw state for differential tests
#include "engine.hpp"

// everything that may touch the table of retired keys has run: key stages on the key stream (k_bind: resurrect_denials), sweeps
// and everything else on the engine's stream
static int drain_key_work(tc_engine* e) {
    hipStream_t s = cur_stream(e);
    if (e->k_busy) {
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    TC_HIP(e, hipStreamSynchronize(s));
    return TC_E_OK;
}

static bool by_count_then_key(const std::pair<std::string, uint64_t>& a, const std::pair<std::string, uint64_t>& b) {
    return a.second != b.second ? a.second > b.second : a.first < b.first;
}

// The table of retired keys, rewritten from the host without its tombstones -- and, past 3 x the reference's limit, trimmed to
// the `limit` most denied keys (TopDeniedKeys::cleanup, metrics.rs:52-64).  `force`: whatever the statistics say.  The engine
// is drained first.  `out` (optional): the keys the table holds afterwards.
static int compact_retired(tc_engine* e, bool force, std::vector<std::pair<std::string, uint64_t>>* out) {
    TC_TRY(drain_key_work(e));
    hipStream_t s = cur_stream(e);
    kt::RetiredRec stats;
    TC_HIP(e, hipMemcpyAsync(&stats, e->retired + kt::RETIRED_CAP, sizeof stats, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    const bool due = stats.count != 0u || stats.len > kt::RETIRED_CAP / 2u;
    if (!due && !force && !out) return TC_E_OK;
    std::vector<kt::RetiredRec> rt(kt::RETIRED_CAP);
    TC_HIP(e, hipMemcpyAsync(rt.data(), e->retired, rt.size() * sizeof(kt::RetiredRec), hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    std::vector<std::pair<std::string, uint64_t>> keep;
    for (const kt::RetiredRec& r : rt)
        if ((r.tag & kt::RT_VALID) && r.count) keep.emplace_back(std::string((const char*)r.bytes, r.len), (uint64_t)r.count);
    const bool trim = keep.size() > 3u * TOPK_MAX;
    if (trim) {
        std::sort(keep.begin(), keep.end(), by_count_then_key);
        keep.resize(TOPK_MAX);
    }
    if (due || force || trim) {
        std::fill(rt.begin(), rt.end(), kt::RetiredRec{});
        // most denied first, and no key further from its home than the device looks (RETIRED_PROBES records: retire_denials /
        // resurrect_denials stop there -- a key placed beyond would be invisible to them, then duplicated or its count lost;
        // ADVICE r4).  A key that does not fit is dropped like the keys TopDeniedKeys::cleanup drops (the least denied go first).
        std::sort(keep.begin(), keep.end(), by_count_then_key);
        size_t kept = 0;
        for (const auto& kv : keep) {
            const uint64_t h = kt::hash_key((const uint8_t*)kv.first.data(), (uint32_t)kv.first.size());
            uint32_t pos = (uint32_t)(h >> 17) & (kt::RETIRED_CAP - 1u);
            uint32_t probes = 0;
            while (rt[pos].tag != kt::RT_EMPTY && probes < kt::RETIRED_PROBES) pos = (pos + 1u) & (kt::RETIRED_CAP - 1u), ++probes;
            if (probes >= kt::RETIRED_PROBES) continue;
            if (kept != (size_t)(&kv - keep.data())) keep[kept] = kv; // (compacts `keep` in place: what the table holds afterwards)
            ++kept;
            rt[pos].tag = h | kt::RT_VALID;
            rt[pos].count = (uint32_t)kv.second;
            rt[pos].len = (uint32_t)kv.first.size();
            memcpy(rt[pos].bytes, kv.first.data(), kv.first.size());
        }
        keep.resize(kept);
        stats = kt::RetiredRec{};
        stats.len = (uint32_t)keep.size();
        TC_HIP(e, hipMemcpyAsync(e->retired, rt.data(), rt.size() * sizeof(kt::RetiredRec), hipMemcpyHostToDevice, s));
        TC_HIP(e, hipMemcpyAsync(e->retired + kt::RETIRED_CAP, &stats, sizeof stats, hipMemcpyHostToDevice, s));
        TC_HIP(e, hipStreamSynchronize(s));
    }
    if (out) *out = std::move(keep);
    return TC_E_OK;
}

// AdaptiveStore::cleanup at now_ns, enqueued on the engine's stream (tc_sweep_expired without the wait; the engine's own
// cleanups, autosweep.hip, come through here too).  The number removed lands in counters[TC_CNT_COUNT] (scratch) and is added
// to TC_CNT_SWEPT.
int sweep_enqueue(tc_engine* e, int64_t now_ns, bool may_go_aside) {
    hipStream_t s = cur_stream(e);
    unsigned long long* scratch = e->counters + TC_CNT_COUNT;
    // (round 6: right behind a pipelined key batch, on the key stream and beside that batch's evaluation -- keys.hip)
    if (may_go_aside && e->key_mode && e->sweep_aside && e->touched && e->key_stream && e->k_busy && !e->m_busy && aside_fresh(e) && e->aside_set >= 0 && e->aside_streak >= 2u && e->depth >= 2 &&
        !e->prof_on && e->sets[e->aside_set].in_use)
        return sweep_keys_device(e, now_ns, scratch, true);
    aside_reset(e);
    if (e->k_busy) { // key stages still in flight on the key stream come first
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    if (e->key_mode) {
        TC_TRY(sweep_keys_device(e, now_ns, scratch)); // (records m_done: the sweep, and a rebuild, change the key table)
    } else {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(nblocks(e->capacity), SWEEP_GRID);
        if (e->fixed)
            hipLaunchKernelGGL(k_sweep_fixed, dim3(blocks), dim3(BLOCK), 0, s, e->tat8, e->rate_id, e->classes, (uint32_t)e->uniform_id, e->capacity,
                               now_ns, e->sweep_part);
        else
            hipLaunchKernelGGL(k_sweep, dim3(blocks), dim3(BLOCK), 0, s, e->cells, e->capacity, now_ns, e->sweep_part);
        hipLaunchKernelGGL(k_sweep_fold, dim3(1), dim3(1024), 0, s, (const uint32_t*)e->sweep_part, blocks, scratch, e->counters);
        TC_HIP(e, hipGetLastError());
    }
    return TC_E_OK;
}

extern "C" int tc_sweep_expired(tc_engine* e, int64_t now_ns, uint64_t* removed) {
    if (!e) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    TC_HIP(e, hipSetDevice(e->device));
    // (the engine's own cleanups read the sweep's outcome with kernels on the engine's stream: autosweep.hip -- they stay there)
    TC_TRY(sweep_enqueue(e, now_ns, !auto_sweep_on(e)));
    if (!removed) return TC_E_OK; // asynchronous: the count goes to TC_CNT_SWEPT
    hipStream_t s = cur_stream(e);
    if (e->k_busy) { // (the sweep ran on the key stream)
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    unsigned long long r = 0;
    TC_HIP(e, hipMemcpyAsync(&r, e->counters + TC_CNT_COUNT, sizeof r, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    *removed = r;
    // a synchronous sweep is where the table of retired keys is looked after (the sweep is what fills it)
    if (e->key_mode && e->retired) TC_TRY(compact_retired(e, false, nullptr));
    return TC_E_OK;
}

extern "C" int tc_read_state(tc_engine* e, uint64_t first, uint64_t n, int64_t* tat, uint64_t* expiry) {
    if (!e || n > e->capacity || first > e->capacity - n) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (n == 0) return TC_E_OK;
    TC_HIP(e, hipSetDevice(e->device));
    if (e->fixed) { // expiry == tat + dvt of the key's plan (gcra_math.hpp: fixed_cell)
        std::vector<int64_t> t(n);
        std::vector<uint16_t> id(n);
        TC_HIP(e, hipMemcpyAsync(t.data(), e->tat8 + first, n * sizeof(int64_t), hipMemcpyDeviceToHost, cur_stream(e)));
        TC_HIP(e, hipMemcpyAsync(id.data(), e->rate_id + first, n * sizeof(uint16_t), hipMemcpyDeviceToHost, cur_stream(e)));
        TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
        for (uint64_t i = 0; i < n; ++i) {
            const int64_t dvt = id[i] < e->host_classes.size() ? e->host_classes[id[i]].dvt : 0;
            const Cell c = tc::fixed_cell(t[i], dvt);
            if (tat) tat[i] = c.tat;
            if (expiry) expiry[i] = c.expiry;
        }
        return TC_E_OK;
    }
    std::vector<Cell> h(n);
    TC_HIP(e, hipMemcpyAsync(h.data(), e->cells + first, n * sizeof(Cell), hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    for (uint64_t i = 0; i < n; ++i) {
        if (tat) tat[i] = h[i].tat;
        if (expiry) expiry[i] = h[i].expiry;
    }
    return TC_E_OK;
}

// ---- denied-key metrics ---------------------------------------------------------
extern "C" int tc_top_denied(tc_engine* e, uint32_t k, uint32_t* slots, uint64_t* counts, uint32_t* n_out) {
    if (!e || !slots || !counts || !n_out) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (!e->denied) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_TRACK_DENIED");
    *n_out = 0;
    if (k == 0) return TC_E_OK;
    if (k > TOPK_MAX) k = TOPK_MAX; // MAX_DENIED_KEYS_LIMIT
    TC_HIP(e, hipSetDevice(e->device));
    hipStream_t s = cur_stream(e);
    uint32_t* hist = e->topk_ws;
    uint32_t* n2 = e->topk_ws + 256;
    uint32_t* lists = e->topk_ws + 258;
    const dim3 grid(std::min<uint64_t>(nblocks(e->capacity), 2048)), block(BLOCK);
    // radix select: T = the k-th largest non-zero count (1 if fewer than k keys were ever denied)
    uint32_t prefix = 0, mask = 0, want = k, T = 1;
    bool all = false;
    for (int shift = 24; shift >= 0 && !all; shift -= 8) {
        uint32_t h[256];
        TC_HIP(e, hipMemsetAsync(hist, 0, 256 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_denied_hist, grid, block, 0, s, e->denied, e->capacity, prefix, mask, (uint32_t)shift, hist);
        TC_HIP(e, hipMemcpyAsync(h, hist, sizeof h, hipMemcpyDeviceToHost, s));
        TC_HIP(e, hipStreamSynchronize(s));
        if (shift == 24) {
            uint64_t total = 0;
            for (uint32_t v : h) total += v;
            if (total <= k) { // everything that was ever denied fits
                all = true;
                break;
            }
        }
        int d = 255;
        for (; d > 0; --d) {
            if (h[d] >= want) break;
            want -= h[d];
        }
        prefix |= (uint32_t)d << shift;
        mask |= 255u << shift;
        T = prefix;
    }
    if (all) T = 1;
    TC_HIP(e, hipMemsetAsync(n2, 0, 2 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_denied_collect, grid, block, 0, s, e->denied, e->capacity, T, n2, lists, TOPK_MAX);
    TC_HIP(e, hipGetLastError());
    uint32_t hn[2];
    TC_HIP(e, hipMemcpyAsync(hn, n2, sizeof hn, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    const uint32_t n_gt = std::min(hn[0], TOPK_MAX), n_eq = std::min(hn[1], TOPK_MAX);
    std::vector<uint32_t> gt(2 * (size_t)n_gt), eq(2 * (size_t)n_eq);
    if (n_gt) TC_HIP(e, hipMemcpyAsync(gt.data(), lists, gt.size() * 4, hipMemcpyDeviceToHost, s));
    if (n_eq) TC_HIP(e, hipMemcpyAsync(eq.data(), lists + 2 * (size_t)TOPK_MAX, eq.size() * 4, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    std::vector<std::pair<uint32_t, uint32_t>> ent; // (count, slot)
    for (uint32_t i = 0; i < n_gt; ++i) ent.emplace_back(gt[2 * i + 1], gt[2 * i]);
    std::vector<std::pair<uint32_t, uint32_t>> ties;
    for (uint32_t i = 0; i < n_eq; ++i) ties.emplace_back(eq[2 * i + 1], eq[2 * i]);
    std::sort(ties.begin(), ties.end(), [](auto& a, auto& b) { return a.second < b.second; }); // deterministic choice among ties
    for (auto& t2 : ties) {
        if (ent.size() >= k) break;
        ent.push_back(t2);
    }
    std::sort(ent.begin(), ent.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    if (ent.size() > k) ent.resize(k);
    for (size_t i = 0; i < ent.size(); ++i) {
        slots[i] = ent[i].second;
        counts[i] = ent[i].first;
    }
    *n_out = (uint32_t)ent.size();
    return TC_E_OK;
}

extern "C" int tc_denied_reset(tc_engine* e) {
    if (!e) return TC_E_INVALID_ARG;
    e->api_seq++;
    if (!e->denied) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_TRACK_DENIED");
    TC_HIP(e, hipSetDevice(e->device));
    if (e->retired) TC_TRY(drain_key_work(e)); // (a bind kernel on the key stream may be moving counts out of the table)
    TC_HIP(e, hipMemsetAsync(e->denied, 0, e->capacity * sizeof(uint32_t), cur_stream(e)));
    if (e->retired) TC_HIP(e, hipMemsetAsync(e->retired, 0, ((size_t)kt::RETIRED_CAP + 1) * sizeof(kt::RetiredRec), cur_stream(e)));
    return TC_E_OK;
}

// Top denied KEYS (key mode): the slots' counters and the side table of keys without a slot, merged.
extern "C" int tc_top_denied_keys(tc_engine* e, uint32_t k, uint8_t* key_bytes, size_t key_bytes_cap, uint32_t* key_off, uint64_t* counts,
                                  uint32_t* n_out) {
    if (!e || !key_off || !counts || !n_out || (key_bytes_cap && !key_bytes)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (!e->key_mode) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_KEY_MODE");
    if (!e->denied) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_TRACK_DENIED");
    *n_out = 0;
    key_off[0] = 0;
    if (k == 0) return TC_E_OK;
    if (k > TOPK_MAX) k = TOPK_MAX;
    // keys that hold a slot.  Keys over 256 bytes are not tracked by the reference (metrics.rs:36-39) and are filtered out, and
    // ties at the cut are broken by KEY bytes while tc_top_denied breaks them by slot: candidates are fetched until every
    // slot whose count reaches the k-th eligible count is among them (or all counted slots are).
    std::vector<std::pair<std::string, uint64_t>> all;
    for (uint32_t k_live = std::min<uint32_t>(k + 64u, TOPK_MAX);;) {
        std::vector<uint32_t> slots(k_live);
        std::vector<uint64_t> cnt(k_live);
        uint32_t n_live = 0;
        TC_TRY(tc_top_denied(e, k_live, slots.data(), cnt.data(), &n_live));
        all.clear();
        if (n_live) {
            std::vector<uint32_t> off(n_live + 1);
            std::vector<uint8_t> bytes(std::max<size_t>(1, (size_t)n_live * 64));
            int rc;
            while ((rc = tc_slot_keys(e, n_live, slots.data(), bytes.data(), bytes.size(), off.data())) == TC_E_INVALID_ARG && bytes.size() < ((size_t)1 << 31))
                bytes.resize(bytes.size() * 4);
            if (rc != TC_E_OK) return rc;
            for (uint32_t i = 0; i < n_live; ++i)
                if (off[i + 1] - off[i] <= kt::RETIRED_KEY) all.emplace_back(std::string((const char*)bytes.data() + off[i], off[i + 1] - off[i]), cnt[i]);
        }
        const bool everything = n_live < k_live || k_live >= TOPK_MAX;
        // (tc_top_denied returns its candidates most denied first: cnt[n_live - 1] is the smallest count fetched)
        const bool cut_is_clean = all.size() >= k && n_live && cnt[n_live - 1] < all[k - 1].second;
        if (everything || cut_is_clean) break;
        k_live = std::min<uint32_t>(k_live * 2u, TOPK_MAX);
    }
    // keys that lost theirs (a key's denials are in exactly one place: see kt::RetiredRec); the table is compacted on the way
    // if its statistics ask for it (it is drained against the key stream first)
    std::vector<std::pair<std::string, uint64_t>> retired;
    TC_TRY(compact_retired(e, false, &retired));
    auto by_count = by_count_then_key;
    all.insert(all.end(), retired.begin(), retired.end());
    std::sort(all.begin(), all.end(), by_count);
    if (all.size() > k) all.resize(k);
    size_t at = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        if (at + all[i].first.size() > key_bytes_cap) return fail(e, TC_E_INVALID_ARG, "tc_top_denied_keys: key_bytes_cap too small");
        memcpy(key_bytes + at, all[i].first.data(), all[i].first.size());
        at += all[i].first.size();
        key_off[i + 1] = (uint32_t)at;
        counts[i] = all[i].second;
    }
    *n_out = (uint32_t)all.size();
    return TC_E_OK;
}

extern "C" int tc_debug_check_keys(tc_engine* e, uint64_t* inconsistencies) {
    if (!e || !inconsistencies) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (!e->key_mode) return fail(e, TC_E_UNSUPPORTED, "tc_debug_check_keys: engine was created without TC_CFG_KEY_MODE");
    TC_HIP(e, hipSetDevice(e->device));
    TC_TRY(drain_key_work(e));
    hipStream_t s = cur_stream(e);
    unsigned long long* bad = nullptr;
    TC_HIP(e, hipMalloc(&bad, 2 * sizeof(unsigned long long)));
    unsigned long long host[2] = {0, 0};
    int top = 0;
    hipError_t rc = hipMemsetAsync(bad, 0, sizeof host, s);
    if (rc == hipSuccess) {
        hipLaunchKernelGGL(k_check_keys, dim3(2048), dim3(BLOCK), 0, s, e->kt, 0, bad);
        hipLaunchKernelGGL(k_check_keys, dim3(2048), dim3(BLOCK), 0, s, e->kt, 1, bad);
        rc = hipGetLastError();
    }
    if (rc == hipSuccess) rc = hipMemcpyAsync(host, bad, sizeof host, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipMemcpyAsync(&top, e->kt.free_top, sizeof top, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    (void)hipFree(bad);
    TC_HIP(e, rc);
    *inconsistencies = host[0] + (host[1] + (uint64_t)(top < 0 ? 0 : top) == e->capacity ? 0u : 1u);
    return TC_E_OK;
}

