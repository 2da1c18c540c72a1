// slots.hip -- tc_rate_limit_batch_slots: staging, grouping (radix sort / bucket path), evaluation, one-launch small batches
#include "engine.hpp"
#include <chrono>

// ---- batch over slots -------------------------------------------------------
// The device-side address of a pinned host array (tc_host_alloc / hipHostMalloc / hipHostRegister), or
// nullptr for pageable memory.
static void* device_view_of_host(const void* host) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory: not an error for us
        return nullptr;
    }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return at.devicePointer;
}

// Results of a TC_B_ASYNC batch -> the caller's host array, on stream `st`: a copy kernel when the array is
// pinned, hipMemcpyAsync otherwise (correct, but the call may wait for the transfer).  hipMemcpyAsync of
// device -> pinned host memory behind kernels stalled the submitting thread for 7-13 ms every few dozen
// batches (host -> device through the SDMA engines does not, and unlike a copy kernel reading over PCIe it
// does not slow the kernels running beside it: 41 vs 100 us for the evaluation).
static int copy_back_async(tc_engine* e, void* host, const void* dev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return TC_E_OK;
    void* hv = device_view_of_host(host);
    if (!hv) {
        TC_HIP(e, copy_async(e, host, dev, bytes, hipMemcpyDeviceToHost, st));
        return TC_E_OK;
    }
    // few blocks: the transfer is bound by the PCIe link, and a grid that fills the chip with waves waiting on
    // the link starves the kernels running beside it
    const uint32_t blocks = (uint32_t)std::min<size_t>((bytes / 16 + BLOCK - 1) / BLOCK + 1, 48);
    hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(BLOCK), 0, st, hv, dev, bytes);
    return TC_E_OK;
}

// Round 5: pageable arrays of a synchronous batch of a few thousand requests.  hipMemcpyAsync stages pageable memory itself, a
// blocking copy at a time (seven in, one or more out): 201 us per reference-shaped call of 4 Ki requests where the same call
// from pinned arrays takes 77.  Two memcpys by the caller's own thread (a few hundred KB) and the pinned path are cheaper.
bool bounce_in(tc_engine* e, const tc_batch& b, Bounced& bo) {
    bo.on = false;
    if (!e->bounce_max || e->fault_countdown || (b.flags & (TC_B_DEVICE_PTRS | TC_B_ASYNC | TC_B_PLAN_DICT)) || b.n_segments) return false;
    const uint64_t n = b.n;
    const size_t key_total = b.key_off ? b.key_off[n] : 0;
    size_t need = 0;
    auto room = [&](size_t bytes) { need += (bytes + 63) & ~(size_t)63; };
    if (b.slot) room(n * 4);
    if (b.key_off) room((n + 1) * 4), room(key_total + 16);
    const int64_t* cols[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    for (const int64_t* c : cols)
        if (c) room(n * 8);
    if (b.allowed) room(n);
    if (b.allowed_bits) room(((n + 63) / 64) * 8);
    int64_t* const o4[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    for (int64_t* c : o4)
        if (c) room(n * 8);
    if (b.status) room(n);
    if (b.result4) room(n * 32);
    if (b.decisions) room(n * sizeof(tc_decision));
    const bool grouped = b.order && (b.flags & TC_B_GROUPED_OUTPUT);
    if (grouped) room(n * 4);
    if (need > e->bounce_max || host_arrays_pinned(b)) return false;
    if (!e->bounce && hipHostMalloc((void**)&e->bounce, e->bounce_max, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        e->bounce = nullptr;
        return false; // (no pinned memory to be had: the plain path)
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* at = e->bounce + off;
        off += (bytes + 63) & ~(size_t)63;
        return at;
    };
    bo.bb = b;
    bo.n_out = 0;
    auto in = [&](const void* user, size_t bytes) -> void* {
        void* at = take(bytes);
        memcpy(at, user, bytes);
        return at;
    };
    auto out = [&](void* user, size_t bytes) -> void* {
        void* at = take(bytes);
        bo.outs[bo.n_out++] = Bounced::Out{user, at, bytes};
        return at;
    };
    if (b.slot) bo.bb.slot = (const uint32_t*)in(b.slot, n * 4);
    if (b.key_off) {
        bo.bb.key_off = (const uint32_t*)in(b.key_off, (n + 1) * 4);
        uint8_t* kb = take(key_total + 16);
        if (key_total) memcpy(kb, b.key_bytes, key_total);
        bo.bb.key_bytes = kb;
    }
    const int64_t** bcols[5] = {&bo.bb.max_burst, &bo.bb.count_per_period, &bo.bb.period, &bo.bb.quantity, &bo.bb.now_ns};
    for (int j = 0; j < 5; ++j)
        if (cols[j]) *bcols[j] = (const int64_t*)in(cols[j], n * 8);
    if (b.allowed) bo.bb.allowed = (uint8_t*)out(b.allowed, n);
    if (b.allowed_bits) bo.bb.allowed_bits = (uint64_t*)out(b.allowed_bits, ((n + 63) / 64) * 8);
    int64_t** bo4[4] = {&bo.bb.limit, &bo.bb.remaining, &bo.bb.reset_after_ns, &bo.bb.retry_after_ns};
    for (int j = 0; j < 4; ++j)
        if (o4[j]) *bo4[j] = (int64_t*)out(o4[j], n * 8);
    if (b.status) bo.bb.status = (uint8_t*)out(b.status, n);
    if (b.result4) bo.bb.result4 = (int64_t*)out(b.result4, n * 32);
    if (b.decisions) bo.bb.decisions = (tc_decision*)out(b.decisions, n * sizeof(tc_decision));
    if (grouped) bo.bb.order = (uint32_t*)out(b.order, n * 4);
    bo.on = true;
    return true;
}

void bounce_out(const Bounced& bo) {
    for (int j = 0; j < bo.n_out; ++j) memcpy(bo.outs[j].user, bo.outs[j].pinned, bo.outs[j].bytes);
}

int stage_in_multi(tc_engine* e, const void* const* src, void* const* dst, const size_t* bytes, uint32_t count, hipStream_t s) {
    mk::CopySegs sg;
    memset(&sg, 0, sizeof sg);
    bool pinned = count != 0 && count <= 8 && !e->fault_countdown && !e->copy_kernel_off;
    size_t largest = 0;
    uint32_t used = 0;
    for (uint32_t i = 0; pinned && i < count; ++i) {
        if (!bytes[i]) continue;
        void* hv = device_view_of_host(src[i]);
        if (!hv) {
            pinned = false;
            break;
        }
        sg.src[used] = hv;
        sg.dst[used] = dst[i];
        sg.bytes[used] = bytes[i];
        largest = std::max(largest, bytes[i]);
        ++used;
    }
    if (pinned && used) {
        // (few blocks per segment: the transfer is bound by the link, not by the number of waves waiting on it)
        const uint32_t bx = (uint32_t)std::min<size_t>((largest / 16 + BLOCK - 1) / BLOCK + 1, 32);
        hipLaunchKernelGGL(mk::k_copy_multi, dim3(bx, used), dim3(BLOCK), 0, s, sg);
        TC_HIP(e, hipGetLastError());
        return TC_E_OK;
    }
    for (uint32_t i = 0; i < count; ++i)
        if (bytes[i]) TC_HIP(e, copy_async(e, dst[i], src[i], bytes[i], hipMemcpyHostToDevice, s));
    return TC_E_OK;
}

// ---- TC_B_PLAN_DICT (round 6) -----------------------------------------------------------------------------------------------
int dict_check(tc_engine* e, const tc_batch& b) {
    if (!(b.flags & TC_B_PLAN_DICT)) return TC_E_OK;
    if (!b.plan_dict || !b.plan_id || b.n_plans == 0 || b.n_plans > 65536u) return fail(e, TC_E_INVALID_ARG, "TC_B_PLAN_DICT: plan_dict / plan_id / n_plans (1..65536)");
    if (b.max_burst || b.count_per_period || b.period) return fail(e, TC_E_INVALID_ARG, "TC_B_PLAN_DICT: max_burst / count_per_period / period columns must be NULL");
    if (b.quantity && b.quantity32) return fail(e, TC_E_INVALID_ARG, "TC_B_PLAN_DICT: quantity and quantity32 are both given");
    if (b.flags & TC_B_REGISTERED_PARAMS) return fail(e, TC_E_INVALID_ARG, "TC_B_PLAN_DICT: a batch of registered plans carries no plans");
    return TC_E_OK;
}

void dict_expand_on_host(tc_engine* e, tc_batch& b) {
    const uint64_t n = b.n;
    for (int j = 0; j < 3; ++j) e->dict_wide[j].resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t k = b.plan_id[i];
        const bool ok = k < b.n_plans;
        for (int j = 0; j < 3; ++j) e->dict_wide[j][i] = ok ? b.plan_dict[3 * (size_t)k + j] : 0;
    }
    b.max_burst = e->dict_wide[0].data(), b.count_per_period = e->dict_wide[1].data(), b.period = e->dict_wide[2].data();
    if (b.quantity32) {
        e->dict_wide[3].resize(n);
        for (uint64_t i = 0; i < n; ++i) e->dict_wide[3][i] = (int64_t)b.quantity32[i];
        b.quantity = e->dict_wide[3].data();
    }
    b.flags &= ~TC_B_PLAN_DICT;
    b.plan_dict = nullptr, b.plan_id = nullptr, b.quantity32 = nullptr, b.n_plans = 0;
}

int dict_expand_device(tc_engine* e, tc_batch& b) {
    const uint32_t n = (uint32_t)b.n;
    hipStream_t s = cur_stream(e);
    for (int j = 0; j < (b.quantity32 ? 4 : 3); ++j) TC_TRY(stage_need(e, e->stage.in[j], e->max_batch));
    // (the engine's staging columns: whatever read them last was enqueued on this stream before)
    hipLaunchKernelGGL(mk::k_expand_plans, dim3(std::min<uint32_t>(nblocks(n), 2048u)), dim3(BLOCK), 0, s, b.plan_dict, b.n_plans, b.plan_id, b.quantity32, n, e->stage.in[0],
                       e->stage.in[1], e->stage.in[2], e->stage.in[3]);
    TC_HIP(e, hipGetLastError());
    b.max_burst = e->stage.in[0], b.count_per_period = e->stage.in[1], b.period = e->stage.in[2];
    if (b.quantity32) b.quantity = e->stage.in[3];
    b.flags &= ~TC_B_PLAN_DICT;
    b.plan_dict = nullptr, b.plan_id = nullptr, b.quantity32 = nullptr, b.n_plans = 0;
    return TC_E_OK;
}

int stage_compact(tc_engine* e, const int64_t* dict, uint32_t n_plans, const uint16_t* plan_id, const uint32_t* q32, uint32_t n, hipStream_t s, int64_t*& d_dict,
                  uint16_t*& d_id, uint32_t*& d_q32, int64_t* const out[4]) {
    TC_TRY(stage_need(e, d_dict, (size_t)65536 * 3));
    TC_TRY(stage_need(e, d_id, e->max_batch));
    if (q32) TC_TRY(stage_need(e, d_q32, e->max_batch));
    const void* src[3] = {dict, plan_id, q32};
    void* dst[3] = {d_dict, d_id, d_q32};
    const size_t bytes[3] = {(size_t)n_plans * 3 * sizeof(int64_t), (size_t)n * sizeof(uint16_t), q32 ? (size_t)n * sizeof(uint32_t) : 0};
    TC_TRY(stage_in_multi(e, src, dst, bytes, 3, s)); // (one copy launch from pinned arrays, else a copy each)
    hipLaunchKernelGGL(mk::k_expand_plans, dim3(std::min<uint32_t>(nblocks(n), 2048u)), dim3(BLOCK), 0, s, (const int64_t*)d_dict, n_plans, (const uint16_t*)d_id,
                       q32 ? (const uint32_t*)d_q32 : (const uint32_t*)nullptr, n, out[0], out[1], out[2], out[3]);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

// output staging for the arrays `b` asks for; `d` gets the device pointers
int stage_outputs(tc_engine* e, const tc_batch& b, tc_batch& d) {
    const uint64_t mb = e->max_batch;
    tc_engine::Stage& st = e->stage;
    if (b.allowed) TC_TRY(stage_need(e, st.allowed, mb));
    if (b.allowed_bits) TC_TRY(stage_need(e, st.bits, (mb + 63) / 64));
    int64_t* const want[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    for (int j = 0; j < 4; ++j)
        if (want[j]) TC_TRY(stage_need(e, st.out[j], mb));
    if (b.status) TC_TRY(stage_need(e, st.status, mb));
    if (b.result4) TC_TRY(stage_need(e, st.result4, mb * 4));
    if (b.decisions) TC_TRY(stage_need(e, st.decisions, mb));
    if (b.order) TC_TRY(stage_need(e, st.order, mb));
    d.allowed = b.allowed ? st.allowed : nullptr;
    d.allowed_bits = b.allowed_bits ? st.bits : nullptr;
    d.limit = b.limit ? st.out[0] : nullptr;
    d.remaining = b.remaining ? st.out[1] : nullptr;
    d.reset_after_ns = b.reset_after_ns ? st.out[2] : nullptr;
    d.retry_after_ns = b.retry_after_ns ? st.out[3] : nullptr;
    d.status = b.status ? st.status : nullptr;
    d.result4 = b.result4 ? st.result4 : nullptr;
    d.decisions = b.decisions ? st.decisions : nullptr;
    d.order = b.order ? st.order : nullptr;
    return TC_E_OK;
}

// staged outputs -> the caller's host arrays, enqueued on `s` (by_kernel: TC_B_ASYNC, see copy_back_async)
int copy_outputs_back(tc_engine* e, const tc_batch& b, hipStream_t s, bool by_kernel) {
    const uint64_t n = b.n;
    const tc_engine::Stage& st = e->stage;
    auto back = [&](void* host, const void* dev, size_t bytes) -> int {
        if (by_kernel) return copy_back_async(e, host, dev, bytes, s);
        TC_HIP(e, copy_async(e, host, dev, bytes, hipMemcpyDeviceToHost, s));
        return TC_E_OK;
    };
    if (b.allowed) TC_TRY(back(b.allowed, st.allowed, n));
    if (b.allowed_bits) TC_TRY(back(b.allowed_bits, st.bits, ((n + 63) / 64) * sizeof(uint64_t)));
    int64_t* hout[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    for (int j = 0; j < 4; ++j)
        if (hout[j]) TC_TRY(back(hout[j], st.out[j], n * sizeof(int64_t)));
    if (b.status) TC_TRY(back(b.status, st.status, n));
    if (b.result4) TC_TRY(back(b.result4, st.result4, n * 4 * sizeof(int64_t)));
    if (b.decisions) TC_TRY(back(b.decisions, st.decisions, n * sizeof(tc_decision)));
    if (b.order && (b.flags & TC_B_GROUPED_OUTPUT)) TC_TRY(back(b.order, st.order, n * sizeof(uint32_t)));
    return TC_E_OK;
}

// every output of a synchronous host batch (+ the key table's error word, into the pinned result block) behind the evaluation in
// one or two k_copy_multi launches; false: not applicable (an output in pageable memory, a failed copy being simulated, too many
// bytes: TCGPU_SYNC_COPY_MAX, default 64 MB -- 0 switches this off) and nothing was enqueued
static bool outputs_back_in_one_launch(tc_engine* e, const tc_batch& b, hipStream_t s, bool with_flag, bool* flag_done) {
    if (e->fault_countdown || e->copy_kernel_off) return false;
    const uint64_t n = b.n;
    const tc_engine::Stage& st = e->stage;
    struct Seg {
        void* host;
        const void* dev;
        size_t bytes;
    } segs[12];
    uint32_t k = 0;
    auto add = [&](void* host, const void* dev, size_t bytes) {
        if (host && bytes) segs[k++] = Seg{host, dev, bytes};
    };
    add(b.allowed, st.allowed, n);
    add(b.allowed_bits, st.bits, ((n + 63) / 64) * sizeof(uint64_t));
    int64_t* hout[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    for (int j = 0; j < 4; ++j) add(hout[j], st.out[j], n * sizeof(int64_t));
    add(b.status, st.status, n);
    add(b.result4, st.result4, n * 4 * sizeof(int64_t));
    add(b.decisions, st.decisions, n * sizeof(tc_decision));
    if (b.flags & TC_B_GROUPED_OUTPUT) add(b.order, st.order, n * sizeof(uint32_t));
    size_t total = 0;
    mk::CopySegs sg[2];
    memset(sg, 0, sizeof sg);
    size_t largest[2] = {0, 0};
    uint32_t used[2] = {0, 0};
    for (uint32_t i = 0; i < k; ++i) {
        void* hv = device_view_of_host(segs[i].host);
        if (!hv) return false;
        total += segs[i].bytes;
        const uint32_t g = i / 8;
        sg[g].src[used[g]] = segs[i].dev, sg[g].dst[used[g]] = hv, sg[g].bytes[used[g]] = segs[i].bytes;
        largest[g] = std::max(largest[g], segs[i].bytes);
        ++used[g];
    }
    static const size_t limit = [] { const char* v = getenv("TCGPU_SYNC_COPY_MAX"); return v ? (size_t)strtoull(v, nullptr, 10) : (size_t)(64u << 20); }();
    if (total > limit) return false;
    if (with_flag) { // (the last group has room: at most 10 outputs + the word)
        const uint32_t g = used[0] < 8 ? 0 : 1;
        void* dv = nullptr;
        if (hipHostGetDevicePointer(&dv, e->host_results, 0) != hipSuccess) return false;
        sg[g].src[used[g]] = e->kt.error_flag, sg[g].dst[used[g]] = static_cast<uint8_t*>(dv) + 192, sg[g].bytes[used[g]] = sizeof(uint32_t);
        largest[g] = std::max(largest[g], sizeof(uint32_t));
        ++used[g];
    }
    for (uint32_t g = 0; g < 2; ++g) {
        if (!used[g]) continue;
        const uint32_t bx = (uint32_t)std::min<size_t>((largest[g] / 16 + BLOCK - 1) / BLOCK + 1, 48);
        hipLaunchKernelGGL(mk::k_copy_multi, dim3(bx, used[g]), dim3(BLOCK), 0, s, sg[g]);
    }
    const bool ok = hipGetLastError() == hipSuccess; // (false: the caller's copies follow and report what is wrong with the stream)
    *flag_done = ok && with_flag;
    return ok;
}

// -> true: the list changed (h.made is the new one)
static bool hot_make(tc_engine::Hot& h, uint64_t capacity) {
    // one note per slot (the longest run), through a scratch hash: 8 192 notes at most, no sort of them all (a first version
    // sorted the notes twice and cost the pipeline 0.3 ms every 8th batch)
    constexpr uint32_t SCR = 2u * ev::HEAVY_SLOTS;
    if (h.scratch.size() != SCR) h.scratch.assign(SCR, 0ull);
    else std::fill(h.scratch.begin(), h.scratch.end(), 0ull);
    auto probe = [&](uint32_t slot) -> unsigned long long& { // entry: (slot + 1) << 32 | length
        uint32_t at = (slot * 0x9E3779B1u) >> (32 - (ev::HEAVY_BITS + 2));
        while (h.scratch[at] != 0ull && (uint32_t)(h.scratch[at] >> 32) != slot + 1u) at = (at + 1u) & (SCR - 1u);
        return h.scratch[at];
    };
    std::vector<unsigned long long>& found = h.found; // length << 32 | ~slot: descending = longest first, then by slot
    found.clear();
    for (uint32_t i = 0; i < ev::HEAVY_SLOTS; ++i) {
        const unsigned long long v = ((volatile unsigned long long*)h.notes_host)[i];
        const uint32_t len = (uint32_t)(v >> 44), slot = (uint32_t)v;
        if (len < h.heavy_min || slot >= capacity) continue;
        unsigned long long& en = probe(slot);
        if ((uint32_t)en < len) en = ((unsigned long long)(slot + 1u) << 32) | len;
    }
    for (const unsigned long long en : h.scratch)
        if (en) found.push_back(((unsigned long long)(uint32_t)en << 32) | (0xFFFFFFFFu - ((uint32_t)(en >> 32) - 1u)));
    if (found.size() > rp::HOT_MAX) {
        std::nth_element(found.begin(), found.begin() + rp::HOT_MAX, found.end(), std::greater<unsigned long long>());
        found.resize(rp::HOT_MAX);
    }
    std::sort(found.begin(), found.end(), std::greater<unsigned long long>());
    std::vector<uint32_t> now(found.size());
    for (size_t i = 0; i < found.size(); ++i) now[i] = 0xFFFFFFFFu - (uint32_t)found[i];
    // Keep the installed tables while the list still serves: nine in ten of its slots are still noted (else it is stale: made
    // afresh, so that the table does not fill with slots that stopped being hot), and no slot with a run long enough to endanger a
    // range (a quarter of what a block finishes) is missing from it.  Which slots are hot is all that counts -- the rank form
    // does not care about their order -- and the tail of a skewed stream's list changes with every look (slots around heavy_min
    // requests per batch come and go; in round 6's first version so did the boundary between the 64th and the 65th heaviest,
    // and every change cost six installs and the hints about the ranges).
    bool same = !h.made.empty();
    if (same) {
        std::vector<uint32_t> cur(h.made);
        std::sort(cur.begin(), cur.end());
        for (size_t i = 0; i < found.size() && same; ++i) {
            const uint32_t len = (uint32_t)(found[i] >> 32), sl = 0xFFFFFFFFu - (uint32_t)found[i];
            if (len < rs::FIN_CAP / 4u) break; // (sorted by length, longest first)
            same = std::binary_search(cur.begin(), cur.end(), sl);
        }
        size_t both = 0;
        for (const uint32_t sl : h.made) both += probe(sl) != 0ull ? 1u : 0u; // (noted at all: on the new list or just below its cut)
        same = same && both * 10u >= h.made.size() * 9u && now.size() * 4u <= h.made.size() * 5u + 128u; // (... nor has a crowd of new hot slots turned up)
    }
    if (now.empty() && h.made.empty()) same = true;
    if (same) {
        h.stable_looks.fetch_add(1u, std::memory_order_relaxed);
        return false;
    }
    h.stable_looks.store(0u, std::memory_order_relaxed);
    h.made.swap(now);
    h.lists_made++;
    return true;
}

static int64_t hot_clock_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The maker's thread: while calls keep coming it looks at the copy's sequence word every ~100 us (so that a list is ready when
// the caller comes back from a wait -- the driver's form: five batches, a device synchronisation, twenty timed batches); after
// 20 ms without a call it sleeps until the next one.
static void hot_worker(tc_engine::Hot* hp, uint64_t capacity) {
    tc_engine::Hot& h = *hp;
    constexpr int64_t IDLE_NS = 20'000'000;
    std::unique_lock<std::mutex> lk(h.mu);
    while (!h.stop) {
        if (hot_clock_ns() - h.last_call_ns.load() > IDLE_NS) {
            h.asleep.store(true);
            if (hot_clock_ns() - h.last_call_ns.load() > IDLE_NS) h.cv.wait_for(lk, std::chrono::milliseconds(500)); // (a call that came in between saw `asleep` or is seen here)
            h.asleep.store(false);
            continue;
        }
        lk.unlock();
        const unsigned long long seq = *(volatile unsigned long long*)(h.notes_host + ev::HEAVY_SLOTS);
        if (seq != h.seq_seen) {
            h.seq_seen = seq;
            if (hot_make(h, capacity)) {
                std::lock_guard<std::mutex> g(h.mu);
                h.next = h.made;
                h.next_ready.store(true, std::memory_order_release);
            }
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        lk.lock();
    }
}

void hot_worker_stop(tc_engine* e) { // (tc_engine_destroy, before the pinned notes are freed)
    tc_engine::Hot& h = e->hot;
    if (!h.worker_on) return;
    {
        std::lock_guard<std::mutex> g(h.mu);
        h.stop = true;
    }
    h.cv.notify_all();
    h.worker.join();
    h.worker_on = false;
}

static void hot_publish(tc_engine* e, hipStream_t st, unsigned long long seq) {
    static_assert(ev::HEAVY_SLOTS % 1024u == 0u, "k_heavy_publish: 1 024 words per block");
    hipLaunchKernelGGL(mk::k_heavy_publish, dim3(ev::HEAVY_SLOTS / 1024u), dim3(1024), 0, st, e->hot.notes_host_dev, e->hot.notes_dev, ev::HEAVY_SLOTS, seq,
                       e->hot.done + 8);
}

static void hot_taken(tc_engine* e, bool had_list) { // a new list in h.slots: the sets install it when they next need it
    tc_engine::Hot& h = e->hot;
    if (++h.version == 0u) h.version = 1u;
    h.backoff = 0;
    *(volatile unsigned long long*)h.hint_cold_host = 0ull; // (what the old list left of the ranges says nothing about the new one)
    // (how many requests the ranges held sizes the sorted part's grid, with a quarter to spare and a second stretch per block if
    // that is short: a list that replaces another one moves it little -- kept, or the next evaluations would get the whole batch's grid)
    if (!had_list) *(volatile unsigned long long*)(h.hint_cold_host + 1) = 0ull;
}

// Round 6: the hot list (range_part.hpp).  The evaluations note every run of at least hot.heavy_min requests in a small device
// table, a copy of which reaches pinned memory now and then (the longest runs since the copy before: the table is cleared with
// every copy, so a stream that stops being skewed empties the list with the next one); whenever a NEW copy has arrived the list
// is made afresh from it: the (up to) rp::HOT_MAX slots with the longest runs, heaviest first.  Never waits; a stream without
// heavy runs leaves the list empty.  (`starts`: this batch takes notes -- the maker's thread is worth having)
static void hot_refresh(tc_engine* e, bool starts) {
    tc_engine::Hot& h = e->hot;
    if (!h.on || !h.notes_host) return;
    if (!h.threaded) {
        const unsigned long long seq = *(volatile unsigned long long*)(h.notes_host + ev::HEAVY_SLOTS);
        if (seq == h.seq_seen) return;
        h.seq_seen = seq;
        if (hot_make(h, e->capacity)) {
            const bool had = !h.slots.empty();
            h.slots = h.made;
            hot_taken(e, had);
        }
        return;
    }
    h.last_call_ns.store(hot_clock_ns());
    if (!h.worker_on) {
        if (!starts) return;
        h.worker = std::thread(hot_worker, &h, (uint64_t)e->capacity);
        h.worker_on = true;
    } else if (h.asleep.load()) {
        std::lock_guard<std::mutex> g(h.mu);
        h.cv.notify_one();
    }
    if (h.next_ready.load(std::memory_order_acquire)) {
        bool had;
        {
            std::lock_guard<std::mutex> g(h.mu);
            had = !h.slots.empty();
            h.slots.swap(h.next);
            h.next_ready.store(false, std::memory_order_relaxed);
        }
        hot_taken(e, had);
    }
}

// How is this batch grouped?  GROUP_RANGE: the range path (radix_sort.hpp / range_part.hpp: every tile partitioned by key range
// in place + one block per range that collects and finishes it in LDS -- two launches instead of a histogram and three LSD
// passes); GROUP_RANGE_RANK (hot_allowed: a lean batch): the same with the slots of the hot list left out of the partition --
// the evaluation's hot role ranks their requests where they stand; GROUP_LSD: the passes.  The host cannot see the batch; it
// goes by the largest range of a RECENT batch of the stream, which every grouping mirrors into pinned memory (never waited
// for).  No hint yet, a hint that predicts a range beyond what a block finishes in LDS and no hot list that could explain it,
// or a batch too large: the LSD passes.  A wrong guess costs time, not correctness (k_finish sorts an oversized range through
// global memory).
enum { GROUP_LSD = 0, GROUP_RANGE = 1, GROUP_RANGE_RANK = 2 };
static int range_applies(tc_engine* e, uint32_t n, bool piped, bool hot_allowed) {
    if (!e->range_ok || !(e->range_mode >= 2 || (e->range_mode == 1 && piped))) return GROUP_LSD;
    const unsigned long long h = *(volatile unsigned long long*)e->range_hint_host;
    const uint64_t hn = h >> 32, hmax = h & 0xFFFFFFFFull;
    // the share of a batch its largest range took (x 2^20), of the last RANGE_HINTS looks at the hint: a stream that
    // alternates between uniform and skewed batches stays on the LSD passes (a skewed batch on the range path costs a
    // block's slow pass over its oversized range: ~0.5 ms for a Zipf(1.1) batch)
    if (hn == 0) return GROUP_LSD; // (no grouping of this engine has run yet)
    e->range_share[e->range_looks++ % tc_engine::RANGE_HINTS] = (uint32_t)std::min<uint64_t>((hmax << 20) / hn, 0xFFFFFFFFull);
    const uint32_t tile = rp::PT_TILE;
    if (n < 256u || n > e->range_max_n || (n + tile - 1) / tile > (uint32_t)rs::FIN_THREADS) return GROUP_LSD;
    uint32_t worst = 0;
    for (uint32_t k = 0; k < tc_engine::RANGE_HINTS; ++k) worst = std::max(worst, e->range_share[k]);
    const uint64_t fits = (uint64_t)(rs::FIN_CAP / 8u * 7u);
    if ((((uint64_t)worst * n) >> 20) <= fits) return GROUP_RANGE;
    // Skewed as a whole -- but is it a few slots that make it so?  With a hot list, the batch is grouped without them; what the
    // largest range then held comes back in a word of its own (the plain hint keeps saying "skewed": nobody counts the whole
    // batch by range while the stream is grouped this way).  If peeling did not help, the passes, for a while.
    tc_engine::Hot& hs = e->hot;
    if (!hot_allowed || !hs.on || hs.slots.empty() || n < 16384u) return GROUP_LSD;
    if (hs.backoff) {
        --hs.backoff;
        return GROUP_LSD;
    }
    const unsigned long long hc = *(volatile unsigned long long*)hs.hint_cold_host;
    if ((hc >> 32) != 0ull && ((hc & 0xFFFFFFFFull) * n) / (hc >> 32) > fits) {
        hs.backoff = 64;
        *(volatile unsigned long long*)hs.hint_cold_host = 0ull;
        return GROUP_LSD;
    }
    return GROUP_RANGE_RANK;
}

// stable sort of (slot, index) by slot in scratch set `ss`, issued on stream `s`;
// returns the buffer holding the result
static const uint64_t* sort_by_slot(tc_engine* e, tc_engine::SortSet& ss, hipStream_t s, const uint32_t* d_slot, uint32_t n,
                                    bool piped, int ranged, const uint32_t* gate = nullptr, uint32_t gate_min = 0, hipEvent_t stop_last = nullptr,
                                    uint8_t* fill = nullptr, uint32_t fill_value = 0) {
    const uint32_t cap = (uint32_t)e->capacity;
    const int bits = std::max(1, bit_width_u64(e->capacity)); // the sentinel key `capacity` must fit
    const int passes = (bits + 7) / 8;
    const int items = piped ? e->sort_items_piped : SORT_ITEMS;
    const uint32_t tile = rs::THREADS * (uint32_t)items;
    const uint32_t tiles = (n + tile - 1) / tile;
    rs::Workspace ws = rs::carve(ss.ws, ss.hist_parity, e->sort_max_tiles);
    ws.violations = e->counters + (TC_CNT_COUNT + 1) + 3;
    uint64_t* bufs[2] = {ss.elem_a, ss.elem_b};
    unsigned long long* hint = e->range_ok ? e->range_hint_dev : nullptr;
    if (ranged) {
        // round 6 (range_part.hpp): tiles of 4 096 requests on 1 024 threads, partitioned in place into elem_b by key range -- and,
        // in the hot form, by hot id: those buckets are gathered behind the ranges' elements, every range is finished into elem_a
        const bool rankm = ranged == GROUP_RANGE_RANK, hotm = rankm;
        const uint32_t ptiles = (n + rp::PT_TILE - 1) / rp::PT_TILE, stride = hotm ? rp::NB_HOT : rp::NR;
        uint32_t* totals = ss.range_totals + (size_t)ss.range_parity * rp::NB_HOT;         // zero: cleared by the finish of the set's previous batch
        uint32_t* totals_next = ss.range_totals + (size_t)(ss.range_parity ^ 1u) * rp::NB_HOT;
        ss.range_parity ^= 1u;
        if (hotm && ss.hot_version != e->hot.version) {
            rp::HotList hl;
            hl.count = (uint32_t)std::min<size_t>(e->hot.slots.size(), rp::HOT_MAX);
            memcpy(hl.slot, e->hot.slots.data(), hl.count * sizeof(uint32_t));
            hipLaunchKernelGGL(rp::k_hot_install, dim3(1), dim3(rp::PT_THREADS), 0, s, hl, ss.hot_dev);
            ss.hot_version = e->hot.version;
        }
        prof_begin_m(e, TC_STAGE_SORT, s);
#define TC_PART(MODE, ILV, HOTP, INFO) \
    TC_LAUNCH_T(e, TC_STAGE_SORT, (hipEvent_t) nullptr, (rp::k_tile_part<MODE, ILV>), dim3(ptiles), dim3(rp::PT_THREADS), 0, s, d_slot, bufs[1], ss.part_table, stride, totals, n, cap, \
                e->range_mul, fill, fill_value, (const rp::HotDev*)(HOTP), (uint32_t*)(INFO))
        // (string mode: the ranges interleaved chunk by chunk -- radix_sort.hpp)
        if (e->range_ilv) {
            if (rankm) TC_PART(rp::PART_RANK, true, ss.hot_dev, ss.hot_info);
            else TC_PART(rp::PART_PLAIN, true, nullptr, nullptr);
        } else if (rankm) TC_PART(rp::PART_RANK, false, ss.hot_dev, ss.hot_info);
        else TC_PART(rp::PART_PLAIN, false, nullptr, nullptr);
#undef TC_PART
        prof_end_m(e, s);
        if (hotm) e->hot.batches_hot++;
        prof_begin_m(e, TC_STAGE_SORT, s);
        hipEvent_t stop = e->prof_on ? nullptr : stop_last;
        // (rank form: HOT_MAX / 32 more blocks scan the hot ids' columns of the table)
#define TC_FIN(ILV, MUL) \
        TC_LAUNCH_T(e, TC_STAGE_SORT, stop, rs::k_finish<ILV>, dim3(rs::NRANGE + (rankm ? rp::HOT_MAX / 32u : 0u)), dim3(rs::FIN_THREADS), 0, s, (const uint64_t*)bufs[1], \
                    (const uint32_t*)ss.part_table, bufs[0], ss.elem_c, (const uint32_t*)totals, totals_next, n, ptiles, rp::PT_TILE, (MUL), e->range_sub_passes, \
                    hotm ? e->hot.hint_cold_dev : hint, stride, rp::NB_HOT, rankm ? ss.hot_P : (uint32_t*)nullptr, rankm ? ss.hot_n : (uint32_t*)nullptr, rp::HOT_MAX, \
                    rankm ? e->hot.hint_cold_dev + 1 : (unsigned long long*)nullptr)
        if (e->range_ilv) TC_FIN(true, cap); // (interleaved: the kernel derives the ranges' width from the capacity)
        else TC_FIN(false, e->range_mul);
#undef TC_FIN
        prof_end_m(e, s);
        return bufs[0];
    }
    ss.hist_parity ^= 1u;
    // the range histogram is counted beside the LSD digits (one more LDS atomic per request) whenever the key space admits
    // the range path: it is where the hint comes from while a stream is on the LSD passes
    const uint32_t msd_mul = e->range_ok ? (e->range_ilv ? rs::RANGE_ILV : e->range_mul) : 0u;
    prof_begin_m(e, TC_STAGE_PREP, s);
    TC_LAUNCH_T(e, TC_STAGE_PREP, (hipEvent_t) nullptr, rs::k_hist<rs::HIST_THREADS>, dim3(rs::HIST_BLOCKS), dim3(rs::HIST_THREADS), 0, s, d_slot, n, cap, passes, ws, tiles, gate, gate_min, fill, fill_value, msd_mul);
    prof_end_m(e, s);
    const uint64_t* in = nullptr;
    for (int p = 0; p < passes; ++p) {
        uint64_t* out = bufs[p & 1];
        prof_begin_m(e, TC_STAGE_SORT, s); // one record per pass: the stage average is per kernel launch
        hipEvent_t stop = (p + 1 == passes && !e->prof_on) ? stop_last : nullptr;
#define TC_PASS(IT, FI) \
    TC_LAUNCH_T(e, TC_STAGE_SORT, stop, (rs::k_onesweep<IT, FI>), dim3(tiles), dim3(rs::THREADS), 0, s, (FI) ? d_slot : (const uint32_t*)nullptr, \
              (FI) ? (const uint64_t*)nullptr : in, out, n, cap, p, ws, gate, gate_min, (FI) ? hint : (unsigned long long*)nullptr)
        if (p == 0) {
            if (items == 32) TC_PASS(32, true);
            else if (items == 16) TC_PASS(16, true);
            else TC_PASS(8, true);
        } else {
            if (items == 32) TC_PASS(32, false);
            else if (items == 16) TC_PASS(16, false);
            else TC_PASS(8, false);
        }
#undef TC_PASS
        prof_end_m(e, s);
        in = out;
    }
    return in;
}

// Uniform batch (one `now`, one `quantity`): is every run regular (tc::run_form) whatever the cells
// hold?  With request 0 of a run allowed, new0 lies in [now - dvt + inc, now + dvt], so the
// cell-dependent provisos of run_form hold by themselves once the parameters satisfy
//     ei > 0,  dvt > 0 (burst >= 2: the entry outlives its own timestamp),  q > 0,
//     ei * q < 2^62,  0 <= now,  now + dvt < 2^62,
// checked here for the batch's scalar rate or for the bounds over all registered plans.
static bool all_runs_regular(const tc_engine* e, const tc_batch& b, const Params& p) {
    const int64_t LIM = (int64_t)1 << 62;
    const int64_t q = p.q_s, now = p.now_s;
    if (q <= 0 || now < 0) return false;
    int64_t lo_ei, hi_ei, lo_dvt, hi_dvt;
    if (p.flags & F_REGISTERED) {
        if (e->uniform_id) {
            const RateClass& rc = e->host_classes[e->uniform_id];
            lo_ei = hi_ei = rc.ei;
            lo_dvt = hi_dvt = rc.dvt;
        } else {
            if (e->host_classes.size() <= 1) return false;
            lo_ei = e->cls_min_ei, hi_ei = e->cls_max_ei, lo_dvt = e->cls_min_dvt, hi_dvt = e->cls_max_dvt;
        }
    } else {
        int64_t ei, dvt;
        if (tc::derive_rate(b.max_burst_scalar, b.count_per_period_scalar, b.period_scalar, ei, dvt) != tc::ST_OK) return false;
        lo_ei = hi_ei = ei;
        lo_dvt = hi_dvt = dvt;
    }
    int64_t inc, lim;
    if (lo_ei <= 0 || lo_dvt <= 0) return false;
    if (__builtin_mul_overflow(hi_ei, q, &inc) || inc >= LIM) return false;
    if (__builtin_add_overflow(now, hi_dvt, &lim) || lim >= LIM) return false;
    return true;
}

// k_eval_sorted<.., ITEMS>: 2 positions per lane when the batch overlaps with its neighbours' sorts,
// 4 when it runs alone (measured; 8 is slower everywhere; TCGPU_EVAL_ITEMS = 1 | 2 | 4 overrides)
template <int ITEMS, bool FIXED>
static void launch_eval_items(tc_engine* e, bool full, bool direct, bool lean, uint32_t n, hipStream_t s, const Params& p, const uint64_t* sorted,
                              uint32_t seq, const uint32_t* gate, uint32_t gate_min, hipEvent_t stop) {
    (void)lean;
    const dim3 grid((n + BLOCK * ITEMS - 1) / (BLOCK * ITEMS)), block(BLOCK);
    if (lean) {
        if constexpr (ITEMS <= 4) {
            // (rank form: the first hot_blocks blocks walk the batch in request order for the hot slots' requests; the sorted part's
            // blocks: what a recent batch left in the ranges, scaled to this batch, + a quarter -- too few: a block takes a second
            // stretch; never more than the whole batch would need)
            const HotEval* hep = e->hot.he_dev;
            uint32_t hot_blocks = 0, cold_grid = grid.x;
            if (hep) {
                hot_blocks = ((n + rp::PT_TILE - 1) / rp::PT_TILE) * HOT_SPLIT; // (a block per half tile)
                const unsigned long long ch = *(volatile unsigned long long*)(e->hot.hint_cold_host + 1);
                if ((ch >> 32) != 0ull) {
                    const uint64_t est = (ch & 0xFFFFFFFFull) * n / (ch >> 32);
                    cold_grid = (uint32_t)std::min<uint64_t>(grid.x, (est + est / 4u) / (BLOCK * ITEMS) + 32u);
                }
                cold_grid = std::max(cold_grid, 1u);
            }
            if (hep) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_lean_hot<ITEMS, FIXED>), dim3(cold_grid + hot_blocks), block, 0, s, p, sorted, e->loaded, seq, e->fill_hint_dev, hep, hot_blocks, cold_grid);
            else TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_sorted_lean<ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->loaded, seq, gate, gate_min, e->fill_hint_dev);
            return;
        }
    }
    if (full && direct) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_sorted<true, true, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else if (full) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_sorted<true, false, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else if (direct) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_sorted<false, true, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_sorted<false, false, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
}

// Sorted positions per lane: 2 for a pipelined batch of a stream that looks uniform (the grid of 2048 blocks of 8 per CU hides the
// random cells best), 4 in order -- and 4 for pipelined batches of a skewed stream (`slim`: the range hint sent the batch to
// the LSD passes), whose evaluation is lighter and gains more from a grid half as large: Zipf 41.7 -> 38.6 us per batch,
// uniform 43.7 -> 46.1 (profiles/r04_v8_items4_ab.txt).
static int eval_items_of(const tc_engine* e, bool piped, bool slim = false) { return e->eval_items ? e->eval_items : ((piped && !slim) ? 2 : 4); }

// LEAN: decisions only, direct stores, nothing asked for but the `allowed` bytes in request order (k_eval_sorted_lean)
static bool lean_applies(const tc_engine* e, bool full, bool direct, bool piped, const Params& p) {
    return e->eval_lean && !full && direct && eval_items_of(e, piped) <= 4 && p.allowed && !p.status && !p.limit && !p.order && !p.row_bits;
}

static void launch_eval_sorted(tc_engine* e, bool full, bool direct, bool piped, uint32_t n, hipStream_t s, const Params& p,
                               const uint64_t* sorted, uint32_t seq, const uint32_t* gate = nullptr, uint32_t gate_min = 0,
                               hipEvent_t stop = nullptr, bool slim = false) {
    const int items = eval_items_of(e, piped, slim);
    const bool lean = lean_applies(e, full, direct, piped, p);
    if (e->fixed) {
        switch (items) {
        case 1: launch_eval_items<1, true>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
        case 2: launch_eval_items<2, true>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
        default: launch_eval_items<4, true>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
        }
    }
    switch (items) {
    case 1: launch_eval_items<1, false>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
    case 2: launch_eval_items<2, false>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
    default: launch_eval_items<4, false>(e, full, direct, lean, n, s, p, sorted, seq, gate, gate_min, stop); return;
    }
}

// The runtime builds a kernel's function object when it is FIRST launched (the symbol's lookup, its argument metadata: 40-150 us
// on the calling thread).  A skewed stream's first batches go through the LSD passes and change over to the rank form once the
// hot list has arrived -- in the driver's form (5 batches, a synchronisation, 20 timed batches) the change-over, and with it the
// first launch of the rank form's three kernels, lay INSIDE the timed region.  Every kernel a pipelined slot batch can launch
// is looked up once per process and device, where the side streams are probed (milliseconds anyway).
static void preload_pipelined_kernels(tc_engine* e) {
    if (e->kernels_preloaded) return;
    e->kernels_preloaded = true;
    static std::mutex mu;
    static std::vector<int> done;
    {
        std::lock_guard<std::mutex> g(mu);
        if (std::find(done.begin(), done.end(), e->device) != done.end()) return;
        done.push_back(e->device);
    }
    hipFuncAttributes at;
#define TC_TOUCH(...) (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&__VA_ARGS__))
    // (once per process and device: both forms of the ranges -- the next engine may be of the other kind)
    TC_TOUCH(rp::k_tile_part<rp::PART_PLAIN, false>);
    TC_TOUCH(rp::k_tile_part<rp::PART_RANK, false>);
    TC_TOUCH(rs::k_finish<false>);
    TC_TOUCH(rp::k_tile_part<rp::PART_PLAIN, true>);
    TC_TOUCH(rp::k_tile_part<rp::PART_RANK, true>);
    TC_TOUCH(rs::k_finish<true>);
    TC_TOUCH(rp::k_hot_install);
    TC_TOUCH(rs::k_hist<rs::HIST_THREADS>);
    TC_TOUCH(rs::k_onesweep<8, true>);
    TC_TOUCH(rs::k_onesweep<8, false>);
    TC_TOUCH(rs::k_onesweep<16, true>);
    TC_TOUCH(rs::k_onesweep<16, false>);
    TC_TOUCH(rs::k_onesweep<32, true>);
    TC_TOUCH(rs::k_onesweep<32, false>);
    TC_TOUCH(mk::k_heavy_publish);
    TC_TOUCH(k_eval_lean_hot<1, true>);
    TC_TOUCH(k_eval_lean_hot<2, true>);
    TC_TOUCH(k_eval_lean_hot<4, true>);
    TC_TOUCH(k_eval_lean_hot<1, false>);
    TC_TOUCH(k_eval_lean_hot<2, false>);
    TC_TOUCH(k_eval_lean_hot<4, false>);
    TC_TOUCH(k_eval_sorted_lean<1, true>);
    TC_TOUCH(k_eval_sorted_lean<2, true>);
    TC_TOUCH(k_eval_sorted_lean<4, true>);
    TC_TOUCH(k_eval_sorted_lean<1, false>);
    TC_TOUCH(k_eval_sorted_lean<2, false>);
    TC_TOUCH(k_eval_sorted_lean<4, false>);
    TC_TOUCH(k_eval_general<true, true>);
    TC_TOUCH(k_eval_general<false, true>);
    TC_TOUCH(k_eval_general<false, false>);
#undef TC_TOUCH
    (void)hipGetLastError();
}

// ---- bucket path (bucket_path.hpp) -------------------------------------------------------------------
// partition of the batch by key range, on stream `s` (k_tile_hist, k_bucket_scan, k_scatter)
// (false: a launch was rejected -- the caller must not enqueue the bucket evaluation behind this partition)
static bool bucket_partition(tc_engine* e, tc_engine::SortSet& ss, hipStream_t s, const uint32_t* d_slot, uint32_t n) {
    const bp::Work& w = ss.bpw;
    const uint32_t tiles = bp::tiles_of(n), cap = (uint32_t)e->capacity;
    prof_begin(e, TC_STAGE_BUCKET_HIST, s);
    hipLaunchKernelGGL(bp::k_tile_hist, dim3(tiles), dim3(bp::TILE_THREADS), w.nbk * sizeof(uint32_t), s, d_slot, n, cap, w);
    prof_end(e, s);
    prof_begin(e, TC_STAGE_BUCKET_SCAN, s);
    hipLaunchKernelGGL(bp::k_bucket_scan, dim3(bp::scan_blocks(w.nbk)), dim3(bp::SCAN_THREADS), 0, s, w, tiles);
    prof_end(e, s);
    prof_begin(e, TC_STAGE_BUCKET_SCATTER, s);
    hipLaunchKernelGGL(bp::k_scatter, dim3(tiles), dim3(bp::TILE_THREADS), bp::scatter_lds_bytes(w.nbk), s, d_slot, n, cap, w);
    prof_end(e, s);
    return hipGetLastError() == hipSuccess;
}

// evaluation of the partitioned batch: one wave per bucket (k_bucket_eval)
static void bucket_eval(tc_engine* e, tc_engine::SortSet& ss, hipStream_t s, const Params& p, bool full) {
    const bp::Work& w = ss.bpw;
    const bool by_slot = (p.flags & F_REGISTERED) && !(p.flags & F_UNIFORM_CLASS);
    const bool fixed = (p.flags & F_FIXED) != 0;
    const dim3 grid(w.nbk), block(64);
    const size_t lds = bp::eval_lds_bytes(w.lb);
    prof_begin(e, TC_STAGE_BUCKET_EVAL, s);
#define TC_BEV(FU, FI, BS) \
    hipLaunchKernelGGL((bp::k_bucket_eval<FU, FI, BS>), grid, block, lds, s, p, w, e->bp_park)
    switch ((full ? 4 : 0) | (fixed ? 2 : 0) | (by_slot ? 1 : 0)) {
    case 0: TC_BEV(false, false, false); break;
    case 1: TC_BEV(false, false, true); break;
    case 2: TC_BEV(false, true, false); break;
    case 3: TC_BEV(false, true, true); break;
    case 4: TC_BEV(true, false, false); break;
    case 5: TC_BEV(true, false, true); break;
    case 6: TC_BEV(true, true, false); break;
    default: TC_BEV(true, true, true); break;
    }
#undef TC_BEV
    prof_end(e, s);
}

// host arrays -> the scratch set's staging columns, enqueued on `st`; `p` gets the device pointers
static int stage_host_inputs(tc_engine* e, tc_engine::SortSet& ss, const HostIn& hin, uint32_t n, hipStream_t st, Params& p,
                             const uint32_t** d_slot) {
    const uint64_t mb = e->max_batch;
    // A SMALL batch (up to async_copy_kernel_n requests: what an actor's queue yields) is bound by the number of things enqueued
    // for it, not by the link: its columns cross in one copy launch when they are pinned (stage_in_multi).  A large one goes
    // through the SDMA engines, a copy each: they do not take the link's latency out on the kernels running beside them.
    // Measured (tools/abi_probe.py, ring of 4 pinned sets, us per reference-shaped call, copies / one launch): 4 Ki requests 164-168 /
    // 53-55, 16 Ki 72-171 / 64-65, 32 Ki 186-285 / 73-74 (the copies' times jump about: the runtime's bookkeeping for a pinned
    // source), 64 Ki 117 / 126 and in bench.py's leg 112-129 / 148: one launch up to 32 Ki.
    const bool one_launch = n <= e->async_copy_kernel_n;
    const void* c_src[6];
    void* c_dst[6];
    size_t c_bytes[6];
    uint32_t c_n = 0;
    if (hin.slot) { // (key batches arrive with their slots already resolved on the device)
        if (!ss.h_slot) TC_HIP(e, hipMalloc(&ss.h_slot, mb * sizeof(uint32_t)));
        if (one_launch) c_src[c_n] = hin.slot, c_dst[c_n] = ss.h_slot, c_bytes[c_n] = (size_t)n * sizeof(uint32_t), ++c_n;
        else TC_HIP(e, copy_async(e, ss.h_slot, hin.slot, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st)); // SDMA when pinned
        *d_slot = ss.h_slot;
        p.slot = ss.h_slot;
    }
    const int64_t** dst[5] = {&p.burst, &p.count, &p.period, &p.q, &p.now};
    for (int j = 0; j < 5; ++j) {
        if (!hin.col[j]) continue;
        if (!ss.h_in[j]) TC_HIP(e, hipMalloc(&ss.h_in[j], mb * sizeof(int64_t)));
        if (one_launch) c_src[c_n] = hin.col[j], c_dst[c_n] = ss.h_in[j], c_bytes[c_n] = (size_t)n * sizeof(int64_t), ++c_n;
        else TC_HIP(e, copy_async(e, ss.h_in[j], hin.col[j], (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
        *dst[j] = ss.h_in[j];
    }
    if (c_n) TC_TRY(stage_in_multi(e, c_src, c_dst, c_bytes, c_n, st));
    if (hin.plan_id) { // TC_B_PLAN_DICT: the compact columns cross PCIe, the wide ones are made here
        for (int j = 0; j < (hin.q32 ? 4 : 3); ++j)
            if (!ss.h_in[j]) TC_HIP(e, hipMalloc(&ss.h_in[j], mb * sizeof(int64_t)));
        TC_TRY(stage_compact(e, hin.plan_dict, hin.n_plans, hin.plan_id, hin.q32, n, st, ss.h_dict, ss.h_plan_id, ss.h_q32, ss.h_in));
        p.burst = ss.h_in[0], p.count = ss.h_in[1], p.period = ss.h_in[2];
        if (hin.q32) p.q = ss.h_in[3];
    }
    return TC_E_OK;
}

// the pieces of a segmented slot column -> the scratch set's staging column, on stream `st`
static int gather_segments(tc_engine* e, tc_engine::SortSet& ss, const tc_batch& b, uint32_t n, hipStream_t st, Params& p, const uint32_t** d_slot) {
    if (!ss.h_slot) TC_HIP(e, hipMalloc(&ss.h_slot, e->max_batch * sizeof(uint32_t)));
    mk::Segments sg;
    memset(&sg, 0, sizeof sg);
    sg.n = b.n_segments;
    uint32_t at = 0;
    for (uint32_t i = 0; i < b.n_segments; ++i) {
        sg.ptr[i] = b.seg_slot[i];
        sg.start[i] = at;
        at += b.seg_n[i];
    }
    sg.start[b.n_segments] = at; // (== n: checked by the caller)
    hipLaunchKernelGGL(mk::k_concat, dim3(std::min<uint32_t>(nblocks(n), 1024u)), dim3(BLOCK), 0, st, sg, ss.h_slot);
    *d_slot = ss.h_slot;
    p.slot = ss.h_slot;
    return TC_E_OK;
}

// all pointers in `b` are device pointers here (hin: the inputs are host arrays still to be staged)
int run_slots_device(tc_engine* e, const tc_batch& b, const HostIn* hin) {
    const uint32_t n = (uint32_t)b.n;
    Params p;
    p.n = n;
    p.flags = 0;
    p.slot = b.slot;
    p.burst = b.max_burst;
    p.count = b.count_per_period;
    p.period = b.period;
    p.q = b.quantity;
    p.now = b.now_ns;
    p.burst_s = b.max_burst_scalar;
    p.count_s = b.count_per_period_scalar;
    p.period_s = b.period_scalar;
    p.q_s = b.quantity_scalar;
    p.now_s = b.now_ns_scalar;
    p.allowed = b.allowed ? b.allowed : (b.allowed_bits ? e->allowed_tmp : nullptr);
    p.limit = b.limit;
    p.remaining = b.remaining;
    p.reset = b.reset_after_ns;
    p.retry = b.retry_after_ns;
    p.status = b.status;
    p.result4 = b.result4;
    p.decisions = b.decisions;
    p.order = (b.flags & TC_B_GROUPED_OUTPUT) ? b.order : nullptr;
    p.cells = e->cells;
    p.tat8 = e->tat8;
    if (e->fixed) p.flags |= F_FIXED;
    if (e->debug_nostore) p.flags |= F_DEBUG_NOSTORE;
    if (!e->general_earlier) p.flags |= F_NO_EARLIER;
    if (e->general_runs) p.flags |= F_GENERAL_RUNS;
    if (e->debug_break_wait) {
        p.flags |= F_DEBUG_NO_ANNOUNCE;
        e->debug_break_wait = false; // one batch
    }
    p.rate_id = e->rate_id;
    p.classes = e->classes;
    p.uniform_class = e->uniform_id;
    p.denied = e->denied;
    p.row_bits = nullptr;
    bool notes = false; // round 6: the evaluation notes its heavy runs for the host's hot list (set below, where the batch's kind is known)
    p.capacity = e->capacity;
    p.counters = e->counters;
    if (b.flags & TC_B_REGISTERED_PARAMS) {
        p.flags |= F_REGISTERED;
        if (e->uniform_id) p.flags |= F_UNIFORM_CLASS;
    }
    const bool full = p.remaining || p.reset || p.retry || p.result4 || p.decisions;
    const dim3 grid(nblocks(n)), block(BLOCK);
    hipStream_t s = cur_stream(e);
    bool bits_in_kernel = false;

    if (b.flags & TC_B_UNIQUE_SLOTS) {
        e->last_grouping_path = 4u;
        if (hin) {
            // (a set's staging columns are only read by work that this stream has already been ordered behind)
            tc_engine::SortSet& ss = e->sets[e->next_set];
            e->next_set = (e->next_set + 1) % e->depth;
            if (ss.in_use) TC_HIP(e, hipStreamWaitEvent(s, ss.consumed, 0));
            const uint32_t* d_slot = nullptr;
            TC_TRY(stage_host_inputs(e, ss, *hin, n, s, p, &d_slot));
        }
        prof_begin(e, TC_STAGE_EVAL, s);
        if (full) hipLaunchKernelGGL(k_eval_unique<true>, grid, block, 0, s, p);
        else hipLaunchKernelGGL(k_eval_unique<false>, grid, block, 0, s, p);
        prof_end(e, s);
        if (hin) {
            tc_engine::SortSet& ss = e->sets[(e->next_set + e->depth - 1) % e->depth];
            TC_HIP(e, hipEventRecord(ss.consumed, s));
            ss.in_use = true;
        }
    } else {
        // grouping: on the set's auxiliary stream when the caller vouches for the inputs
        // (overlaps with the evaluation of earlier batches), else in order on `s`
        tc_engine::SortSet& ss = e->sets[e->next_set];
        e->next_set = (e->next_set + 1) % e->depth;
        bool piped = (b.flags & TC_B_INPUTS_READY) != 0;
        if (piped) {
            int rc = ensure_side_streams(e);
            if (rc != TC_E_OK) return rc;
            piped = e->n_aux != 0; // no free hardware queue: in order on the main stream
            preload_pipelined_kernels(e);
        }
        // (the per-request columns of a TC_B_ASYNC host batch are still to be staged: they are in `hin`)
        auto column = [&](const int64_t* dev, int j) {
            return dev != nullptr || (hin && (hin->col[j] != nullptr || (hin->plan_id && j < 3) || (hin->q32 && j == 3)));
        };
        const bool params_by_slot = (b.flags & TC_B_REGISTERED_PARAMS) || (!column(p.burst, 0) && !column(p.count, 1) && !column(p.period, 2));
        const bool uniform = !column(p.q, 3) && !column(p.now, 4) && params_by_slot;
        // direct: every run is regular whatever the cells hold: owners store directly, no commit launch
        const bool direct = uniform && all_runs_regular(e, b, p);
        // Uniform batches whose runs are all regular, running IN ORDER on the engine's stream, are enqueued on
        // BOTH grouping paths: the partition by key range (bucket_path.hpp) and, gated behind it, the sort.  The
        // partition publishes its largest bucket; if that is longer than bp_skew (a skewed stream) the bucket
        // kernels leave at once and the sort path runs, otherwise the other way round -- decided on the
        // device, batch by batch (the host may be hundreds of batches ahead).  Pipelined batches are sorted:
        // beside the evaluation of earlier batches the partition's scattered 4-byte writes cost more than
        // the sort's three coalesced passes (69-86 vs 66 us per 1 Mi batch).
        // (TC_B_GROUPED_OUTPUT promises rows grouped by key: that is the sorted order)
        // A skewed batch pays for the partition and then takes the sort path anyway.  The host cannot wait for the
        // gate, but the device mirrors it into pinned memory: when a recent batch tripped it, the next BP_BACKOFF
        // batches are not partitioned at all (the sort path alone is always correct), then the path is tried again.
        // (round 4: a batch the range path takes is grouped by it alone, in order as well: two grouping launches and the
        // evaluation, nothing enqueued twice)
        // (the hot form is the lean kernel's: its hot role ranks the hot slots' requests where they stand -- eval_kernels.hpp; per-slot
        // denial counters want a slot's requests side by side, grouped rows want them sorted: the LSD passes)
        const bool rank_ok = e->hot.on && e->hot.rank_on && uniform && direct && lean_applies(e, full, direct, piped, p) && !p.denied && !p.order;
        // Only the batches that could take the hot form pay for the notes it is chosen by (a general Zipf batch lost 7 us to them
        // and to the copies for the host: profiles/r06_v15_hot_forms_driver_ab.txt); batches too small to care are left out.
        notes = rank_ok && e->hot.notes_dev != nullptr && n >= 16384u;
        if (notes) {
            p.heavy = e->hot.notes_dev;
            p.heavy_min = e->hot.heavy_min;
            p.heavy_tag = (uint32_t)(e->hot.evals & ((1u << ev::HEAVY_TAG_BITS) - 1u));
        }
        hot_refresh(e, notes);
        const int ranged = range_applies(e, n, piped, rank_ok);
        e->hot.he_dev = nullptr;
        // The range hint is written by the grouping kernels of the sort paths (k_hist's range row, k_finish).  An in-order batch
        // on the bucket path leaves none: an engine that only ever sees in-order batches would stay on the bucket path for
        // good -- 84 us per 1 Mi batch where the range path takes 63, found by tools/batch_sizes.py; bench.py's in-order
        // figures had pipelined batches before them -- and a stream that turns uniform would never be noticed.  So the first
        // in-order batch of an engine, and every 32nd that the range path turned down, is grouped by the sort path alone,
        // which looks.
        bool look = false;
        if (!ranged && !piped && e->range_ok && (e->range_mode >= 2) && n >= 256u && n <= e->range_max_n)
            look = (*(volatile unsigned long long*)e->range_hint_host >> 32) == 0ull || (++e->range_relook & 31u) == 0u;
        bool eligible = !ranged && !look && direct && e->bp_ok && n >= e->bp_min_n && n <= e->bp_max_n && !p.order && (!piped || e->bp_piped);
        if (eligible && e->bp_gate_host && e->bp_backoff_len) {
            const uint32_t seen = *(volatile uint32_t*)e->bp_gate_host;
            if (seen > e->bp_skew && e->bp_backoff == 0) {
                e->bp_backoff = e->bp_backoff_len;
                *(volatile uint32_t*)e->bp_gate_host = 0; // (consumed; the next partition that runs writes a fresh one)
            }
            if (e->bp_backoff) {
                --e->bp_backoff;
                eligible = false;
            }
        }
        const bool bucketed = eligible;
        e->last_grouping_path = ranged == GROUP_RANGE_RANK ? 5u : (ranged ? 1u : (bucketed ? 3u : 2u)); // (bucketed: decided on the device in the end -- tc_engine_info says what was enqueued)
        // Grouped rows + a bitmask of them, every run regular: the evaluation's waves hold 64 consecutive rows each and
        // pack their decisions with one ballot (no byte column, no k_pack_bits launch).
        if (b.allowed_bits && p.order && direct) {
            p.row_bits = b.allowed_bits;
            bits_in_kernel = true;
            if (!b.allowed) p.allowed = nullptr;
        }
        const uint32_t* gate = bucketed ? ss.bpw.maxb : nullptr;
        const uint64_t* sorted;
        const uint32_t* d_slot = b.slot;
        if (!hin && !b.n_segments) {
            ss.readers[ss.next_reader % 4].ptr = b.slot;
            ss.readers[ss.next_reader % 4].n = n;
            ss.next_reader++;
        }
        if (piped) {
            hipStream_t ax = e->aux[e->next_aux];
            e->next_aux = (e->next_aux + 1) % e->n_aux;
            if (ss.in_use) TC_HIP(e, hipStreamWaitEvent(ax, ss.consumed, 0)); // the evaluation that read this set is done
            if (ss.publish_due) { // ... and the notes it took go to the host's hot list
                hot_publish(e, ax, ss.publish_due);
                ss.publish_due = 0ull;
            }
            // (the request columns do not depend on the key stage: their transfer goes out before the wait for it -- round 5: it
            // used to sit behind that wait, so that a chunk's columns only started to cross PCIe once its keys were resolved)
            if (hin) TC_TRY(stage_host_inputs(e, ss, *hin, n, ax, p, &d_slot)); // PCIe transfer overlaps earlier evaluations
            if (e->wait_before_sort) TC_HIP(e, hipStreamWaitEvent(ax, e->wait_before_sort, 0)); // its key stage
            if (b.n_segments) TC_TRY(gather_segments(e, ss, b, n, ax, p, &d_slot));
            if (bucketed && !bucket_partition(e, ss, ax, d_slot, n)) return fail(e, TC_E_HIP, "bucket partition launch failed");
            const bool ride = e->stop_events && !e->prof_on; // `sorted` rides on the last pass's own completion signal
            // TC_B_OUTPUTS_IDLE: nothing enqueued earlier touches this call's `allowed` bytes, so they are preset here, on
            // the grouping stream, to what most decisions of a recent batch were, and the evaluation only stores the
            // others -- 1 Mi one-byte stores scattered over the batch were 6 of its 53 us
            uint8_t* fill = nullptr;
            uint32_t fill_value = 0;
            const bool only_allowed = !full && p.allowed && !p.status && !p.limit && !p.order && !p.row_bits && p.allowed == b.allowed;
            if ((b.flags & TC_B_OUTPUTS_IDLE) && e->prefill_on && !hin && !bucketed && only_allowed &&
                (uniform ? lean_applies(e, full, direct, true, p) : true)) {
                fill = b.allowed;
                fill_value = *(volatile uint32_t*)e->fill_hint_host & 1u;
                p.flags |= fill_value ? F_PREFILL1 : F_PREFILL0;
            }
            sorted = sort_by_slot(e, ss, ax, d_slot, n, true, ranged, gate, e->bp_skew, ride ? ss.sorted : nullptr, fill, fill_value);
            if (!ride) TC_HIP(e, hipEventRecord(ss.sorted, ax));
            TC_HIP(e, hipStreamWaitEvent(s, ss.sorted, 0));
            ss.grouped_aside = true;
        } else {
            // everything that used this set earlier is ordered before us on `s`: evaluations ran on `s`,
            // and every auxiliary sort was joined into `s` before its evaluation
            if (hin) TC_TRY(stage_host_inputs(e, ss, *hin, n, s, p, &d_slot));
            if (b.n_segments) TC_TRY(gather_segments(e, ss, b, n, s, p, &d_slot));
            ss.grouped_aside = false;
            if (bucketed && !bucket_partition(e, ss, s, d_slot, n)) return fail(e, TC_E_HIP, "bucket partition launch failed");
            sorted = sort_by_slot(e, ss, s, d_slot, n, false, ranged, gate, e->bp_skew);
        }
        if (bucketed) bucket_eval(e, ss, s, p, full);
        prof_begin_m(e, TC_STAGE_EVAL, s);
        bool consumed_rides = false;
        if (uniform) {
            uint32_t seq = 0u;
            if (direct) {
                if (++e->loaded_seq == 0u) e->loaded_seq = 1u;
                seq = e->loaded_seq;
            }
            // (direct: the evaluation is the last reader of the set -- `consumed` can ride on its completion signal)
            consumed_rides = direct && e->stop_events && !e->prof_on;
            const bool rankm = ranged == GROUP_RANGE_RANK;
            static_assert(rp::PT_TILE == 1u << 12 && rp::HOT_MAX == HOT_IDS_MAX, "the hot role: a tile is 4 096 requests, its LDS arrays hold every hot id");
            e->hot.he_dev = rankm ? ss.hot_eval : nullptr; // (the set's record of where the partition left its notes: device memory)
            // (slim: the stream looked skewed to the range hint -- not merely "no hint yet" or a batch too large for the path)
            // (rank form: what is left in the sorted part is the stream's uniform tail, and the hot role's blocks want the CU's eight
            // block slots beside it: 2 positions per lane -- 36.6 us per Zipf batch against 39.2 with 4, profiles/r06_v12_zipf_items_ab.txt)
            const bool slim = piped && e->range_ok && (!ranged && (*(volatile unsigned long long*)e->range_hint_host >> 32) != 0ull && n <= e->range_max_n);
            launch_eval_sorted(e, full, direct, piped, n, s, p, sorted, seq, gate, e->bp_skew, consumed_rides ? ss.consumed : nullptr, slim);
            prof_end_m(e, s);
            e->hot.he_dev = nullptr;
            if (!direct) {
                prof_begin(e, TC_STAGE_COMMIT, s);
                hipLaunchKernelGGL(k_commit_list, dim3(64), block, 0, s, e->pend, e->pend_count, e->cells, e->tat8);
                prof_end(e, s);
            }
        } else {
            if (++e->chain_seq >= 0x7FFFFFFFu) e->chain_seq = 1u; // 0 = "never written"; 31 bits: the spec word carries a flag
            // (the evaluation is the last reader of the set: `consumed` rides on its completion signal)
            consumed_rides = e->stop_events && !e->prof_on;
            hipEvent_t stop = consumed_rides ? ss.consumed : nullptr;
            // (decisions only, most decisions of a recent batch denied: the variant without the allowed-runs rule -- see the kernel)
            const bool drained = !full && e->general_lean && (*(volatile uint32_t*)e->fill_hint_host & 1u) == 0u;
            if (full) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_general<true, true>), grid, block, 0, s, p, sorted, e->chain, e->chain_seq, e->fill_hint_dev);
            else if (drained) TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_general<false, false>), grid, block, 0, s, p, sorted, e->chain, e->chain_seq, e->fill_hint_dev);
            else TC_LAUNCH_T(e, TC_STAGE_EVAL, stop, (k_eval_general<false, true>), grid, block, 0, s, p, sorted, e->chain, e->chain_seq, e->fill_hint_dev);
            prof_end_m(e, s);
        }
        e->wait_before_sort = nullptr;
        if (notes) {
            // a copy of the notes for the host behind the first few evaluations, then behind every eighth (one block, ~3 us of
            // the engine's stream; the host never waits for it)
            const uint64_t k = ++e->hot.evals;
            if (k <= 4 || ((k & 7u) == 0u && (e->hot.stable_looks < 4u || (k & 31u) == 0u))) { // (a list that has settled is looked at every 32nd batch)
                const unsigned long long seq = ((unsigned long long)(++e->hot.published) << ev::HEAVY_TAG_BITS) | p.heavy_tag;
                // (A pipelined stream past its first batches: on the grouping stream of the batch that next uses this set, which waits
                // for this evaluation anyway -- a few batches later, a copy of whole evaluations as before, and not a kernel more on the
                // stream every step waits for.  The first four: at once, the list they are for is what a skewed stream is waiting for.)
                if (piped && k > 4) ss.publish_due = seq;
                else hot_publish(e, s, seq);
            }
        }
        // a later TC_B_INPUTS_READY batch may re-sort into this set on the auxiliary stream
        if (!consumed_rides) TC_HIP(e, hipEventRecord(ss.consumed, s));
        ss.in_use = true;
    }
    if (b.allowed_bits && !bits_in_kernel) {
        prof_begin(e, TC_STAGE_PACK, s);
        hipLaunchKernelGGL(k_pack_bits, grid, block, 0, s, p.allowed, n, b.allowed_bits);
        prof_end(e, s);
    }
    TC_HIP(e, hipGetLastError());
    e->batches++; // (a batch that failed on the way here was not applied and is not counted)
    return TC_E_OK;
}

// Host-pointer batch whose slot column is already in e->stage.slot: stage the
// other inputs, run, copy the outputs back, synchronise.
// key_error_flag: a key batch -- "did a key fail to get a slot" comes back with the results, behind the same wait (it used to be
// a copy and a wait of its own behind this one: ~25 us of a 4 Ki-request call's 200)
// d_slot: the slot column where it already is in device memory (a key batch's resolved slots), else e->stage.slot
int run_slots_host_staged(tc_engine* e, const tc_batch& b, uint32_t* key_error_flag, const uint32_t* d_slot, bool columns_staged) {
    const uint64_t n = b.n;
    hipStream_t s = cur_stream(e);
    tc_batch d = b;
    d.flags |= TC_B_DEVICE_PTRS;
    d.flags &= ~(TC_B_INPUTS_READY | TC_B_ASYNC);
    d.slot = d_slot ? d_slot : e->stage.slot;
    d.key_bytes = nullptr;
    d.key_off = nullptr;
    const int64_t* hin[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    const int64_t** din[5] = {&d.max_burst, &d.count_per_period, &d.period, &d.quantity, &d.now_ns};
    const void* c_src[5];
    void* c_dst[5];
    size_t c_bytes[5];
    uint32_t c_n = 0;
    for (int j = 0; j < 5; ++j) {
        if (hin[j]) {
            TC_TRY(stage_need(e, e->stage.in[j], e->max_batch));
            c_src[c_n] = hin[j], c_dst[c_n] = e->stage.in[j], c_bytes[c_n] = n * sizeof(int64_t);
            ++c_n;
            *din[j] = e->stage.in[j];
        }
    }
    // (the request columns: one launch from pinned arrays, else a copy each -- a key batch has sent them with its keys already)
    if (!columns_staged) TC_TRY(stage_in_multi(e, c_src, c_dst, c_bytes, c_n, s));
    if (b.flags & TC_B_PLAN_DICT) { // the compact columns cross PCIe, the wide ones are made on the device
        for (int j = 0; j < (b.quantity32 ? 4 : 3); ++j) TC_TRY(stage_need(e, e->stage.in[j], e->max_batch));
        TC_TRY(stage_compact(e, b.plan_dict, b.n_plans, b.plan_id, b.quantity32, (uint32_t)n, s, e->stage.dict, e->stage.plan_id, e->stage.q32, e->stage.in));
        d.max_burst = e->stage.in[0], d.count_per_period = e->stage.in[1], d.period = e->stage.in[2];
        if (b.quantity32) d.quantity = e->stage.in[3];
        d.flags &= ~TC_B_PLAN_DICT;
        d.plan_dict = nullptr, d.plan_id = nullptr, d.quantity32 = nullptr, d.n_plans = 0;
    }
    TC_TRY(stage_outputs(e, b, d));
    TC_TRY(run_slots_device(e, d));
    // Results back.  Round 5: into pinned arrays ONE copy launch behind the evaluation takes every output and the key-error word
    // with it -- hipMemcpyAsync to pinned memory started 16 us after the evaluation had ended, and the word was a second copy
    // behind the first (25 of a 4 Ki-request call's 103 us on the device, tools/trace_seq.py): the reference-shaped call of
    // 4 Ki / 64 Ki / 256 Ki requests 106 -> 84, 240 -> 201, 563 -> 537 us.
    bool flag_by_kernel = false;
    if (!outputs_back_in_one_launch(e, b, s, key_error_flag != nullptr, &flag_by_kernel)) {
        TC_TRY(copy_outputs_back(e, b, s));
        if (key_error_flag) TC_HIP(e, hipMemcpyAsync(key_error_flag, e->kt.error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    TC_HIP(e, hipStreamSynchronize(s));
    if (key_error_flag && flag_by_kernel) *key_error_flag = *(volatile const uint32_t*)(e->host_results + 192);
    return poisoned(e); // (a synchronous batch whose kernels flagged an invariant fails itself, not the next call)
}

// results back to the caller's arrays behind the evaluation + the batch's completion event
int finish_async(tc_engine* e, const tc_batch& b) {
    hipStream_t s = cur_stream(e);
    TC_TRY(copy_outputs_back(e, b, s, true));
    hipEvent_t ev = nullptr;
    if (!e->async_pool.empty()) {
        ev = e->async_pool.back();
        e->async_pool.pop_back();
    } else {
        TC_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    e->async_done.push_back(ev);
    TC_HIP(e, hipEventRecord(ev, s));
    return TC_E_OK;
}

// TC_B_ASYNC: the host batch's inputs are staged on the stream that groups it (overlapping the evaluation
// of earlier batches), the outputs are copied back behind its evaluation, nothing waits.
static int run_slots_host_async(tc_engine* e, const tc_batch& b) {
    tc_batch d = b;
    d.flags |= TC_B_DEVICE_PTRS | TC_B_INPUTS_READY; // the host arrays are complete at call time by contract
    d.flags &= ~TC_B_ASYNC;
    d.slot = nullptr;
    d.max_burst = d.count_per_period = d.period = d.quantity = d.now_ns = nullptr;
    d.key_bytes = nullptr;
    d.key_off = nullptr;
    HostIn hin;
    hin.slot = b.slot;
    hin.col[0] = b.max_burst, hin.col[1] = b.count_per_period, hin.col[2] = b.period, hin.col[3] = b.quantity, hin.col[4] = b.now_ns;
    if (b.flags & TC_B_PLAN_DICT) {
        hin.plan_dict = b.plan_dict, hin.plan_id = b.plan_id, hin.q32 = b.quantity32, hin.n_plans = b.n_plans;
        d.flags &= ~TC_B_PLAN_DICT;
        d.plan_dict = nullptr, d.plan_id = nullptr, d.quantity32 = nullptr, d.n_plans = 0;
    }
    TC_TRY(stage_outputs(e, b, d));
    TC_TRY(run_slots_device(e, d, &hin));
    return finish_async(e, b);
}

bool host_arrays_pinned(const tc_batch& b) {
    const bool dict = (b.flags & TC_B_PLAN_DICT) != 0;
    const void* arrays[] = {b.slot, b.key_bytes, b.key_off, b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns, b.allowed, b.allowed_bits,
                            b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns, b.status, b.result4, b.decisions,
                            dict ? (const void*)b.plan_dict : nullptr, dict ? (const void*)b.plan_id : nullptr, dict ? (const void*)b.quantity32 : nullptr};
    for (const void* a : arrays)
        if (a && !device_view_of_host(a)) return false;
    return true;
}

void host_sub_batch(const tc_batch& b, uint64_t at, uint64_t cn, tc_batch& c) {
    c = b;
    c.n = cn;
    c.flags |= TC_B_ASYNC;
    if (c.slot) c.slot += at;
    if (c.key_off) c.key_off += at; // (offsets stay absolute into key_bytes: the staging copies the chunk's byte range only)
    if (c.max_burst) c.max_burst += at;
    if (c.count_per_period) c.count_per_period += at;
    if (c.period) c.period += at;
    if (c.quantity) c.quantity += at;
    if (c.now_ns) c.now_ns += at;
    if ((c.flags & TC_B_PLAN_DICT) && c.plan_id) c.plan_id += at;
    if ((c.flags & TC_B_PLAN_DICT) && c.quantity32) c.quantity32 += at;
    if (c.allowed) c.allowed += at;
    if (c.allowed_bits) c.allowed_bits += at / 64;
    if (c.limit) c.limit += at;
    if (c.remaining) c.remaining += at;
    if (c.reset_after_ns) c.reset_after_ns += at;
    if (c.retry_after_ns) c.retry_after_ns += at;
    if (c.status) c.status += at;
    if (c.result4) c.result4 += 4 * at;
    if (c.decisions) c.decisions += at;
}

int wait_own_async(tc_engine* e, size_t mine) {
    if (mine == 0 || e->async_done.empty()) return poisoned(e);
    // completion is in issue order: the newest one done means all of them are (earlier TC_B_ASYNC batches of the caller's stay
    // in the queue -- complete, but the caller's to collect with tc_wait_batches)
    TC_HIP(e, hipEventSynchronize(e->async_done.back()));
    for (size_t k = 0; k < mine && !e->async_done.empty(); ++k) {
        e->async_pool.push_back(e->async_done.back());
        e->async_done.pop_back();
    }
    return poisoned(e);
}

static int run_slots_host_async(tc_engine* e, const tc_batch& b);
// a large synchronous host-pointer slot batch, pipelined in chunks (engine.hpp: HOST_CHUNK_MIN)
static int run_slots_host_chunked(tc_engine* e, const tc_batch& b) {
    size_t mine = 0;
    int rc = TC_E_OK;
    uint64_t at = 0;
    for (const uint64_t cn : host_chunk_plan(e, b.n)) {
        if (rc != TC_E_OK) break;
        tc_batch c;
        host_sub_batch(b, at, cn, c);
        rc = run_slots_host_async(e, c);
        if (rc == TC_E_OK) ++mine;
        else if (at) e->err += " (a later chunk of a pipelined host batch: the " + std::to_string(at) + " requests before it were applied)";
        at += cn;
    }
    e->batches -= mine ? mine - 1 : 0; // one batch, as far as the caller is concerned
    const int wrc = wait_own_async(e, mine);
    return rc != TC_E_OK ? rc : wrc;
}

// ---- small host-pointer batches: one launch (k_small_batch) ----------------------------------------
constexpr size_t SMALL_KEY_BYTES = 128 * 1024; // string mode: batches with a larger key arena take the big pipeline

bool small_batch_applies(const tc_engine* e, const tc_batch& b) {
    if (e->small_off || b.n > (uint64_t)SMALL_MAX) return false;
    if (b.flags & (TC_B_DEVICE_PTRS | TC_B_ASYNC | TC_B_GROUPED_OUTPUT)) return false;
    if (b.allowed_bits) return false; // (packed bits come out of the big pipeline only)
    if (e->key_mode && b.key_off[b.n] > SMALL_KEY_BYTES) return false;
    return true;
}

// Host-pointer batch of at most SMALL_MAX requests, slot or string mode.  Same results as the big pipeline
// (the run walker is the plain sequence); ordering against key stages on the key stream as tc_rate_limit.
int run_small_batch(tc_engine* e, const tc_batch& b) {
    aside_reset(e); // (resolves its keys inside the kernel, on the engine's stream)
    const size_t n = b.n;
    if (!e->small_io) {
        const size_t bytes = 64 + SMALL_KEY_BYTES + (size_t)SMALL_MAX * (4 + 4 + 4 + 5 * 8 + 1 + 1 + 4 * 8 + 32 + 32) + 16 * 16;
        TC_HIP(e, hipHostMalloc((void**)&e->small_io, bytes, hipHostMallocDefault));
        void* dv = nullptr;
        TC_HIP(e, hipHostGetDevicePointer(&dv, e->small_io, 0));
        e->small_io_dev = (uint8_t*)dv;
        e->small_io_bytes = bytes;
    }
    size_t off = 64; // [0, 64): header: word 0 = "a key could not be bound"
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = (off + bytes + 15) & ~(size_t)15;
        return at;
    };
    uint8_t* h = e->small_io;
    uint8_t* d = e->small_io_dev;
    std::memset(h, 0, 64);
    Params p;
    std::memset(&p, 0, sizeof p);
    p.n = (uint32_t)n;
    const uint8_t* d_key_bytes = nullptr;
    const uint32_t* d_key_off = nullptr;
    uint32_t* d_slot_out = nullptr;
    if (e->key_mode) {
        const size_t total = b.key_off[n];
        const size_t o_off = take((n + 1) * 4), o_bytes = take(total + 16);
        e->small_slots_at = take(n * 4);
        d_slot_out = (uint32_t*)(d + e->small_slots_at);
        std::memcpy(h + o_off, b.key_off, (n + 1) * 4);
        if (total) std::memcpy(h + o_bytes, b.key_bytes, total);
        d_key_off = (const uint32_t*)(d + o_off);
        d_key_bytes = d + o_bytes;
    } else {
        const size_t o = take(n * 4);
        std::memcpy(h + o, b.slot, n * 4);
        p.slot = (const uint32_t*)(d + o);
    }
    const int64_t* hin[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    const int64_t** din[5] = {&p.burst, &p.count, &p.period, &p.q, &p.now};
    for (int j = 0; j < 5; ++j) {
        if (!hin[j]) continue;
        const size_t o = take(n * 8);
        std::memcpy(h + o, hin[j], n * 8);
        *din[j] = (const int64_t*)(d + o);
    }
    p.burst_s = b.max_burst_scalar;
    p.count_s = b.count_per_period_scalar;
    p.period_s = b.period_scalar;
    p.q_s = b.quantity_scalar;
    p.now_s = b.now_ns_scalar;
    struct Out {
        void* host;
        size_t at, bytes;
    } outs[9];
    int n_out = 0;
    auto out_col = [&](void* host, size_t bytes) -> uint8_t* {
        if (!host) return nullptr;
        const size_t o = take(bytes);
        outs[n_out++] = Out{host, o, bytes};
        return d + o;
    };
    p.allowed = out_col(b.allowed, n);
    p.status = out_col(b.status, n);
    p.limit = (int64_t*)out_col(b.limit, n * 8);
    p.remaining = (int64_t*)out_col(b.remaining, n * 8);
    p.reset = (int64_t*)out_col(b.reset_after_ns, n * 8);
    p.retry = (int64_t*)out_col(b.retry_after_ns, n * 8);
    p.result4 = (int64_t*)out_col(b.result4, n * 32);
    p.decisions = (tc_decision*)out_col(b.decisions, n * sizeof(tc_decision));
    if (off > e->small_io_bytes) return fail(e, TC_E_INVALID_ARG, "small batch does not fit its staging block");
    p.cells = e->cells;
    p.tat8 = e->tat8;
    if (e->fixed) p.flags |= F_FIXED;
    p.rate_id = e->rate_id;
    p.classes = e->classes;
    p.uniform_class = e->uniform_id;
    p.denied = e->denied;
    p.capacity = e->capacity;
    p.counters = e->counters;
    if (b.flags & TC_B_REGISTERED_PARAMS) {
        p.flags |= F_REGISTERED;
        if (e->uniform_id) p.flags |= F_UNIFORM_CLASS;
    }
    hipStream_t s = cur_stream(e);
    if (e->key_mode && e->k_busy) { // key stages on the key stream come first
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    prof_begin(e, TC_STAGE_EVAL, s);
    hipLaunchKernelGGL(k_small_batch, dim3(1), dim3(SMALL_MAX), 0, s, p, e->kt, e->key_mode ? 1 : 0, d_key_bytes, d_key_off,
                       (uint32_t*)d, e->counters + TC_CNT_KEYS_INSERTED, d_slot_out);
    prof_end(e, s);
    TC_HIP(e, hipGetLastError());
    if (e->key_mode) { // the key table may have changed: later key stages on the key stream wait for this
        TC_HIP(e, hipEventRecord(e->m_done, s));
        e->m_busy = true;
    }
    TC_HIP(e, hipStreamSynchronize(s));
    e->batches++;
    for (int j = 0; j < n_out; ++j) std::memcpy(outs[j].host, h + outs[j].at, outs[j].bytes);
    uint32_t full[2] = {0, 0}; // [0]: in this batch; [1]: in an asynchronous batch before it
    std::memcpy(full, h, sizeof full);
    if (full[0]) TC_HIP(e, hipMemsetAsync(e->kt.error_flag, 0, sizeof(uint32_t), s));
    if (full[0] || full[1])
        return fail(e, TC_E_TABLE_FULL, full[0] ? "key table full: some keys got status Internal (raise capacity or sweep)"
                                                : "key table full in an earlier TC_B_ASYNC batch: some of its keys got status Internal");
    return TC_E_OK;
}

extern "C" int tc_rate_limit_batch_slots(tc_engine* e, const tc_batch* bp) {
    // (callers built against the struct without the segment fields pass its old size: those fields read as zero)
    if (!e || !bp || bp->struct_size < offsetof(tc_batch, n_segments)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    tc_batch b;
    memset(&b, 0, sizeof b);
    memcpy(&b, bp, std::min<size_t>(bp->struct_size, sizeof b));
    if (b.n == 0) return TC_E_OK;
    if (b.n > e->max_batch) return fail(e, TC_E_BATCH_TOO_LARGE, "batch larger than max_batch");
    aside_reset(e); // (a key engine's cells touched by slot: the next sweep goes behind it -- keys.hip: sweep_keys_device)
    const bool segmented = b.n_segments != 0;
    if (segmented) {
        if (!(b.flags & TC_B_DEVICE_PTRS) || (b.flags & (TC_B_UNIQUE_SLOTS | TC_B_ASYNC)))
            return fail(e, TC_E_INVALID_ARG, "a segmented slot column needs TC_B_DEVICE_PTRS (and is neither TC_B_UNIQUE_SLOTS nor TC_B_ASYNC)");
        if (b.n_segments > TC_MAX_SEGMENTS || !b.seg_slot || !b.seg_n) return fail(e, TC_E_INVALID_ARG, "segments: 1..64 pieces, both arrays given");
        uint64_t tot = 0;
        for (uint32_t i = 0; i < b.n_segments; ++i) {
            if (b.seg_n[i] && !b.seg_slot[i]) return fail(e, TC_E_INVALID_ARG, "segments: NULL piece");
            tot += b.seg_n[i];
        }
        if (tot != b.n) return fail(e, TC_E_INVALID_ARG, "segments: the pieces do not add up to n");
    } else if (!b.slot) return fail(e, TC_E_INVALID_ARG, "slot column is NULL");
    if ((b.flags & TC_B_GROUPED_OUTPUT) && !b.order) return fail(e, TC_E_INVALID_ARG, "TC_B_GROUPED_OUTPUT needs `order`");
    if (e->key_mode) return fail(e, TC_E_INVALID_ARG, "key-mode engine: slots are assigned by the key table; use tc_rate_limit_batch_keys");
    if (e->fixed) {
        if (!(b.flags & TC_B_REGISTERED_PARAMS))
            return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: batches use the registered plans (TC_B_REGISTERED_PARAMS)");
    }
    TC_HIP(e, hipSetDevice(e->device));
    int rc;
    const bool dev_ptrs = (b.flags & TC_B_DEVICE_PTRS) != 0;
    if (dev_ptrs && (b.flags & TC_B_ASYNC)) return fail(e, TC_E_INVALID_ARG, "TC_B_ASYNC is for host-pointer batches (device-pointer batches are asynchronous anyway)");
    // TC_B_PLAN_DICT (round 6): a device batch is decoded by a kernel here; a host batch small enough for the one-launch path
    // on the host (its inputs travel inside the launch's pinned block); every other host batch where its columns are staged
    TC_TRY(dict_check(e, b));
    if (!(b.flags & TC_B_PLAN_DICT)) b.plan_dict = nullptr, b.plan_id = nullptr, b.quantity32 = nullptr, b.n_plans = 0;
    else if (dev_ptrs) TC_TRY(dict_expand_device(e, b));
    else if (!(b.flags & TC_B_ASYNC) && b.n <= (uint64_t)SMALL_MAX) dict_expand_on_host(e, b);
    // the engine cleans by itself (tc_set_sweep_policy): maybe_clean_expired in front of the batch (autosweep.hip)
    if (auto_sweep_on(e)) TC_TRY(auto_sweep_before(e, b.n, false, !dev_ptrs || !b.now_ns, (!dev_ptrs && b.now_ns) ? b.now_ns[0] : b.now_ns_scalar));
    if (dev_ptrs) {
        rc = run_slots_device(e, b);
    } else if (b.flags & TC_B_ASYNC) {
        rc = run_slots_host_async(e, b);
    } else if (small_batch_applies(e, b)) {
        rc = run_small_batch(e, b);
    } else if (host_chunking_applies(e, b)) {
        rc = run_slots_host_chunked(e, b);
    } else {
        // host pointers: stage in, run, stage out, synchronise (a small batch in pageable memory: through the engine's pinned block)
        Bounced bo;
        const tc_batch& u = bounce_in(e, b, bo) ? bo.bb : b;
        TC_TRY(stage_need(e, e->stage.slot, e->max_batch));
        TC_HIP(e, copy_async(e, e->stage.slot, u.slot, u.n * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream(e)));
        rc = run_slots_host_staged(e, u);
        if (bo.on && rc == TC_E_OK) bounce_out(bo);
    }
    // fixed layout: once a request has been decided the plans can no longer change (a batch that was rejected, or
    // whose staging failed, applied nothing and seals nothing)
    if (e->fixed && rc == TC_E_OK) e->sealed = true;
    if (rc == TC_E_OK && auto_sweep_on(e))
        rc = auto_sweep_after(e, b.n, false, (dev_ptrs && b.now_ns) ? b.now_ns + (b.n - 1) : nullptr, (!dev_ptrs && b.now_ns) ? b.now_ns[b.n - 1] : b.now_ns_scalar);
    return rc;
}

extern "C" int tc_wait_batches(tc_engine* e, uint32_t max_in_flight) {
    if (!e) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    TC_HIP(e, hipSetDevice(e->device));
    while (e->async_done.size() > max_in_flight) {
        hipEvent_t ev = e->async_done.front();
        TC_HIP(e, hipEventSynchronize(ev));
        e->async_done.pop_front();
        e->async_pool.push_back(ev);
    }
    return poisoned(e);
}

extern "C" void* tc_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void tc_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}
