// keys.hip -- string keys: key stages on the key stream, tc_rate_limit_batch_keys, tc_rate_limit, the `trait Store` shims, key introspection
#include "engine.hpp"

// The runtime builds a kernel's function object when it is first launched (40-150 us on the calling thread: slots.hip,
// preload_pipelined_kernels).  A string-mode server's first sweep -- five kernels nobody has launched yet -- comes in the middle of its
// pipelined key batches; they are looked up once per process and device where the first pipelined key stage is enqueued.
static void preload_sweep_kernels(tc_engine* e) {
    static std::mutex mu;
    static std::vector<int> done;
    {
        std::lock_guard<std::mutex> g(mu);
        if (std::find(done.begin(), done.end(), e->device) != done.end()) return;
        done.push_back(e->device);
    }
    hipFuncAttributes at;
#define TC_TOUCH(...) (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&__VA_ARGS__))
    TC_TOUCH(mk::k_sweep_keys);
    TC_TOUCH(mk::k_sweep_decide);
    TC_TOUCH(mk::k_sweep_tombstones);
    TC_TOUCH(kt::k_table_clear_compact);
    TC_TOUCH(kt::k_table_reinsert);
    TC_TOUCH(mk::k_touch_mark);
    TC_TOUCH(mk::k_sweep_fixup);
#undef TC_TOUCH
}

// keys (device arena) -> out_slot[0..n): found slot, freshly bound slot, or NO_SLOT.
// on_key_stream: issue on the key stream (the caller vouched for the inputs), else on the main stream.
int resolve_keys_device(tc_engine* e, const uint8_t* d_bytes, const uint32_t* d_off, uint32_t n, bool insert,
                               uint32_t* out_slot, bool on_key_stream) {
    hipStream_t s = on_key_stream ? e->key_stream : cur_stream(e);
    if (on_key_stream) {
        if (!e->sweep_kernels_preloaded) {
            e->sweep_kernels_preloaded = true;
            preload_sweep_kernels(e);
        }
        if (e->m_busy) TC_HIP(e, hipStreamWaitEvent(s, e->m_done, 0));
    } else {
        if (e->k_busy) TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        aside_reset(e); // (key work on the engine's stream: the next sweep goes behind it)
    }
    const dim3 grid(nblocks(n)), block(kt::THREADS);
    bool carried = false;
    prof_begin(e, TC_STAGE_HASH, s);
    if (insert) {
        hipLaunchKernelGGL(kt::k_probe<true>, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, e->k_state, e->k_aux, e->k_claim);
        hipLaunchKernelGGL(kt::k_bind, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, (const uint8_t*)e->k_state,
                           (const uint32_t*)e->k_aux, e->k_claim);
        // (the stage's completion event rides on its last kernel: an event record of its own is one more packet on the chain)
        carried = !e->prof_on;
        TC_LAUNCH(carried ? (on_key_stream ? e->k_done : e->m_done) : (hipEvent_t) nullptr, kt::k_follow, grid, block, 0, s, n, out_slot,
                  (const uint8_t*)e->k_state, (const uint32_t*)e->k_aux, e->kt, (const uint32_t*)e->k_claim, grid.x,
                  e->counters + TC_CNT_KEYS_INSERTED);
    } else {
        hipLaunchKernelGGL(kt::k_probe<false>, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, e->k_state,
                           e->k_aux, (uint32_t*)nullptr);
    }
    prof_end(e, s);
    TC_HIP(e, hipGetLastError());
    if (on_key_stream) {
        if (!carried) TC_HIP(e, hipEventRecord(e->k_done, s));
        e->k_busy = true;
        e->m_busy = false;
    } else {
        if (!carried) TC_HIP(e, hipEventRecord(e->m_done, s));
        e->m_busy = true;
        e->k_busy = false;
    }
    return TC_E_OK;
}

// host key arena -> staged on device
// cols (optional): the batch whose request columns travel in the same launch (run_slots_host_staged's staging: e->stage.in[j])
int stage_keys(tc_engine* e, const uint8_t* key_bytes, const uint32_t* key_off, uint64_t n,
                      const uint8_t** d_bytes, const uint32_t** d_off, const tc_batch* cols) {
    const size_t total = key_off[n];
    if (total > e->k_stage_bytes_cap) {
        if (e->k_stage_bytes) (void)hipFree(e->k_stage_bytes);
        e->k_stage_bytes = nullptr;
        const size_t want = std::max<size_t>(total * 2, 1 << 16);
        TC_HIP(e, hipMalloc(&e->k_stage_bytes, want));
        e->k_stage_bytes_cap = want;
    }
    const void* c_src[7] = {key_bytes, key_off};
    void* c_dst[7] = {e->k_stage_bytes, e->k_stage_off};
    size_t c_bytes[7] = {total, (n + 1) * sizeof(uint32_t)};
    uint32_t c_n = 2;
    if (cols) {
        const int64_t* hin[5] = {cols->max_burst, cols->count_per_period, cols->period, cols->quantity, cols->now_ns};
        for (int j = 0; j < 5; ++j) {
            if (!hin[j]) continue;
            TC_TRY(stage_need(e, e->stage.in[j], e->max_batch));
            c_src[c_n] = hin[j], c_dst[c_n] = e->stage.in[j], c_bytes[c_n] = n * sizeof(int64_t);
            ++c_n;
        }
    }
    TC_TRY(stage_in_multi(e, c_src, c_dst, c_bytes, c_n, cur_stream(e)));
    *d_bytes = e->k_stage_bytes ? e->k_stage_bytes : (const uint8_t*)e->k_stage_off;
    *d_off = e->k_stage_off;
    return TC_E_OK;
}

// one key (host bytes) -> slot on the host; insert binds a fresh slot if unseen
int resolve_one_key(tc_engine* e, const uint8_t* key, size_t key_len, bool insert, uint32_t* slot) {
    if (key_len > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "key too long");
    const uint32_t off[2] = {0u, (uint32_t)key_len};
    const uint8_t* d_bytes;
    const uint32_t* d_off;
    int rc = stage_keys(e, key, off, 1, &d_bytes, &d_off);
    if (rc != TC_E_OK) return rc;
    rc = resolve_keys_device(e, d_bytes, d_off, 1, insert, e->one_slot, false); // (the slot lands in pinned memory: no copy behind it)
    if (rc != TC_E_OK) return rc;
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    *slot = *(volatile const uint32_t*)e->one_slot_host;
    return TC_E_OK;
}

// a key-mode sweep on the engine's stream (maintenance_kernels.hpp: k_sweep_keys, k_sweep_decide, k_sweep_tombstones), then the
// rare work: the table rebuilt once tombstones + the keys just unbound exceed 1/4 of it, the overflow arena (keys longer than
// 112 bytes) compacted once more than half of it is handed out -- both decided on the device, near-empty launches otherwise
// `aside` (round 6, maint.hip decides): the newest work on the table was a pipelined key batch's key stage, and the last `aside_streak`
// things on the engine's stream are the evaluations of such batches, nothing else.  The sweep then goes onto the KEY stream: wait for
// the evaluation in front of the newest TWO (done, or nearly: the evaluations run one to two batches behind the key stages), mark the
// slots those two batches asked for, scan / decide / tombstones (/ rebuild) leaving the marked slots alone -- beside those batches'
// grouping and evaluation, which touch nothing but the marked slots' cells --, then wait for the newest evaluation and look at the
// marked slots (mk::k_sweep_fixup).  The key stages of later batches follow on the same stream.  configs[4]: the key stream used to
// idle 240-280 us around every sweep, 120-130 of them waiting for the grouping and the evaluations in front of it.
int sweep_keys_device(tc_engine* e, int64_t now_ns, unsigned long long* removed_scratch, bool aside) {
    kt::Table& t = e->kt;
    hipStream_t s = aside ? e->key_stream : cur_stream(e);
    uint32_t* flag = t.error_flag + 1; // spare word of the table's misc block
    int* top_save = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(t.overflow_used) + 48);
    unsigned long long* oflag = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(t.overflow_used) + 32);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(nblocks(e->capacity), mk::SWEEP_GRID);
    const dim3 block(kt::THREADS);
    tc_engine::SortSet* last = aside ? &e->sets[e->aside_set] : nullptr;
    if (aside) {
        // the newest `marked` batches' slots are marked (2 of a streak of 3 or more); the evaluation in front of them is waited for
        const uint32_t marked = std::min<uint32_t>(std::min<uint32_t>(e->aside_streak - 1u, e->depth - 1u), 2u);
        tc_engine::SortSet& before = e->sets[(e->aside_set + e->depth - marked) % e->depth];
        if (before.in_use) TC_HIP(e, hipStreamWaitEvent(s, before.consumed, 0));
        for (uint32_t k = 0; k < marked; ++k) {
            const tc_engine::SortSet& ms = e->sets[(e->aside_set + e->depth - k) % e->depth];
            hipLaunchKernelGGL(mk::k_touch_mark, dim3(nblocks(ms.k_n)), dim3(BLOCK), 0, s, (const uint32_t*)ms.k_slot, ms.k_n, (uint32_t)e->capacity, e->touched);
        }
    }
    hipLaunchKernelGGL(mk::k_sweep_keys, dim3(blocks), dim3(BLOCK), 0, s, e->cells, t, now_ns, e->sweep_work, e->denied, aside ? (const uint8_t*)e->touched : (const uint8_t*)nullptr);
    hipLaunchKernelGGL(mk::k_sweep_decide, dim3(1), dim3(mk::DECIDE_THREADS), 0, s, t, e->sweep_work, blocks, top_save, flag, oflag,
                       removed_scratch, e->counters);
    hipLaunchKernelGGL(mk::k_sweep_tombstones, dim3(blocks), dim3(BLOCK), 0, s, t, e->sweep_work, (const int*)top_save, (const uint32_t*)flag,
                       e->spread_free ? 1u : 0u);
    hipLaunchKernelGGL(kt::k_table_clear_compact, dim3(std::min<uint64_t>(nblocks(t.nb_mask + 1), 4096)), block, 0, s, t, (const uint32_t*)flag, oflag);
    if (aside) {
        hipLaunchKernelGGL(kt::k_table_reinsert, dim3(std::min<uint64_t>((t.capacity + kt::RE_TILE - 1) / kt::RE_TILE, 2048)), block, 0, s, t, (const uint32_t*)flag,
                           (const unsigned long long*)oflag);
        TC_HIP(e, hipStreamWaitEvent(s, last->consumed, 0)); // the newest batch's evaluation
        // (the sweep's completion event -- work on the engine's stream that touches the table waits for it -- rides on its last kernel)
        TC_LAUNCH(e->k_done, mk::k_sweep_fixup, dim3(std::min<uint64_t>(nblocks((e->capacity + 15) / 16), 4096)), dim3(BLOCK), 0, s, e->cells, t, now_ns, e->touched,
                  e->denied, removed_scratch, e->counters);
        TC_HIP(e, hipGetLastError());
        e->k_busy = true;
        e->m_busy = false;
        e->sweeps_aside++;
        aside_swept(e);
        return TC_E_OK;
    }
    // (the sweep's completion event -- later key stages on the key stream wait for it -- rides on its last kernel)
    TC_LAUNCH(e->m_done, kt::k_table_reinsert, dim3(std::min<uint64_t>((t.capacity + kt::RE_TILE - 1) / kt::RE_TILE, 2048)), block, 0, s, t, (const uint32_t*)flag,
              (const unsigned long long*)oflag);
    TC_HIP(e, hipGetLastError());
    e->m_busy = true;
    return TC_E_OK;
}

// did any key of the batches since the last check fail to get a slot?
static int check_key_errors(tc_engine* e) {
    uint32_t flag = 0;
    TC_HIP(e, hipMemcpyAsync(&flag, e->kt.error_flag, sizeof flag, hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    if (flag) {
        TC_HIP(e, hipMemsetAsync(e->kt.error_flag, 0, sizeof flag, cur_stream(e)));
        return fail(e, TC_E_TABLE_FULL, "key table full: some keys got status Internal (raise capacity or sweep)");
    }
    return TC_E_OK;
}

// requests that were turned away for want of a slot were counted as errors (status Internal); they are about to be applied again
static int forget_errors(tc_engine* e, uint64_t k) {
    hipLaunchKernelGGL(mk::k_counter_add, dim3(1), dim3(1), 0, cur_stream(e), e->counters + (TC_CNT_COUNT + 1) + 2, 0ull - (unsigned long long)k);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}
// ... but were not after all (the sweep or the second pass failed with something other than a full table -- the first
// pass's Internal statuses stand, and so must their count)
static void remember_errors(tc_engine* e, uint64_t k) {
    hipLaunchKernelGGL(mk::k_counter_add, dim3(1), dim3(1), 0, cur_stream(e), e->counters + (TC_CNT_COUNT + 1) + 2, (unsigned long long)k);
    (void)hipGetLastError();
}

// TC_B_ASYNC key batch: key arena and offsets are staged on the key stream (SDMA), resolved there, grouped on
// an auxiliary stream, evaluated in order, results copied back behind the evaluation; nothing waits.  A full
// key table shows as status Internal on the affected requests; the TC_E_TABLE_FULL return code is delivered
// by the next synchronous key call.  (Staging on the auxiliary stream that will group the batch, to overlap
// the transfer with the previous batch's key stage, measured slower: 1.5 vs 2.0 G decisions/s.)
static int run_keys_host_async(tc_engine* e, const tc_batch& b) {
    const uint32_t n = (uint32_t)b.n;
    TC_TRY(ensure_side_streams(e));
    const bool piped = e->key_stream != nullptr && e->n_aux != 0;
    tc_engine::SortSet& ss = e->sets[e->next_set];
    hipStream_t ks = piped ? e->key_stream : cur_stream(e);
    // (the chunk of a larger batch: its offsets start at `base`, not 0 -- only its own bytes are staged, and the kernels are
    // handed the staging pointer moved back by `base`, so that bytes + key_off[i] lands where it should)
    const size_t base = b.key_off[0];
    if (b.key_off[n] < base) return fail(e, TC_E_INVALID_ARG, "key_off must not decrease");
    const size_t total = b.key_off[n] - base;
    if (total > ss.h_key_cap) { // grow (hipFree waits for everything that may still read the old buffer)
        if (ss.h_key_bytes) (void)hipFree(ss.h_key_bytes);
        ss.h_key_bytes = nullptr;
        ss.h_key_cap = 0;
        const size_t want = std::max<size_t>(total * 2, 1 << 16);
        TC_HIP(e, hipMalloc(&ss.h_key_bytes, want));
        ss.h_key_cap = want;
    }
    if (!ss.h_key_off) TC_HIP(e, hipMalloc(&ss.h_key_off, (e->max_batch + 1) * sizeof(uint32_t)));
    // the key stage and the evaluation that last used this set's staging and slot column are done
    if (ss.in_use) TC_HIP(e, hipStreamWaitEvent(ks, ss.consumed, 0));
    if (n <= e->async_copy_kernel_n) { // (a small batch: one copy launch when the arrays are pinned -- slots.hip: stage_host_inputs)
        const void* c_src[2] = {b.key_bytes + base, b.key_off};
        void* c_dst[2] = {ss.h_key_bytes, ss.h_key_off};
        const size_t c_bytes[2] = {total, ((size_t)n + 1) * sizeof(uint32_t)};
        TC_TRY(stage_in_multi(e, c_src, c_dst, c_bytes, 2, ks));
    } else {
        if (total) TC_HIP(e, copy_async(e, ss.h_key_bytes, b.key_bytes + base, total, hipMemcpyHostToDevice, ks));
        TC_HIP(e, copy_async(e, ss.h_key_off, b.key_off, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ks));
    }
    TC_TRY(resolve_keys_device(e, ss.h_key_bytes - base, ss.h_key_off, n, true, ss.k_slot, piped));
    // (the engine cleans by itself: a synchronous caller may have to apply rejected requests again -- it needs the slots)
    if (e->as.chunk_slots_at != UINT64_MAX)
        TC_HIP(e, hipMemcpyAsync(e->k_slot + e->as.chunk_slots_at, ss.k_slot, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ks));
    tc_batch d = b;
    d.flags = (b.flags & ~(TC_B_ASYNC | TC_B_INPUTS_READY)) | TC_B_DEVICE_PTRS | (piped ? TC_B_INPUTS_READY : 0u);
    d.slot = ss.k_slot;
    d.key_bytes = nullptr;
    d.key_off = nullptr;
    d.max_burst = d.count_per_period = d.period = d.quantity = d.now_ns = nullptr;
    HostIn hin;
    hin.col[0] = b.max_burst, hin.col[1] = b.count_per_period, hin.col[2] = b.period, hin.col[3] = b.quantity, hin.col[4] = b.now_ns;
    if (b.flags & TC_B_PLAN_DICT) {
        hin.plan_dict = b.plan_dict, hin.plan_id = b.plan_id, hin.q32 = b.quantity32, hin.n_plans = b.n_plans;
        d.flags &= ~TC_B_PLAN_DICT;
        d.plan_dict = nullptr, d.plan_id = nullptr, d.quantity32 = nullptr, d.n_plans = 0;
    }
    if (piped) e->wait_before_sort = e->k_done;
    TC_TRY(stage_outputs(e, b, d));
    const int set_index = (int)e->next_set; // (== &ss: run_slots_device takes this set)
    const uint32_t chained = aside_chain(e);
    aside_reset(e);
    TC_TRY(run_slots_device(e, d, &hin));
    if (piped) aside_note(e, chained, set_index, n);
    return finish_async(e, b);
}

// a large synchronous host-pointer key batch, pipelined in chunks (engine.hpp: HOST_CHUNK_MIN)
static int run_keys_host_chunked(tc_engine* e, const tc_batch& b) {
    size_t mine = 0;
    int rc = TC_E_OK;
    uint64_t at = 0;
    for (const uint64_t cn : host_chunk_plan(e, b.n)) {
        if (rc != TC_E_OK) break;
        tc_batch c;
        host_sub_batch(b, at, cn, c);
        e->as.chunk_slots_at = auto_sweep_on(e) ? at : UINT64_MAX;
        rc = run_keys_host_async(e, c);
        e->as.chunk_slots_at = UINT64_MAX;
        if (rc == TC_E_OK) ++mine;
        else if (at) e->err += " (a later chunk of a pipelined host batch: the " + std::to_string(at) + " requests before it were applied)";
        at += cn;
    }
    e->batches -= mine ? mine - 1 : 0; // one batch, as far as the caller is concerned
    const int wrc = wait_own_async(e, mine);
    if (rc != TC_E_OK) return rc;
    if (wrc != TC_E_OK) return wrc;
    // (the slot copies behind the key stages -- what retry_rejected reads -- are the last thing on the key stream)
    if (auto_sweep_on(e) && e->key_stream) TC_HIP(e, hipStreamSynchronize(e->key_stream));
    return check_key_errors(e);
}

// a validated key batch through the path its flags name
static int keys_batch_dispatch(tc_engine* e, const tc_batch& b) {
    if (b.flags & TC_B_ASYNC) return run_keys_host_async(e, b);
    if (host_chunking_applies(e, b)) return run_keys_host_chunked(e, b);
    const uint8_t* d_bytes = b.key_bytes;
    const uint32_t* d_off = b.key_off;
    const bool dev = (b.flags & TC_B_DEVICE_PTRS) != 0;
    if (!dev && small_batch_applies(e, b)) return run_small_batch(e, b);
    if (!dev) { // (a small batch in pageable memory: through the engine's pinned block -- slots.hip: bounce_in)
        Bounced bo;
        if (bounce_in(e, b, bo)) {
            int rc = stage_keys(e, bo.bb.key_bytes, bo.bb.key_off, bo.bb.n, &d_bytes, &d_off, &bo.bb);
            if (rc != TC_E_OK) return rc;
            rc = resolve_keys_device(e, d_bytes, d_off, (uint32_t)b.n, true, e->k_slot, false);
            if (rc != TC_E_OK) return rc;
            uint32_t flag = 0;
            rc = run_slots_host_staged(e, bo.bb, &flag, e->k_slot, true);
            if (rc != TC_E_OK) return rc;
            bounce_out(bo);
            if (flag) {
                TC_HIP(e, hipMemsetAsync(e->kt.error_flag, 0, sizeof flag, cur_stream(e)));
                return fail(e, TC_E_TABLE_FULL, "key table full: some keys got status Internal (raise capacity or sweep)");
            }
            return TC_E_OK;
        }
    }
    if (!dev) { // (round 5: keys, offsets and the request columns in ONE launch when they are all pinned -- one launch less on the chain)
        int rc = stage_keys(e, b.key_bytes, b.key_off, b.n, &d_bytes, &d_off, &b);
        if (rc != TC_E_OK) return rc;
    }
    tc_batch s = b;
    s.key_bytes = nullptr;
    s.key_off = nullptr;
    if (dev) {
        // the slots go into the scratch set the grouping stage is about to use
        tc_engine::SortSet& ss = e->sets[e->next_set];
        bool piped = (b.flags & TC_B_INPUTS_READY) != 0;
        if (piped) {
            int rc = ensure_side_streams(e);
            if (rc != TC_E_OK) return rc;
            piped = e->key_stream != nullptr && e->n_aux != 0;
        }
        if (piped && ss.in_use) TC_HIP(e, hipStreamWaitEvent(e->key_stream, ss.consumed, 0)); // ss.k_slot is free again
        int rc = resolve_keys_device(e, d_bytes, d_off, (uint32_t)b.n, true, ss.k_slot, piped);
        if (rc != TC_E_OK) return rc;
        if (piped) e->wait_before_sort = e->k_done;
        else s.flags &= ~TC_B_INPUTS_READY; // the slots were resolved on the main stream: group there too
        s.slot = ss.k_slot;
        const int set_index = (int)e->next_set; // (run_slots_device takes this set)
        const uint32_t chained = aside_chain(e);
        aside_reset(e);
        rc = run_slots_device(e, s);
        if (rc == TC_E_OK && piped && e->n_aux != 0) aside_note(e, chained, set_index, (uint32_t)b.n);
        return rc;
    }
    int rc = resolve_keys_device(e, d_bytes, d_off, (uint32_t)b.n, true, e->k_slot, false);
    if (rc != TC_E_OK) return rc;
    // host pointers for everything else: the slot path's staging, with the slot column where the key stage left it
    uint32_t flag = 0;
    rc = run_slots_host_staged(e, b, &flag, e->k_slot, true);
    if (rc != TC_E_OK) return rc;
    if (flag) {
        TC_HIP(e, hipMemsetAsync(e->kt.error_flag, 0, sizeof flag, cur_stream(e)));
        return fail(e, TC_E_TABLE_FULL, "key table full: some keys got status Internal (raise capacity or sweep)");
    }
    return TC_E_OK;
}

// A SYNCHRONOUS host-pointer key batch came back with TC_E_TABLE_FULL while the engine cleans by itself: where the reference's
// HashMap would simply have grown, drop what has expired by the call's newest timestamp and apply the requests that were
// turned away once more.  They are exactly the requests whose key got no slot -- every request of such a key, so the
// sub-batch, in index order, IS the tail of those keys' sequences (keys are independent) -- and they touched no state.
// `small`: the batch went through k_small_batch (its resolved slots are in the pinned block), else through the pipeline (e->k_slot).
static int retry_rejected(tc_engine* e, const tc_batch& b0, bool small) {
    const size_t n = b0.n;
    std::vector<uint32_t> slots(n);
    if (small) {
        memcpy(slots.data(), e->small_io + e->small_slots_at, n * sizeof(uint32_t));
    } else {
        TC_HIP(e, hipMemcpyAsync(slots.data(), e->k_slot, n * sizeof(uint32_t), hipMemcpyDeviceToHost, cur_stream(e)));
        TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    }
    // (TC_B_PLAN_DICT: the rejected requests go round again in the wide form -- the caller's arrays are only read)
    std::vector<int64_t> wide[4];
    tc_batch bw = b0;
    if (bw.flags & TC_B_PLAN_DICT) {
        for (int j = 0; j < 3; ++j) wide[j].resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t k = bw.plan_id[i];
            for (int j = 0; j < 3; ++j) wide[j][i] = k < bw.n_plans ? bw.plan_dict[3 * (size_t)k + j] : 0;
        }
        bw.max_burst = wide[0].data(), bw.count_per_period = wide[1].data(), bw.period = wide[2].data();
        if (bw.quantity32) {
            wide[3].resize(n);
            for (size_t i = 0; i < n; ++i) wide[3][i] = (int64_t)bw.quantity32[i];
            bw.quantity = wide[3].data();
        }
        bw.flags &= ~TC_B_PLAN_DICT;
        bw.plan_dict = nullptr, bw.plan_id = nullptr, bw.quantity32 = nullptr, bw.n_plans = 0;
    }
    const tc_batch& b = bw;
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < n; ++i)
        if (slots[i] == kt::NO_SLOT) idx.push_back((uint32_t)i);
    if (idx.empty()) return TC_E_TABLE_FULL; // (the flag was an earlier asynchronous batch's)
    const size_t m = idx.size();
    int64_t newest = b.now_ns ? INT64_MIN : b.now_ns_scalar;
    if (b.now_ns)
        for (uint32_t i : idx) newest = std::max(newest, b.now_ns[i]);
    TC_TRY(forget_errors(e, m));
    {
        const int rcs = auto_sweep_for_retry(e, newest);
        if (rcs != TC_E_OK) {
            remember_errors(e, m);
            return rcs;
        }
    }
    std::vector<uint8_t> arena;
    std::vector<uint32_t> off(m + 1, 0u);
    for (size_t k = 0; k < m; ++k) {
        const uint32_t i = idx[k];
        arena.insert(arena.end(), b.key_bytes + b.key_off[i], b.key_bytes + b.key_off[i + 1]);
        off[k + 1] = (uint32_t)arena.size();
    }
    if (arena.empty()) arena.push_back(0);
    const int64_t* in_cols[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    std::vector<int64_t> sub_in[5];
    for (int j = 0; j < 5; ++j)
        if (in_cols[j]) {
            sub_in[j].resize(m);
            for (size_t k = 0; k < m; ++k) sub_in[j][k] = in_cols[j][idx[k]];
        }
    tc_batch sub = b;
    sub.n = m;
    sub.key_bytes = arena.data();
    sub.key_off = off.data();
    sub.max_burst = in_cols[0] ? sub_in[0].data() : nullptr;
    sub.count_per_period = in_cols[1] ? sub_in[1].data() : nullptr;
    sub.period = in_cols[2] ? sub_in[2].data() : nullptr;
    sub.quantity = in_cols[3] ? sub_in[3].data() : nullptr;
    sub.now_ns = in_cols[4] ? sub_in[4].data() : nullptr;
    std::vector<uint8_t> o_allowed(b.allowed || b.allowed_bits ? m : 0), o_status(b.status ? m : 0);
    int64_t* const want[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    std::vector<int64_t> o_col[4], o_r4(b.result4 ? 4 * m : 0);
    for (int j = 0; j < 4; ++j) o_col[j].resize(want[j] ? m : 0);
    std::vector<tc_decision> o_dec(b.decisions ? m : 0);
    sub.allowed = o_allowed.empty() ? nullptr : o_allowed.data();
    sub.allowed_bits = nullptr;
    sub.status = b.status ? o_status.data() : nullptr;
    sub.limit = want[0] ? o_col[0].data() : nullptr;
    sub.remaining = want[1] ? o_col[1].data() : nullptr;
    sub.reset_after_ns = want[2] ? o_col[2].data() : nullptr;
    sub.retry_after_ns = want[3] ? o_col[3].data() : nullptr;
    sub.result4 = b.result4 ? o_r4.data() : nullptr;
    sub.decisions = b.decisions ? o_dec.data() : nullptr;
    e->as.in_retry = true;
    const int rc = tc_rate_limit_batch_keys(e, &sub);
    e->as.in_retry = false;
    if (rc != TC_E_OK && rc != TC_E_TABLE_FULL) { // (nothing more was applied; the first pass's results stand)
        remember_errors(e, m);
        return rc;
    }
    e->batches--; // one batch, as far as the caller is concerned
    for (size_t k = 0; k < m; ++k) {
        const uint32_t i = idx[k];
        if (b.allowed) b.allowed[i] = o_allowed[k];
        if (b.allowed_bits) b.allowed_bits[i >> 6] = (b.allowed_bits[i >> 6] & ~(1ull << (i & 63u))) | ((uint64_t)(o_allowed[k] & 1u) << (i & 63u));
        if (b.status) b.status[i] = o_status[k];
        for (int j = 0; j < 4; ++j)
            if (want[j]) want[j][i] = o_col[j][k];
        if (b.result4) memcpy(b.result4 + 4 * (size_t)i, o_r4.data() + 4 * k, 4 * sizeof(int64_t));
        if (b.decisions) b.decisions[i] = o_dec[k];
    }
    return rc;
}

extern "C" int tc_rate_limit_batch_keys(tc_engine* e, const tc_batch* bp) {
    if (!e || !bp || bp->struct_size < offsetof(tc_batch, n_segments)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (!e->key_mode) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_KEY_MODE");
    tc_batch b;
    memset(&b, 0, sizeof b);
    memcpy(&b, bp, std::min<size_t>(bp->struct_size, sizeof b));
    if (b.n_segments) return fail(e, TC_E_INVALID_ARG, "segments belong to slot batches");
    if (b.n == 0) return TC_E_OK;
    if (b.n > e->max_batch) return fail(e, TC_E_BATCH_TOO_LARGE, "batch larger than max_batch");
    if (!b.key_bytes || !b.key_off) return fail(e, TC_E_INVALID_ARG, "key_bytes/key_off is NULL");
    if ((b.flags & TC_B_GROUPED_OUTPUT) && !b.order) return fail(e, TC_E_INVALID_ARG, "TC_B_GROUPED_OUTPUT needs `order`");
    if (b.flags & (TC_B_REGISTERED_PARAMS | TC_B_UNIQUE_SLOTS))
        return fail(e, TC_E_INVALID_ARG, "registered params / unique-slot promise do not apply to string keys");
    if ((b.flags & TC_B_ASYNC) && (b.flags & TC_B_DEVICE_PTRS))
        return fail(e, TC_E_INVALID_ARG, "TC_B_ASYNC is for host-pointer batches (device-pointer batches are asynchronous anyway)");
    TC_HIP(e, hipSetDevice(e->device));
    // TC_B_PLAN_DICT (round 6; slots.hip): device batches decoded by a kernel here, small host batches on the host
    TC_TRY(dict_check(e, b));
    if (!(b.flags & TC_B_PLAN_DICT)) b.plan_dict = nullptr, b.plan_id = nullptr, b.quantity32 = nullptr, b.n_plans = 0;
    else if (b.flags & TC_B_DEVICE_PTRS) TC_TRY(dict_expand_device(e, b));
    else if (!(b.flags & TC_B_ASYNC) && b.n <= (uint64_t)SMALL_MAX) dict_expand_on_host(e, b);
    if (!auto_sweep_on(e)) return keys_batch_dispatch(e, b);
    // the engine cleans by itself (tc_set_sweep_policy): maybe_clean_expired in front of the batch, the feed behind it
    const bool dev = (b.flags & TC_B_DEVICE_PTRS) != 0;
    const bool now_known = !dev || !b.now_ns;
    TC_TRY(auto_sweep_before(e, b.n, true, now_known, (!dev && b.now_ns) ? b.now_ns[0] : b.now_ns_scalar));
    const bool small = !dev && !(b.flags & TC_B_ASYNC) && small_batch_applies(e, b);
    int rc = keys_batch_dispatch(e, b);
    if (rc == TC_E_TABLE_FULL && !dev && !(b.flags & (TC_B_ASYNC | TC_B_GROUPED_OUTPUT)) && !e->as.in_retry) rc = retry_rejected(e, b, small);
    if (rc != TC_E_OK && rc != TC_E_TABLE_FULL) return rc;
    const int rc2 = auto_sweep_after(e, b.n, true, (dev && b.now_ns) ? b.now_ns + (b.n - 1) : nullptr,
                                     (!dev && b.now_ns) ? b.now_ns[b.n - 1] : b.now_ns_scalar);
    return rc2 != TC_E_OK ? rc2 : rc;
}

extern "C" int tc_rate_limit(tc_engine* e, const uint8_t* key, size_t key_len, int64_t max_burst,
                             int64_t count_per_period, int64_t period, int64_t quantity, int64_t now_ns,
                             tc_result* out) {
    if (!e || !out || (!key && key_len)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (key_len > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "key too long");
    if (e->fixed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: single calls carry their own rate; use a registered batch");
    TC_HIP(e, hipSetDevice(e->device));
    // maybe_clean_expired (adaptive_cleanup.rs:205-211) in front of the call, when the engine cleans by itself
    TC_TRY(auto_sweep_before(e, 1, true, true, now_ns));
    hipStream_t s = cur_stream(e);
    uint32_t slot = 0;
    InlineKey ik;
    memset(&ik, 0, sizeof ik);
    const uint8_t* d_long = nullptr;
    if (e->key_mode) {
        ik.len = (uint32_t)key_len;
        ik.is_inline = key_len <= sizeof ik.bytes;
        if (ik.is_inline) {
            if (key_len) memcpy(ik.bytes, key, key_len);
        } else { // long key: through the staging arena
            const uint32_t off[2] = {0u, (uint32_t)key_len};
            const uint32_t* d_off;
            int rc = stage_keys(e, key, off, 1, &d_long, &d_off);
            if (rc != TC_E_OK) return rc;
        }
        if (e->k_busy) { // key stages on the key stream come first
            TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
            e->k_busy = false;
        }
    } else {
        // slot-mode engines: the key is the 4-byte little-endian slot id
        if (key_len != 4) return fail(e, TC_E_INVALID_ARG, "slot-mode engine: key must be a 4-byte slot id");
        memcpy(&slot, key, 4);
    }
    Params p;
    memset(&p, 0, sizeof p);
    p.n = 1;
    p.burst_s = max_burst;
    p.count_s = count_per_period;
    p.period_s = period;
    p.q_s = quantity;
    p.now_s = now_ns;
    p.cells = e->cells;
    p.rate_id = e->rate_id;
    p.classes = e->classes;
    p.capacity = e->capacity;
    p.counters = e->counters;
    p.denied = e->denied;
    prof_begin(e, TC_STAGE_EVAL, s);
    hipLaunchKernelGGL(k_rate_limit_one, dim3(1), dim3(64), 0, s, p, e->kt, e->key_mode ? 1 : 0, ik, d_long, slot, e->one_result,
                       e->counters + TC_CNT_KEYS_INSERTED);
    prof_end(e, s);
    TC_HIP(e, hipGetLastError());
    if (e->key_mode) { // the key table may have changed: later key stages on the key stream wait for this
        TC_HIP(e, hipEventRecord(e->m_done, s));
        e->m_busy = true;
    }
    TC_TRY(auto_sweep_after(e, 1, true, nullptr, now_ns));
    TC_HIP(e, hipStreamSynchronize(s));
    OneResult r;
    memcpy(&r, (const void*)e->one_result_host, sizeof r); // (written by the kernel into pinned memory)
    e->batches++;
    if (r.table_full) {
        uint32_t zero = 0;
        TC_HIP(e, hipMemcpyAsync(e->kt.error_flag, &zero, sizeof zero, hipMemcpyHostToDevice, s));
        TC_HIP(e, hipStreamSynchronize(s));
        if (auto_sweep_on(e) && !e->as.in_retry) {
            // where the reference's map would have grown: drop what has expired by now and apply the request once more
            // (the first attempt touched no state and was counted as an error: taken back)
            TC_TRY(forget_errors(e, 1));
            const int rcs = auto_sweep_for_retry(e, now_ns);
            if (rcs != TC_E_OK) {
                remember_errors(e, 1);
                return rcs;
            }
            e->as.in_retry = true;
            const int rc2 = tc_rate_limit(e, key, key_len, max_burst, count_per_period, period, quantity, now_ns, out);
            e->as.in_retry = false;
            e->batches--;
            if (rc2 != TC_E_OK && rc2 != TC_E_TABLE_FULL) remember_errors(e, 1); // (the second attempt was not applied, nor counted)
            return rc2;
        }
        return fail(e, TC_E_TABLE_FULL, "key table full: the key got status Internal (raise capacity or sweep)");
    }
    out->allowed = r.d.allowed;
    out->status = r.d.status;
    out->limit = r.limit;
    out->remaining = r.d.remaining;
    out->reset_after_ns = r.d.reset_after_ns;
    out->retry_after_ns = r.d.retry_after_ns;
    return TC_E_OK;
}

// *slot = NO_SLOT when a key-mode lookup (insert == false) does not find the key
static int store_slot_of(tc_engine* e, const uint8_t* key, size_t key_len, bool insert, uint64_t* slot) {
    TC_HIP(e, hipSetDevice(e->device)); // (the key is resolved on the engine's device whatever the caller's current one is)
    if (e->key_mode) {
        if (!key && key_len) return TC_E_INVALID_ARG;
        static const uint8_t empty = 0;
        uint32_t s32 = kt::NO_SLOT;
        int rc = resolve_one_key(e, key_len ? key : &empty, key_len, insert, &s32);
        if (rc != TC_E_OK) return rc;
        if (insert && s32 == kt::NO_SLOT) return check_key_errors(e) == TC_E_OK ? fail(e, TC_E_TABLE_FULL, "key table full") : TC_E_TABLE_FULL;
        *slot = s32;
        return TC_E_OK;
    }
    if (!key || key_len != 4) return fail(e, TC_E_INVALID_ARG, "slot-mode engine: key must be a 4-byte slot id");
    uint32_t s;
    memcpy(&s, key, 4);
    if (s >= e->capacity) return fail(e, TC_E_INVALID_ARG, "slot out of range");
    *slot = s;
    return TC_E_OK;
}

// feed: a mutating operation while the engine cleans by itself -- the policy feed goes out behind the kernel, in front of the wait
static int store_op(tc_engine* e, uint64_t slot, int op, int64_t a, int64_t b, uint64_t ttl, int64_t now,
                    StoreOpResult* r, bool feed = false, bool key_batch = false) {
    if (e->fixed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: the 8-byte layout cannot hold a free ttl (Store operations need the 16-byte cell)");
    if (now < 0) return fail(e, TC_E_INVALID_ARG, "now_ns < 0");
    TC_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_store_op, dim3(1), dim3(64), 0, cur_stream(e), e->cells, slot, op, a, b, ttl, now, e->op_result);
    TC_HIP(e, hipGetLastError());
    if (feed) TC_TRY(auto_sweep_after(e, 1, key_batch, nullptr, now));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    memcpy(r, (const void*)e->op_result_host, sizeof *r); // (written by the kernel into pinned memory)
    return TC_E_OK;
}

extern "C" int tc_store_get(tc_engine* e, const uint8_t* key, size_t key_len, int64_t now_ns, int64_t* value, int* found) {
    if (!e || !value || !found) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, false, &slot);
    if (rc != TC_E_OK) return rc;
    if (slot == kt::NO_SLOT) { // key-mode: key never seen -> None
        *value = 0;
        *found = 0;
        return TC_E_OK;
    }
    StoreOpResult r;
    rc = store_op(e, slot, 0, 0, 0, 0, now_ns, &r);
    if (rc != TC_E_OK) return rc;
    *value = r.value;
    *found = r.flag;
    return TC_E_OK;
}

extern "C" int tc_store_compare_and_swap_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t old_value,
                                                  int64_t new_value, uint64_t ttl_ns, int64_t now_ns, int* swapped) {
    if (!e || !swapped) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    TC_HIP(e, hipSetDevice(e->device));
    TC_TRY(auto_sweep_before(e, 1, false, true, now_ns, 1)); // adaptive_cleanup.rs:229: self.maybe_clean_expired(now)
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, false, &slot);
    if (rc != TC_E_OK) return rc;
    if (slot == kt::NO_SLOT) { // adaptive_cleanup.rs:242: None => Ok(false)
        *swapped = 0;
        return TC_E_OK;
    }
    StoreOpResult r;
    rc = store_op(e, slot, 1, old_value, new_value, ttl_ns, now_ns, &r, true, false);
    if (rc != TC_E_OK) return rc;
    *swapped = r.flag;
    return TC_E_OK;
}

extern "C" int tc_store_set_if_not_exists_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t value,
                                                   uint64_t ttl_ns, int64_t now_ns, int* was_set) {
    if (!e || !was_set) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    TC_HIP(e, hipSetDevice(e->device));
    TC_TRY(auto_sweep_before(e, 1, true, true, now_ns, 1)); // adaptive_cleanup.rs:262: self.maybe_clean_expired(now)
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, true, &slot);
    if (rc == TC_E_TABLE_FULL && auto_sweep_on(e)) { // the reference's map would have grown: drop what has expired, once more
        TC_TRY(auto_sweep_for_retry(e, now_ns));
        rc = store_slot_of(e, key, key_len, true, &slot);
    }
    if (rc != TC_E_OK) return rc;
    StoreOpResult r;
    rc = store_op(e, slot, 2, value, 0, ttl_ns, now_ns, &r, true, true);
    if (rc != TC_E_OK) return rc;
    *was_set = r.flag;
    return TC_E_OK;
}

extern "C" int tc_lookup_slot(tc_engine* e, const uint8_t* key, size_t key_len, int64_t* slot) {
    if (!e || !slot) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (!e->key_mode) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_KEY_MODE");
    TC_HIP(e, hipSetDevice(e->device));
    uint64_t s = kt::NO_SLOT;
    int rc = store_slot_of(e, key, key_len, false, &s);
    if (rc != TC_E_OK) return rc;
    *slot = s == kt::NO_SLOT ? -1 : (int64_t)s;
    return TC_E_OK;
}

extern "C" int tc_slot_keys(tc_engine* e, uint32_t n, const uint32_t* slots, uint8_t* key_bytes, size_t key_bytes_cap,
                            uint32_t* key_off) {
    if (!e || (n && (!slots || !key_off)) || (key_bytes_cap && !key_bytes)) return TC_E_INVALID_ARG;
    e->api_seq++;
    if (!e->key_mode) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_KEY_MODE");
    if (n == 0) return TC_E_OK;
    TC_HIP(e, hipSetDevice(e->device));
    hipStream_t s = cur_stream(e);
    if (e->k_busy) {
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    struct Tmp { // scratch, released on every exit
        uint32_t* slots = nullptr;
        kt::KeyRec* rec = nullptr;
        ~Tmp() {
            if (slots) (void)hipFree(slots);
            if (rec) (void)hipFree(rec);
        }
    } d;
    TC_HIP(e, hipMalloc(&d.slots, n * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&d.rec, n * sizeof(kt::KeyRec)));
    TC_HIP(e, hipMemcpyAsync(d.slots, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_keyrecs, dim3(nblocks(n)), dim3(BLOCK), 0, s, e->kt, d.slots, n, d.rec);
    std::vector<kt::KeyRec> h(n);
    TC_HIP(e, hipMemcpyAsync(h.data(), d.rec, n * sizeof(kt::KeyRec), hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    size_t at = 0;
    int rc = TC_E_OK;
    key_off[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t len = h[i].len == kt::NO_SLOT ? 0u : h[i].len; // unbound slot: empty key
        if (at + len <= key_bytes_cap) {
            if (len <= kt::INLINE_KEY) {
                memcpy(key_bytes + at, h[i].bytes, len);
            } else {
                uint64_t off;
                memcpy(&off, h[i].bytes, 8);
                uint32_t half = 0;
                TC_HIP(e, hipMemcpy(&half, e->kt.overflow_half, sizeof half, hipMemcpyDeviceToHost));
                TC_HIP(e, hipMemcpy(key_bytes + at, e->kt.overflow + (size_t)half * e->kt.overflow_bytes + off, len, hipMemcpyDeviceToHost));
            }
            at += len;
        } else {
            rc = TC_E_INVALID_ARG; // buffer too small: offsets still describe what fits
        }
        key_off[i + 1] = (uint32_t)at;
    }
    if (rc != TC_E_OK) return fail(e, rc, "tc_slot_keys: key_bytes_cap too small");
    return TC_E_OK;
}
