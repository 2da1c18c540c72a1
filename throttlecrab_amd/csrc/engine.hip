// engine.hip -- engine life cycle: allocation, streams, rate plans, counters, profiling, test hooks (include/tcgpu.h)
#include "engine.hpp"
#include <mutex>

namespace {
struct SidePool {
    int device;
    hipStream_t main;
    bool main_owned; // the main stream is one an engine created for itself: the whole set (it + its grouping streams) is one engine's at a time
    int priority;
    std::vector<hipStream_t> streams;
    uint32_t users;
    uint32_t tried, same_queue, same_pipe, second_best;
    bool assumed;
};
std::mutex g_side_mu;
std::vector<SidePool> g_side_pools;
} // namespace

hipStream_t cur_stream(tc_engine* e) {
    if (e->user_stream) return e->user_stream;
    if (!e->own_stream) {
        // a main stream an earlier engine of this process created for itself, probed grouping streams with, and left behind?
        static const bool pool_on = [] { const char* v = getenv("TCGPU_STREAM_POOL"); return !v || atoi(v) != 0; }();
        const char* as = getenv("TCGPU_ASSUME_CONCURRENT");
        const bool assume = as && atoi(as) != 0;
        const uint32_t want = e->n_aux_want + ((e->cfg_flags & TC_CFG_KEY_MODE) ? 1u : 0u);
        if (pool_on) {
            std::lock_guard<std::mutex> lk(g_side_mu);
            for (SidePool& sp : g_side_pools)
                if (sp.main_owned && sp.users == 0 && sp.device == e->device && sp.priority == e->aux_priority && sp.assumed == assume && sp.streams.size() >= want) {
                    sp.users = 1;
                    e->own_stream = sp.main;
                    e->own_pooled = true;
                    break;
                }
        }
        if (!e->own_stream && hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    }
    return e->own_stream;
}

// begin / end of one kernel of `stage` on stream `s` (no-ops unless profiling)
void prof_begin(tc_engine* e, int stage, hipStream_t s) {
    if (!e->prof_on) return;
    if (e->prof_used == e->prof_stage.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        e->prof_ev.push_back(a);
        e->prof_ev.push_back(b);
        e->prof_stage.push_back(-1);
    }
    e->prof_stage[e->prof_used] = stage;
    (void)hipEventRecord(e->prof_ev[2 * e->prof_used], s);
}

void prof_end(tc_engine* e, hipStream_t s) {
    if (!e->prof_on || e->prof_used == e->prof_stage.size()) return;
    (void)hipEventRecord(e->prof_ev[2 * e->prof_used + 1], s);
    e->prof_used++;
}

// one record whose two events ride on a kernel's own dispatch packet (TC_LAUNCH_T); false: no record could be made
// (the launch then goes untimed).  TCGPU_PROF_MARKERS=1: the old way, marker events before and after the kernel.
bool prof_pair(tc_engine* e, int stage, hipEvent_t* start, hipEvent_t* stop) {
    if (!e->prof_on || e->prof_markers) return false;
    if (e->prof_used == e->prof_stage.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
        e->prof_ev.push_back(a);
        e->prof_ev.push_back(b);
        e->prof_stage.push_back(-1);
    }
    e->prof_stage[e->prof_used] = stage;
    *start = e->prof_ev[2 * e->prof_used];
    *stop = e->prof_ev[2 * e->prof_used + 1];
    e->prof_used++;
    return true;
}

// every host <-> device copy of a batch goes through here, so that tests can make one of them fail
hipError_t copy_async(tc_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (e->fault_countdown && --e->fault_countdown == 0) return hipErrorInvalidValue;
    return hipMemcpyAsync(dst, src, bytes, kind, st);
}

int fail(tc_engine* e, int code, const char* msg) {
    if (e) e->err = msg;
    return code;
}

extern "C" uint32_t tc_abi_version(void) { return TCGPU_ABI_VERSION; }

extern "C" const char* tc_last_error(const tc_engine* e) { return e ? e->err.c_str() : "null engine"; }

static size_t sort_ws_words(uint32_t max_tiles) { return rs::workspace_words(max_tiles); }

// the device-side address of the poison word goes into the counter block, POISON_PTR_WORDS behind the violation counter
// (tc::invariant_failed finds it there: no kernel carries an extra argument for something that never happens)
int publish_poison_ptr(tc_engine* e) {
    void* dv = nullptr;
    TC_HIP(e, hipHostGetDevicePointer(&dv, e->poison_host, 0));
    const unsigned long long v = (unsigned long long)(uintptr_t)dv;
    TC_HIP(e, hipMemcpy(e->counters + (TC_CNT_COUNT + 1) + 3 + tc::POISON_PTR_WORDS, &v, sizeof v, hipMemcpyHostToDevice));
    return TC_E_OK;
}

// sticky: once a kernel has flagged a broken invariant the engine answers nothing else
int poisoned(tc_engine* e) {
    if (e->poison_host && *(volatile uint32_t*)e->poison_host != 0u) {
        e->err = "an internal invariant failed on the device (a wait gave up, or a closed form met a state it was proven not to meet): "
                 "results and state are undefined -- destroy the engine";
        return TC_E_INVARIANT;
    }
    return TC_E_OK;
}

int engine_alloc(tc_engine* e) {
    TC_HIP(e, hipSetDevice(e->device));
    const uint64_t cap = e->capacity, mb = e->max_batch;
    e->fixed = (e->cfg_flags & TC_CFG_FIXED_PARAMS) != 0;
    if (e->fixed) TC_HIP(e, hipMalloc(&e->tat8, cap * sizeof(int64_t)));
    else TC_HIP(e, hipMalloc(&e->cells, cap * sizeof(Cell)));
    TC_HIP(e, hipMalloc(&e->rate_id, cap * sizeof(uint16_t)));
    TC_HIP(e, hipMalloc(&e->classes, (size_t)MAX_CLASSES * sizeof(RateClass)));
    e->host_classes.assign(1, RateClass{0, 0, 0, 0});
    if (e->cfg_flags & TC_CFG_TRACK_DENIED) {
        TC_HIP(e, hipMalloc(&e->denied, cap * sizeof(uint32_t)));
        TC_HIP(e, hipMemsetAsync(e->denied, 0, cap * sizeof(uint32_t), (hipStream_t)0));
        TC_HIP(e, hipMalloc(&e->topk_ws, (256 + 2 + 4 * (size_t)TOPK_MAX) * sizeof(uint32_t)));
    }
    const size_t cnt_words = (TC_CNT_COUNT + 1) + (size_t)NSHARD * SHARD_WORDS;
    TC_HIP(e, hipMalloc(&e->counters, cnt_words * sizeof(unsigned long long)));
    if (e->fixed) hipLaunchKernelGGL(k_fill_i64, dim3(std::min<uint64_t>(nblocks(cap), 4096)), dim3(BLOCK), 0, (hipStream_t)0, e->tat8, cap, tc::TAT_VACANT);
    else TC_HIP(e, hipMemsetAsync(e->cells, 0, cap * sizeof(Cell), (hipStream_t)0));
    TC_HIP(e, hipMemsetAsync(e->rate_id, 0, cap * sizeof(uint16_t), (hipStream_t)0));
    TC_HIP(e, hipMemsetAsync(e->classes, 0, (size_t)MAX_CLASSES * sizeof(RateClass), (hipStream_t)0));
    TC_HIP(e, hipMemsetAsync(e->counters, 0, cnt_words * sizeof(unsigned long long), (hipStream_t)0));
    TC_HIP(e, hipHostMalloc((void**)&e->poison_host, 64, hipHostMallocDefault));
    *e->poison_host = 0u;
    TC_TRY_EARLY(publish_poison_ptr(e));
    e->sort_max_tiles = (uint32_t)((mb + rs::THREADS * SORT_ITEMS - 1) / (rs::THREADS * SORT_ITEMS));
    const size_t words = sort_ws_words(e->sort_max_tiles);
    e->n_aux = (e->cfg_flags & TC_CFG_KEY_MODE) ? AUX_KEY_MODE : AUX_SLOT_MODE;
    if (const char* d = getenv("TCGPU_AUX_STREAMS")) e->n_aux = (uint32_t)std::min(std::max(atoi(d), 1), AUX_MAX);
    e->depth = e->n_aux + 3; // scratch sets: the grouping streams run up to two batches further ahead of the evaluation (47.3 -> 45-47 us)
    if (const char* d = getenv("TCGPU_PIPE_DEPTH")) e->depth = (uint32_t)std::min(std::max(atoi(d), 1), PIPE_DEPTH_MAX);
    int prio_lo = 0, prio_hi = 0;
    TC_HIP(e, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    if (const char* d = getenv("TCGPU_EVAL_ITEMS")) e->eval_items = atoi(d);
    if (const char* d = getenv("TCGPU_EVAL_LEAN")) e->eval_lean = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_STOP_EVENTS")) e->stop_events = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_PROF_MARKERS")) e->prof_markers = atoi(d) != 0;
#ifdef TCGPU_DEBUG_KNOBS // (make DEBUG_KNOBS=1: measurement builds only -- the shipped library cannot be told to skip its results)
    if (const char* d = getenv("TCGPU_DEBUG_NO_DECISION_STORE")) e->debug_nostore = atoi(d) != 0;
#endif
    if (const char* d = getenv("TCGPU_PREFILL")) e->prefill_on = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_GENERAL_EARLIER")) e->general_earlier = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_GENERAL_RUNS")) e->general_runs = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_GENERAL_LEAN")) e->general_lean = atoi(d) != 0;
    {
        TC_HIP(e, hipHostMalloc((void**)&e->fill_hint_host, 64, hipHostMallocDefault));
        *e->fill_hint_host = 1u;
        void* dv = nullptr;
        TC_HIP(e, hipHostGetDevicePointer(&dv, e->fill_hint_host, 0));
        e->fill_hint_dev = (uint32_t*)dv;
    }
    {
        // range path: rs::NRANGE ranges of the key space; a slot's offset inside its range must fit 16 bits
        if (const char* d = getenv("TCGPU_RANGE")) e->range_mode = atoi(d);
        e->range_max_n = (uint32_t)(rs::FIN_CAP * (uint64_t)rs::NRANGE * 3ull / 4ull); // mean range 3/4 of what a block finishes in LDS
        if (const char* d = getenv("TCGPU_RANGE_MAX_N")) e->range_max_n = (uint32_t)std::max(atoll(d), 1ll);
        if (e->range_mode != 0 && cap > 65536 && cap < 0xFFFFFFFFull) {
            e->range_mul = rs::range_mul((uint32_t)cap);
            // string mode: a key table hands neighbouring slots to the keys of one batch -- interleaved ranges (radix_sort.hpp)
            e->range_ilv = (e->cfg_flags & TC_CFG_KEY_MODE) != 0;
            if (const char* d = getenv("TCGPU_RANGE_ILV")) e->range_ilv = atoi(d) != 0;
            if (e->range_ilv && rs::ilv_width((uint32_t)cap) > 65536u) e->range_ilv = false;
            const uint32_t width = e->range_ilv ? rs::ilv_width((uint32_t)cap) : rs::range_width(e->range_mul);
            if (width <= 65536u) {
                e->range_sub_passes = width <= 256u ? 1 : 2;
                e->range_ok = true;
                TC_HIP(e, hipHostMalloc((void**)&e->range_hint_host, 64, hipHostMallocDefault));
                *e->range_hint_host = 0ull;
                void* dv = nullptr;
                TC_HIP(e, hipHostGetDevicePointer(&dv, e->range_hint_host, 0));
                e->range_hint_dev = (unsigned long long*)dv;
                // round 6: hot slots (range_part.hpp)
                if (const char* d = getenv("TCGPU_HOT")) e->hot.on = atoi(d) != 0;
                if (const char* d = getenv("TCGPU_HOT_MIN")) e->hot.heavy_min = (uint32_t)std::max(atoi(d), 2);
                if (const char* d = getenv("TCGPU_HOT_RANK")) e->hot.rank_on = atoi(d) != 0;
                if (const char* d = getenv("TCGPU_HOT_THREAD")) e->hot.threaded = atoi(d) != 0;
                if (e->hot.on) {
                    const size_t words = (size_t)ev::HEAVY_SLOTS + 8;
                    TC_HIP(e, hipMalloc(&e->hot.notes_dev, words * sizeof(unsigned long long)));
                    TC_HIP(e, hipMemsetAsync(e->hot.notes_dev, 0, words * sizeof(unsigned long long), (hipStream_t)0));
                    TC_HIP(e, hipHostMalloc((void**)&e->hot.notes_host, words * sizeof(unsigned long long) + 64, hipHostMallocDefault));
                    memset(e->hot.notes_host, 0, words * sizeof(unsigned long long) + 64);
                    TC_HIP(e, hipHostGetDevicePointer(&dv, e->hot.notes_host, 0));
                    e->hot.notes_host_dev = (unsigned long long*)dv;
                    e->hot.hint_cold_host = e->hot.notes_host + words; // (a line of its own behind the notes)
                    e->hot.hint_cold_dev = e->hot.notes_host_dev + words;
                    TC_HIP(e, hipMalloc(&e->hot.done, 64));
                    TC_HIP(e, hipMemsetAsync(e->hot.done, 0, 64, (hipStream_t)0));
                }
            }
        }
    }
    if (const char* d = getenv("TCGPU_SORT_ITEMS_PIPED")) {
        const int v = atoi(d);
        if (v == 8 || v == 16 || v == 32) e->sort_items_piped = v;
    }
    if (const char* d = getenv("TCGPU_NO_SMALL_BATCH")) e->small_off = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_COPY_KERNEL")) e->copy_kernel_off = atoi(d) == 0;
    e->host_chunk = HOST_CHUNK_DEFAULT;
    if (const char* d = getenv("TCGPU_HOST_CHUNK")) e->host_chunk = (uint64_t)std::max(0ll, atoll(d)) / 64 * 64;
    if (const char* d = getenv("TCGPU_BOUNCE_MAX")) e->bounce_max = (size_t)std::max(0ll, atoll(d));
    if (const char* d = getenv("TCGPU_ASYNC_COPY_KERNEL_N")) e->async_copy_kernel_n = (uint32_t)std::max(0ll, atoll(d));
    const char* pe = getenv("TCGPU_AUX_PRIORITY");
    const bool aux_high = pe && atoi(pe) != 0; // default: lowest priority (measured ~1 % better: the evaluation kernel is the critical path)
    // the evaluation kernel on the main stream is the critical path of the pipeline: grouping runs at
    // the lowest priority and fills what the evaluation leaves free (TCGPU_AUX_PRIORITY=1 flips it)
    e->n_aux_want = e->n_aux;
    e->aux_priority = aux_high ? prio_hi : prio_lo;
    TC_HIP(e, hipMalloc(&e->probe_ws, 2 * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&e->probe_stamps, 2 * sizeof(long long)));
    if (const char* pp = getenv("TCGPU_PIPE_PROBE")) e->pipe_probe = atoi(pp) != 0;
    // (the grouping streams themselves are created by ensure_side_streams, against the actual main stream)
    for (uint32_t si = 0; si < e->depth; ++si) {
        tc_engine::SortSet& ss = e->sets[si];
        TC_HIP(e, hipMalloc(&ss.elem_a, mb * sizeof(uint64_t)));
        TC_HIP(e, hipMalloc(&ss.elem_b, mb * sizeof(uint64_t)));
        if (e->range_ok) {
            TC_HIP(e, hipMalloc(&ss.elem_c, std::min<uint64_t>(mb, e->range_max_n) * sizeof(uint64_t)));
            TC_HIP(e, hipMalloc(&ss.range_totals, 2 * rp::NB_HOT * sizeof(uint32_t)));
            TC_HIP(e, hipMemsetAsync(ss.range_totals, 0, 2 * rp::NB_HOT * sizeof(uint32_t), (hipStream_t)0));
            const uint64_t tiles = (std::min<uint64_t>(mb, e->range_max_n) + rp::PT_TILE - 1) / rp::PT_TILE;
            TC_HIP(e, hipMalloc(&ss.part_table, tiles * rp::NB_HOT * sizeof(uint32_t)));
            TC_HIP(e, hipMalloc(&ss.hot_dev, sizeof(rp::HotDev)));
            TC_HIP(e, hipMemsetAsync(ss.hot_dev, 0, sizeof(rp::HotDev), (hipStream_t)0));
            if (e->hot.on && e->hot.rank_on) {
                TC_HIP(e, hipMalloc(&ss.hot_info, std::min<uint64_t>(mb, e->range_max_n) * sizeof(uint32_t)));
                TC_HIP(e, hipMalloc(&ss.hot_P, tiles * rp::HOT_MAX * sizeof(uint32_t)));
                TC_HIP(e, hipMalloc(&ss.hot_n, (rp::HOT_MAX + 8) * sizeof(uint32_t)));
                TC_HIP(e, hipMemsetAsync(ss.hot_n, 0, (rp::HOT_MAX + 8) * sizeof(uint32_t), (hipStream_t)0));
                TC_HIP(e, hipMalloc(&ss.hot_eval, sizeof(ev::HotEval)));
                ev::HotEval he{};
                he.info = ss.hot_info, he.prefix = ss.hot_P, he.n = ss.hot_n, he.slot = ss.hot_dev->slot, he.count = &ss.hot_dev->count;
                he.done = e->hot.done, he.ids = rp::HOT_MAX, he.tile_shift = 12;
                TC_HIP(e, hipMemcpy(ss.hot_eval, &he, sizeof he, hipMemcpyHostToDevice));
            }
        }
        TC_HIP(e, hipMalloc(&ss.ws, words * sizeof(uint32_t)));
        TC_HIP(e, hipMemsetAsync(ss.ws, 0, words * sizeof(uint32_t), (hipStream_t)0));
        TC_HIP(e, hipEventCreateWithFlags(&ss.sorted, hipEventDisableTiming));
        TC_HIP(e, hipEventCreateWithFlags(&ss.consumed, hipEventDisableTiming));
    }
    {
        e->bp_max_n = (uint32_t)std::min<uint64_t>(mb, bp::MAX_N);
        e->bp_lb = bp::pick_lb(cap, e->bp_max_n);
        const char* off = getenv("TCGPU_BUCKET");
        e->bp_ok = e->bp_lb >= 0 && !(off && atoi(off) == 0);
        e->bp_min_n = 16384;
        if (const char* d = getenv("TCGPU_BUCKET_PIPED")) e->bp_piped = atoi(d) != 0;
        if (const char* d = getenv("TCGPU_BUCKET_BACKOFF")) e->bp_backoff_len = (uint32_t)std::max(atoi(d), 0);
        if (const char* d = getenv("TCGPU_BUCKET_MIN_N")) e->bp_min_n = (uint32_t)std::max(atoi(d), 1);
        if (const char* d = getenv("TCGPU_BUCKET_SKEW")) e->bp_skew = (uint32_t)std::min<long long>(std::max(atoll(d), 1ll), bp::MAX_SKEW);
        if (e->bp_ok) {
            // the partition kernels size their LDS by the bucket count (k_scatter: 8 B per bucket + marks, k_tile_hist:
            // 4 B per bucket) and the evaluation by the bucket width: a key space whose bucket count needs more than a
            // block may have is sorted instead (a rejected launch would leave stale elements behind a valid gate)
            hipDeviceProp_t prop;
            TC_HIP(e, hipGetDeviceProperties(&prop, e->device));
            const uint32_t nbk = bp::buckets_of(cap, e->bp_lb);
            const size_t need = std::max({bp::scatter_lds_bytes(nbk), (size_t)nbk * sizeof(uint32_t), bp::eval_lds_bytes(e->bp_lb)});
            if (need > (size_t)prop.sharedMemPerBlock) e->bp_ok = false;
        }
        if (e->bp_ok) {
            TC_HIP(e, hipHostMalloc((void**)&e->bp_gate_host, sizeof(uint32_t), hipHostMallocDefault));
            *e->bp_gate_host = 0;
            e->bp_nbk = bp::buckets_of(cap, e->bp_lb);
            const size_t bytes = bp::work_bytes(e->bp_max_n, e->bp_nbk);
            for (uint32_t si = 0; si < e->depth; ++si) {
                tc_engine::SortSet& ss = e->sets[si];
                TC_HIP(e, hipMalloc(&ss.bp_scratch, bytes));
                ss.bpw = bp::carve(ss.bp_scratch, e->bp_max_n, e->bp_nbk, e->bp_lb);
                ss.bpw.skew = e->bp_skew;
                ss.bpw.gate_host = e->bp_gate_host;
            }
            TC_HIP(e, hipMalloc(&e->bp_park, (size_t)e->bp_max_n * sizeof(PendEntry)));
        }
    }
    TC_HIP(e, hipMalloc(&e->pend, (mb / 32 + 1024) * sizeof(PendEntry)));
    TC_HIP(e, hipMalloc(&e->chain, (mb / 64 + 2) * sizeof(ChainRec)));
    TC_HIP(e, hipMemsetAsync(e->chain, 0, (mb / 64 + 2) * sizeof(ChainRec), (hipStream_t)0));
    TC_HIP(e, hipMalloc(&e->loaded, (mb / 64 + 2) * sizeof(uint32_t)));
    TC_HIP(e, hipMemsetAsync(e->loaded, 0, (mb / 64 + 2) * sizeof(uint32_t), (hipStream_t)0));
    TC_HIP(e, hipMalloc(&e->pend_count, 2 * sizeof(uint32_t)));
    TC_HIP(e, hipMemsetAsync(e->pend_count, 0, 2 * sizeof(uint32_t), (hipStream_t)0));
    TC_HIP(e, hipMalloc(&e->allowed_tmp, mb));
    TC_HIP(e, hipMalloc(&e->sweep_part, 3 * (size_t)SWEEP_GRID * sizeof(uint32_t)));
    {
        static_assert(sizeof(OneResult) <= 64 && sizeof(StoreOpResult) <= 64, "one cache line each");
        TC_HIP(e, hipHostMalloc((void**)&e->host_results, 256, hipHostMallocDefault));
        memset(e->host_results, 0, 256);
        void* dv = nullptr;
        TC_HIP(e, hipHostGetDevicePointer(&dv, e->host_results, 0));
        uint8_t* d = static_cast<uint8_t*>(dv);
        e->one_result = reinterpret_cast<OneResult*>(d), e->one_result_host = reinterpret_cast<const OneResult*>(e->host_results);
        e->op_result = reinterpret_cast<StoreOpResult*>(d + 64), e->op_result_host = reinterpret_cast<const StoreOpResult*>(e->host_results + 64);
        e->one_slot = reinterpret_cast<uint32_t*>(d + 128), e->one_slot_host = reinterpret_cast<const uint32_t*>(e->host_results + 128);
    }
    TC_HIP(e, hipStreamSynchronize((hipStream_t)0)); // set-up runs on the null stream: no private stream yet
    return TC_E_OK;
}

// ---- string-key mode ----------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int key_mode_alloc(tc_engine* e, uint64_t key_arena_bytes) {
    const uint64_t cap = e->capacity, mb = e->max_batch;
    uint64_t nb = 1;
    while (nb < 2 * cap) nb <<= 1; // load factor <= 0.5
    // keys up to 112 bytes live inside their slot's KeyRec (one 128-byte line); longer ones in the overflow arena
    const uint64_t overflow = align_up(key_arena_bytes ? key_arena_bytes : std::max<uint64_t>(1u << 20, cap * 4), 256);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t o_ktab = take(nb * sizeof(kt::Entry)), o_rec = take(cap * sizeof(kt::KeyRec)), o_bound = take(cap), o_ovf = take(2 * overflow),
                 o_free = take(cap * 4), o_misc = take(64), o_tombs = take(kt::TOMB_SHARDS * 4), o_pos = take(cap * 4),
                 o_swlist = take((cap + mk::SWEEP_GRID * BLOCK) * 4), o_swoff = take(mk::SWEEP_GRID * 4);
    TC_HIP(e, hipMalloc(&e->kt_block, off));
    uint8_t* base = (uint8_t*)e->kt_block;
    TC_HIP(e, hipMemsetAsync(base + o_ktab, 0, nb * sizeof(kt::Entry), (hipStream_t)0));
    TC_HIP(e, hipMemsetAsync(base + o_misc, 0, 64, (hipStream_t)0));
    TC_HIP(e, hipMemsetAsync(base + o_tombs, 0, kt::TOMB_SHARDS * 4, (hipStream_t)0));
    kt::Table& t = e->kt;
    t.ktab = (kt::Entry*)(base + o_ktab);
    t.nb_mask = nb - 1;
    t.rec = (kt::KeyRec*)(base + o_rec);
    t.bound = base + o_bound;
    t.overflow = base + o_ovf;
    t.overflow_bytes = overflow;
    t.overflow_used = (unsigned long long*)(base + o_misc);
    t.free_top = (int*)(base + o_misc + 8);
    t.tombs = (uint32_t*)(base + o_tombs);
    t.error_flag = (uint32_t*)(base + o_misc + 16); // (+20: the rebuild's flag word)
    t.overflow_half = (uint32_t*)(base + o_misc + 24);
    // (+32, +40: the overflow compaction's flag words; +48: where the free stack stood when the sweep began)
    t.free_slots = (uint32_t*)(base + o_free);
    e->sweep_work.list = (uint32_t*)(base + o_swlist); // (a block's stretch starts at its first slot; the last one's may reach past the capacity)
    e->sweep_work.part = e->sweep_part;
    e->sweep_work.off = (uint32_t*)(base + o_swoff);
    t.pos_col = (uint32_t*)(base + o_pos);
    TC_HIP(e, hipMemsetAsync(t.pos_col, 0, cap * 4, (hipStream_t)0));
    t.capacity = (uint32_t)cap;
    t.retired = nullptr;
    t.denied = e->denied;
    if (e->denied) { // denial counts of keys that lose their slot (kt::RetiredRec)
        TC_HIP(e, hipMalloc(&e->retired, ((size_t)kt::RETIRED_CAP + 1) * sizeof(kt::RetiredRec))); // (+ the statistics record)
        TC_HIP(e, hipMemsetAsync(e->retired, 0, ((size_t)kt::RETIRED_CAP + 1) * sizeof(kt::RetiredRec), (hipStream_t)0));
        t.retired = e->retired;
    }
    if (const char* d = getenv("TCGPU_SPREAD_FREE")) e->spread_free = atoi(d) != 0;
    hipLaunchKernelGGL(kt::k_init_free, dim3(std::min<uint64_t>(nblocks(cap), 2048)), dim3(kt::THREADS), 0, (hipStream_t)0,
                       t.free_slots, t.bound, (uint32_t)cap, e->spread_free ? 1u : 0u);
    const int top = (int)cap;
    TC_HIP(e, hipMemcpyAsync(t.free_top, &top, sizeof top, hipMemcpyHostToDevice, (hipStream_t)0));
    TC_HIP(e, hipMalloc(&e->k_slot, mb * 4));
    for (uint32_t si = 0; si < e->depth; ++si) TC_HIP(e, hipMalloc(&e->sets[si].k_slot, mb * 4));
    TC_HIP(e, hipMalloc(&e->touched, (cap + 15) / 16 * 16 + 16));
    TC_HIP(e, hipMemsetAsync(e->touched, 0, (cap + 15) / 16 * 16 + 16, (hipStream_t)0));
    if (const char* d = getenv("TCGPU_SWEEP_ASIDE")) e->sweep_aside = atoi(d) != 0;
    TC_HIP(e, hipEventCreateWithFlags(&e->k_done, hipEventDisableTiming));
    TC_HIP(e, hipEventCreateWithFlags(&e->m_done, hipEventDisableTiming));
    TC_HIP(e, hipMalloc(&e->k_state, mb));
    TC_HIP(e, hipMalloc(&e->k_aux, mb * 4));
    TC_HIP(e, hipMalloc(&e->k_claim, ((size_t)nblocks(mb) + 1) * 4));
    TC_HIP(e, hipMalloc(&e->k_stage_off, (mb + 1) * 4));
    TC_HIP(e, hipStreamSynchronize((hipStream_t)0)); // set-up runs on the null stream: no private stream yet
    e->key_mode = true;
    return TC_E_OK;
}

// true if work on `b` runs while work on `a` is running (different hardware queues)
static int streams_concurrent(tc_engine* e, hipStream_t a, hipStream_t b, bool* out) {
    uint32_t zero[2] = {0u, 0u}, saw = 0;
    TC_HIP(e, hipMemcpyAsync(e->probe_ws, zero, sizeof zero, hipMemcpyHostToDevice, a));
    TC_HIP(e, hipStreamSynchronize(a));
    TC_HIP(e, hipStreamSynchronize(b));
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, e->probe_ws, e->probe_ws + 1, 30000LL); // <= 300 us
    hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(64), 0, b, e->probe_ws);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipStreamSynchronize(a));
    TC_HIP(e, hipStreamSynchronize(b));
    TC_HIP(e, hipMemcpy(&saw, e->probe_ws + 1, sizeof saw, hipMemcpyDeviceToHost));
    *out = saw != 0u;
    return TC_E_OK;
}

// true if a kernel on `b` is held back while a kernel on `a` still hands out blocks (same dispatch pipe; see k_probe_occupy).
// PIPE_REPS tries: the collision shows in some of them only (one in five in tools/pipeprobe.hip).  ~50 us per try.
static int streams_collide(tc_engine* e, hipStream_t a, hipStream_t b, bool* out) {
    constexpr int PIPE_REPS = 16;
    constexpr long long ROUND_TICKS = 800; // 8 us per round of blocks
    hipDeviceProp_t prop;
    TC_HIP(e, hipGetDeviceProperties(&prop, e->device));
    const uint32_t resident = (uint32_t)std::max(1, prop.multiProcessorCount) * 8u; // 256-thread blocks the chip holds
    long long* stamps = reinterpret_cast<long long*>(e->probe_stamps);
    *out = false;
    for (int rep = 0; rep < PIPE_REPS && !*out; ++rep) {
        hipLaunchKernelGGL(k_probe_occupy, dim3(4 * resident), dim3(256), 0, a, ROUND_TICKS, stamps);
        hipLaunchKernelGGL(k_probe_stamp, dim3(1), dim3(64), 0, b, stamps + 1);
        TC_HIP(e, hipGetLastError());
        TC_HIP(e, hipStreamSynchronize(a));
        TC_HIP(e, hipStreamSynchronize(b));
        long long h[2] = {0, 0};
        TC_HIP(e, hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        // (the stamp normally lands one round after the start -- when the first blocks leave; behind a colliding kernel, four)
        if (h[1] - h[0] > 5 * ROUND_TICKS / 2) *out = true;
    }
    return TC_E_OK;
}

// The grouping streams (and the key stream) must not share a hardware queue with the main stream or
// with each other: two active streams on one queue serialise, which costs the pipeline half its
// throughput.  Which queue a new stream lands on depends on everything the
// process created before (torch, RCCL, other engines), so candidates are created and PROBED against
// the main stream in use; the ones that do not run concurrently are dropped.  A few ms, once per main stream.
// (Late in round 4: nor a dispatch pipe with the main stream.  With the process's streams created in another order than
// bench.py's -- any GPU work on the caller's stream before the engine's first pipelined batch -- the third grouping stream
// landed four queues behind the main stream, and pipelined batches took 104 us instead of 42: tools/batch_sizes.py BS_PRE=1,
// tools/pipeprobe.hip.  Candidates that collide with the main stream are dropped like the ones that share its queue.)
// Round 6: what the probe found is kept per (process, device, main stream, priority).  Probing costs
// 96-221 launches of a 35 us kernel -- 3.4-7.8 ms on the first pipelined batch of EVERY engine, more than half of all GPU time
// in the driver's traces -- and its answer is a property of the streams, not of the engine: the grouping streams that passed
// stay in a pool when their engine goes (or while it lives: engines of one process on one main stream share them; every
// dependency is an event either way), and the next engine on that main stream takes them after one concurrency check of each
// against the main stream (a main stream handle can be reused for a new stream) instead of probing sixteen candidates.

void release_side_streams(tc_engine* e) {
    std::vector<hipStream_t> mine;
    if (e->key_stream) mine.push_back(e->key_stream);
    for (hipStream_t& a : e->aux)
        if (a) mine.push_back(a), a = nullptr;
    e->key_stream = nullptr;
    e->n_aux = 0;
    for (hipStream_t s : mine) (void)hipStreamSynchronize(s);
    if (e->side_pooled) {
        std::lock_guard<std::mutex> lk(g_side_mu);
        for (SidePool& sp : g_side_pools)
            if (!sp.main_owned && sp.device == e->device && sp.main == e->side_for && sp.priority == e->aux_priority && sp.users) --sp.users; // (the streams stay)
    } else {
        for (hipStream_t s : mine) (void)hipStreamDestroy(s);
    }
    e->side_pooled = false;
    e->side_ready = false;
}

int ensure_side_streams(tc_engine* e) {
    hipStream_t m = cur_stream(e);
    if (e->side_ready && e->side_for == m) return TC_E_OK;
    TC_HIP(e, hipStreamSynchronize(m));
    release_side_streams(e);
    const uint32_t want = e->n_aux_want + (e->key_mode ? 1u : 0u);
    const char* as = getenv("TCGPU_ASSUME_CONCURRENT");
    const bool assume = as && atoi(as) != 0;
    static const bool pool_on = [] { const char* v = getenv("TCGPU_STREAM_POOL"); return !v || atoi(v) != 0; }();
    if (pool_on) {
        std::unique_lock<std::mutex> lk(g_side_mu);
        for (size_t pi = 0; pi < g_side_pools.size(); ++pi) {
            SidePool& sp = g_side_pools[pi];
            if (sp.device != e->device || sp.main != m || sp.priority != e->aux_priority || sp.assumed != assume || sp.streams.size() < want) continue;
            if (sp.main_owned && !(e->own_pooled && m == e->own_stream)) continue; // (somebody else's set)
            bool ok = true;
            // Is it still the main stream the verdict was about (a handle can be reused for a new stream)?  One concurrency
            // check per stream, ~0.3 ms; the pipe test is not repeated -- its sixteen tries per stream ARE the probe's cost, and
            // which pipe a queue feeds does not change while the queue lives.
            for (uint32_t k = 0; ok && !assume && !sp.main_owned && k < want; ++k) { // (an owned main stream never left the pool: nothing to re-check)
                const int rc = streams_concurrent(e, m, sp.streams[k], &ok);
                if (rc != TC_E_OK) return rc;
            }
            if (!ok) {
                if (sp.users == 0) { // nobody holds them: drop the entry and probe afresh
                    for (hipStream_t s : sp.streams) (void)hipStreamDestroy(s);
                    g_side_pools.erase(g_side_pools.begin() + (long)pi);
                }
                break;
            }
            if (!sp.main_owned) ++sp.users;
            size_t gi = 0;
            if (e->key_mode) e->key_stream = sp.streams[gi++];
            e->n_aux = 0;
            for (; gi < want && e->n_aux < (uint32_t)AUX_MAX; ++gi) e->aux[e->n_aux++] = sp.streams[gi];
            e->probe_tried = sp.tried, e->probe_same_queue = sp.same_queue, e->probe_same_pipe = sp.same_pipe, e->probe_second_best = sp.second_best;
            e->probe_assumed = sp.assumed;
            e->probe_pooled = true;
            e->next_aux = 0;
            e->side_for = m;
            e->side_ready = true;
            e->side_pooled = true;
            return TC_E_OK;
        }
    }
    e->probe_pooled = false;
    std::vector<hipStream_t> good, bad, soft;
    e->probe_tried = e->probe_same_queue = e->probe_same_pipe = e->probe_second_best = 0;
    e->probe_assumed = assume;
    for (int c = 0; c < 16 && good.size() < want; ++c) {
        ++e->probe_tried;
        hipStream_t s = nullptr;
        int rc = TC_E_OK;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, e->aux_priority) != hipSuccess) rc = fail(e, TC_E_HIP, "hipStreamCreateWithPriority failed");
        bool ok = false;
        // TCGPU_ASSUME_CONCURRENT=1 (counter passes of a profiler that serialises every dispatch: the probe then finds
        // no concurrent stream and the batches would run in order, on other kernel variants than the timed run's):
        // take the candidates unprobed.  Ordering is by events either way; only the overlap is lost.
        if (assume) ok = true;
        else if (rc == TC_E_OK) rc = streams_concurrent(e, m, s, &ok);
        for (size_t g = 0; !assume && rc == TC_E_OK && ok && g < good.size(); ++g) rc = streams_concurrent(e, good[g], s, &ok);
        if (rc == TC_E_OK && !ok) ++e->probe_same_queue;
        bool second_best = false; // concurrent with everything, off the main stream's pipe, but on the pipe of a stream already kept
        if (!assume && rc == TC_E_OK && ok && e->pipe_probe) {
            bool collide = false;
            rc = streams_collide(e, m, s, &collide);
            ok = !collide;
            if (collide) ++e->probe_same_pipe;
            for (size_t g = 0; rc == TC_E_OK && ok && !second_best && g < good.size(); ++g) {
                rc = streams_collide(e, good[g], s, &collide);
                second_best = collide;
            }
        }
        if (rc != TC_E_OK) {
            if (s) (void)hipStreamDestroy(s);
            for (hipStream_t x : good) (void)hipStreamDestroy(x);
            for (hipStream_t x : bad) (void)hipStreamDestroy(x);
            for (hipStream_t x : soft) (void)hipStreamDestroy(x);
            return rc;
        }
        if (ok && second_best) soft.push_back(s);
        else (ok ? good : bad).push_back(s);
    }
    // (four pipes: main + three streams of their own pipe each is what there is; whoever wants more takes the second best)
    while (good.size() < want && !soft.empty()) {
        hipStream_t s = soft.front();
        soft.erase(soft.begin());
        bool ok = true; // (streams kept after it were not probed against it)
        for (size_t g = 0; ok && g < good.size(); ++g)
            if (streams_concurrent(e, good[g], s, &ok) != TC_E_OK) ok = false;
        (ok ? good : bad).push_back(s);
        if (ok) ++e->probe_second_best;
    }
    for (hipStream_t s : soft) (void)hipStreamDestroy(s);
    for (hipStream_t s : bad) (void)hipStreamDestroy(s);
    size_t gi = 0;
    if (e->key_mode && !good.empty()) e->key_stream = good[gi++];
    e->n_aux = 0;
    for (; gi < good.size() && e->n_aux < (uint32_t)AUX_MAX; ++gi) e->aux[e->n_aux++] = good[gi];
    for (; gi < good.size(); ++gi) (void)hipStreamDestroy(good[gi]); // (more than an engine uses)
    e->next_aux = 0;
    e->side_for = m;
    e->side_ready = true;
    if (pool_on && good.size() >= want && want != 0) { // the full set: worth keeping for the engines to come
        std::lock_guard<std::mutex> lk(g_side_mu);
        bool have = false;
        for (const SidePool& sp : g_side_pools) have = have || (sp.device == e->device && sp.main == m && sp.priority == e->aux_priority);
        if (!have) {
            const bool owned = m == e->own_stream && !e->user_stream;
            SidePool sp{e->device, m, owned, e->aux_priority, {}, 1u, e->probe_tried, e->probe_same_queue, e->probe_same_pipe, e->probe_second_best, assume};
            if (owned) e->own_pooled = true; // (the main stream stays with the set when this engine goes)
            if (e->key_stream) sp.streams.push_back(e->key_stream);
            for (uint32_t k = 0; k < e->n_aux; ++k) sp.streams.push_back(e->aux[k]);
            g_side_pools.push_back(sp);
            e->side_pooled = true;
        }
    }
    return TC_E_OK; // n_aux == 0: no free hardware queue, TC_B_INPUTS_READY batches run in order on the main stream
}

extern "C" tc_engine* tc_engine_create(const tc_config* cfg, int* err) {
    int dummy;
    if (!err) err = &dummy;
    *err = TC_E_OK;
    if (!cfg || cfg->struct_size < sizeof(tc_config) || cfg->capacity == 0 || cfg->max_batch == 0 ||
        cfg->capacity >= 0x7FFFFFFFull || cfg->max_batch >= 0x3FFFFFFFull) {
        *err = TC_E_INVALID_ARG;
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id < 0 || cfg->device_id >= ndev) {
        *err = TC_E_NO_DEVICE;
        return nullptr;
    }
    if ((cfg->flags & TC_CFG_FIXED_PARAMS) && (cfg->flags & TC_CFG_KEY_MODE)) {
        *err = TC_E_UNSUPPORTED; // string keys carry their rate with every request: nothing is fixed
        return nullptr;
    }
    tc_engine* e = new (std::nothrow) tc_engine();
    if (!e) {
        *err = TC_E_NOMEM;
        return nullptr;
    }
    e->device = cfg->device_id;
    e->capacity = cfg->capacity;
    e->max_batch = cfg->max_batch;
    e->cfg_flags = cfg->flags;
    int rc = engine_alloc(e);
    if (rc == TC_E_OK && (cfg->flags & TC_CFG_KEY_MODE)) rc = key_mode_alloc(e, cfg->key_arena_bytes);
    if (rc != TC_E_OK) {
        fprintf(stderr, "tcgpu: engine_create failed: %s\n", e->err.c_str());
        *err = rc;
        tc_engine_destroy(e);
        return nullptr;
    }
    return e;
}

extern "C" void tc_engine_destroy(tc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->user_stream) (void)hipStreamSynchronize(e->user_stream);
    if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    hipStream_t key_stream_was = e->key_stream;
    release_side_streams(e); // (back to the pool, or destroyed)
    (void)key_stream_was;
    for (tc_engine::SortSet& ss : e->sets) {
        if (ss.sorted) (void)hipEventDestroy(ss.sorted);
        if (ss.consumed) (void)hipEventDestroy(ss.consumed);
        void* sp[] = {ss.bp_scratch, ss.elem_a, ss.elem_b, ss.elem_c, ss.range_totals, ss.part_table, ss.hot_dev, ss.hot_info, ss.hot_P, ss.hot_n, ss.hot_eval, ss.ws, ss.h_slot, ss.h_in[0], ss.h_in[1], ss.h_in[2], ss.h_in[3], ss.h_in[4], ss.h_key_bytes, ss.h_key_off};
        for (void* p : sp)
            if (p) (void)hipFree(p);
    }
    void* ptrs[] = {e->route_ws, e->bp_park, e->cells, e->tat8, e->rate_id, e->classes, e->denied, e->topk_ws, e->probe_ws, e->probe_stamps, e->counters, e->pend, e->chain, e->loaded, e->pend_count,
                    e->allowed_tmp, e->stage.slot, e->stage.in[0], e->stage.in[1], e->stage.in[2],
                    e->stage.in[3], e->stage.in[4], e->stage.allowed, e->stage.bits, e->stage.out[0],
                    e->stage.out[1], e->stage.out[2], e->stage.out[3], e->stage.status, e->stage.result4, e->stage.decisions, e->stage.order};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->bp_gate_host) (void)hipHostFree(e->bp_gate_host);
    if (e->host_results) (void)hipHostFree(e->host_results);
    if (e->bounce) (void)hipHostFree(e->bounce);
    if (e->fill_hint_host) (void)hipHostFree(e->fill_hint_host);
    if (e->range_hint_host) (void)hipHostFree(e->range_hint_host);
    hot_worker_stop(e); // (it reads the pinned notes)
    if (e->hot.notes_host) (void)hipHostFree(e->hot.notes_host);
    if (e->hot.notes_dev) (void)hipFree(e->hot.notes_dev);
    if (e->hot.done) (void)hipFree(e->hot.done);
    if (e->route_l0_done) (void)hipEventDestroy(e->route_l0_done);
    for (uint32_t k = 0; k < e->debug_fillers; ++k) {
        (void)hipStreamSynchronize(e->debug_filler[k]);
        (void)hipStreamDestroy(e->debug_filler[k]);
    }
    if (e->poison_host) (void)hipHostFree(e->poison_host);
    if (e->as.feed_host) (void)hipHostFree(e->as.feed_host);
    if (e->k_done) (void)hipEventDestroy(e->k_done);
    if (e->m_done) (void)hipEventDestroy(e->m_done);
    for (tc_engine::SortSet& ss : e->sets)
        if (ss.k_slot) (void)hipFree(ss.k_slot);
    void* kptrs[] = {e->sweep_part, e->retired, e->kt_block, e->k_slot, e->k_state, e->k_aux, e->k_claim, e->k_stage_bytes, e->k_stage_off, e->touched};
    for (void* p : kptrs)
        if (p) (void)hipFree(p);
    if (e->small_io) (void)hipHostFree(e->small_io);
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->async_done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->async_pool) (void)hipEventDestroy(ev);
    if (e->own_stream && e->own_pooled) { // back to the pool with its grouping streams: the next engine without a stream of the caller's takes the set
        std::lock_guard<std::mutex> lk(g_side_mu);
        for (SidePool& sp : g_side_pools)
            if (sp.main_owned && sp.main == e->own_stream) sp.users = 0;
    } else if (e->own_stream) {
        (void)hipStreamDestroy(e->own_stream);
    }
    delete e;
}

extern "C" int tc_engine_set_stream(tc_engine* e, void* hip_stream) {
    if (!e) return TC_E_INVALID_ARG;
    e->api_seq++;
    TC_HIP(e, hipSetDevice(e->device));
    if (e->user_stream || e->own_stream) TC_HIP(e, hipStreamSynchronize(cur_stream(e))); // drain the old one first
    e->user_stream = (hipStream_t)hip_stream;
    e->side_ready = false; // the side streams are probed against the main stream in use
    return TC_E_OK;
}

extern "C" int tc_synchronize(tc_engine* e) {
    if (!e) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    // every TC_B_ASYNC batch recorded its completion on this stream
    while (!e->async_done.empty()) {
        e->async_pool.push_back(e->async_done.front());
        e->async_done.pop_front();
    }
    return poisoned(e);
}

// (burst, count, period) -> class id, creating (and uploading) the class if new.
// 0 = invalid triple; -1 = dictionary full.
int intern_class(tc_engine* e, int64_t burst, int64_t count, int64_t period, bool* grew) {
    RateClass rc{0, 0, 0, 0};
    if (tc::derive_rate(burst, count, period, rc.ei, rc.dvt) != tc::ST_OK) return 0;
    rc.burst = burst;
    const int64_t key[3] = {burst, count, period};
    const std::string k((const char*)key, sizeof key);
    auto it = e->class_of.find(k);
    if (it != e->class_of.end()) return it->second;
    if (e->host_classes.size() >= MAX_CLASSES) return -1;
    const uint16_t id = (uint16_t)e->host_classes.size();
    e->cls_min_ei = std::min(e->cls_min_ei, rc.ei);
    e->cls_max_ei = std::max(e->cls_max_ei, rc.ei);
    e->cls_min_dvt = std::min(e->cls_min_dvt, rc.dvt);
    e->cls_max_dvt = std::max(e->cls_max_dvt, rc.dvt);
    e->host_classes.push_back(rc);
    e->class_of.emplace(k, id);
    *grew = true;
    return id;
}

int upload_classes(tc_engine* e) {
    TC_HIP(e, hipMemcpyAsync(e->classes, e->host_classes.data(), e->host_classes.size() * sizeof(RateClass),
                             hipMemcpyHostToDevice, cur_stream(e)));
    return TC_E_OK;
}

extern "C" int tc_register_params_uniform(tc_engine* e, int64_t max_burst, int64_t count_per_period, int64_t period) {
    if (!e) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    bool grew = false;
    const int id = intern_class(e, max_burst, count_per_period, period, &grew);
    if (id == 0) return fail(e, TC_E_INVALID_ARG, "tc_register_params_uniform: invalid (burst,count,period)");
    if (id < 0) return fail(e, TC_E_UNSUPPORTED, "more than 65535 distinct rate plans registered");
    if (e->fixed) {
        if (e->sealed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: plans cannot change once a request has been decided");
        const RateClass& rc = e->host_classes[id];
        if (!tc::fixed_plan_ok(rc.ei, rc.dvt)) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: the plan needs burst >= 2 and an emission interval / tolerance below 2^60 ns");
    }
    TC_HIP(e, hipSetDevice(e->device));
    if (grew) {
        int rc = upload_classes(e);
        if (rc != TC_E_OK) return rc;
    }
    hipLaunchKernelGGL(k_fill_rate_id, dim3(std::min<uint64_t>(nblocks(e->capacity), 4096)), dim3(BLOCK), 0, cur_stream(e),
                       e->rate_id, e->capacity, (uint16_t)id);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    e->uniform_id = (uint16_t)id;
    return TC_E_OK;
}

extern "C" int tc_register_params(tc_engine* e, uint64_t n, const uint32_t* slots, const int64_t* max_burst,
                                  const int64_t* count_per_period, const int64_t* period) {
    if (!e || !max_burst || !count_per_period || !period) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (n == 0) return TC_E_OK;
    if (!slots && n > e->capacity) return fail(e, TC_E_INVALID_ARG, "tc_register_params: n > capacity");
    // validate everything before touching the dictionary or the device
    if (e->fixed && e->sealed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: plans cannot change once a request has been decided");
    for (uint64_t i = 0; i < n; ++i) {
        if (slots && slots[i] >= e->capacity) return fail(e, TC_E_INVALID_ARG, "tc_register_params: slot out of range");
        int64_t ei, dvt;
        if (tc::derive_rate(max_burst[i], count_per_period[i], period[i], ei, dvt) != tc::ST_OK)
            return fail(e, TC_E_INVALID_ARG, "tc_register_params: invalid (burst,count,period)");
        if (e->fixed && !tc::fixed_plan_ok(ei, dvt))
            return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: every plan needs burst >= 2 and an emission interval / tolerance below 2^60 ns");
    }
    std::vector<uint16_t> ids(n);
    bool grew = false;
    int last_id = 0;
    // plans are few (tiers) and keys many: a small direct-mapped memo in front of the dictionary (10 M keys hashed over
    // 1000 tiers: 0.2 s instead of 2 s of map look-ups)
    struct Memo {
        int64_t b, c, p;
        int id;
    };
    std::vector<Memo> memo(4096, Memo{0, 0, 0, -1});
    for (uint64_t i = 0; i < n; ++i) {
        if (i && max_burst[i] == max_burst[i - 1] && count_per_period[i] == count_per_period[i - 1] && period[i] == period[i - 1]) {
            ids[i] = (uint16_t)last_id; // runs of one plan are the common case
            continue;
        }
        const uint64_t h = ((uint64_t)max_burst[i] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)count_per_period[i] * 0xC2B2AE3D27D4EB4Full) ^
                           ((uint64_t)period[i] * 0x165667B19E3779F9ull);
        Memo& m = memo[(h >> 40) & 4095u];
        if (m.id > 0 && m.b == max_burst[i] && m.c == count_per_period[i] && m.p == period[i]) {
            last_id = m.id;
        } else {
            last_id = intern_class(e, max_burst[i], count_per_period[i], period[i], &grew);
            if (last_id < 0) return fail(e, TC_E_UNSUPPORTED, "more than 65535 distinct rate plans registered");
            m = Memo{max_burst[i], count_per_period[i], period[i], last_id};
        }
        ids[i] = (uint16_t)last_id;
    }
    TC_HIP(e, hipSetDevice(e->device));
    if (grew) {
        int rc = upload_classes(e);
        if (rc != TC_E_OK) return rc;
    }
    struct Tmp { // upload buffers, released on every exit
        uint16_t* id = nullptr;
        uint32_t* slot = nullptr;
        ~Tmp() {
            if (id) (void)hipFree(id);
            if (slot) (void)hipFree(slot);
        }
    } d;
    TC_HIP(e, hipMalloc(&d.id, n * sizeof(uint16_t)));
    if (slots) TC_HIP(e, hipMalloc(&d.slot, n * sizeof(uint32_t)));
    TC_HIP(e, hipMemcpyAsync(d.id, ids.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice, cur_stream(e)));
    if (slots) TC_HIP(e, hipMemcpyAsync(d.slot, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream(e)));
    hipLaunchKernelGGL(k_scatter_rate_id, dim3(nblocks(n)), dim3(BLOCK), 0, cur_stream(e), e->rate_id, d.slot, d.id, n);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    e->uniform_id = 0; // per-slot plans from now on: evaluation reads rate_id[]
    return TC_E_OK;
}

extern "C" int tc_counters_refresh(tc_engine* e) {
    if (!e) return TC_E_INVALID_ARG;
    e->api_seq++;
    TC_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(NSHARD), 0, cur_stream(e), e->counters);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

extern "C" int tc_counters(tc_engine* e, uint64_t out[TC_CNT_COUNT]) {
    if (!e || !out) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    int rc = tc_counters_refresh(e);
    if (rc != TC_E_OK) return rc;
    int top = 0;
    if (e->key_mode) { // string mode: the store's size is the number of keys that hold a slot (key stages in flight come first)
        if (e->k_busy) {
            TC_HIP(e, hipStreamWaitEvent(cur_stream(e), e->k_done, 0));
            e->k_busy = false;
        }
        TC_HIP(e, hipMemcpyAsync(&top, e->kt.free_top, sizeof top, hipMemcpyDeviceToHost, cur_stream(e)));
    }
    TC_HIP(e, hipMemcpyAsync(out, e->counters, TC_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    out[TC_CNT_BATCHES] = e->batches;
    if (e->key_mode) out[TC_CNT_LIVE_SLOTS] = e->capacity - std::min<uint64_t>(e->capacity, (uint64_t)std::max(top, 0));
    return TC_E_OK;
}

// ---- tc_debug_occupy: fillers for the forward-progress tests --------------------------------------------------------------
static __global__ void k_filler(long long ticks) { // every thread stays for `ticks` of the 100 MHz wall clock
    extern __shared__ uint32_t s_fill[];
    if (threadIdx.x == 0) s_fill[0] = 1u; // (keeps the dynamic LDS allocated)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int tc_debug_occupy(tc_engine* e, const uint32_t* cu_mask, uint32_t blocks, uint32_t lds_bytes, uint64_t microseconds) {
    if (!e || blocks == 0 || blocks > 65536 || lds_bytes > 160 * 1024 || microseconds > 2000000ull) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = cu_mask ? cu_mask[i] : 0xFFFFFFFFu;
    uint32_t at = e->debug_fillers;
    for (uint32_t k = 0; k < e->debug_fillers; ++k)
        if (memcmp(e->debug_filler_mask[k], mask, sizeof mask) == 0) at = k;
    if (at == e->debug_fillers) {
        if (at == 4) return fail(e, TC_E_INVALID_ARG, "tc_debug_occupy: at most 4 distinct CU masks per engine");
        TC_HIP(e, hipExtStreamCreateWithCUMask(&e->debug_filler[at], 8, mask));
        memcpy(e->debug_filler_mask[at], mask, sizeof mask);
        e->debug_fillers++;
    }
    if (lds_bytes > 64 * 1024)
        TC_HIP(e, hipFuncSetAttribute(reinterpret_cast<const void*>(k_filler), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_filler, dim3(blocks), dim3(256), lds_bytes < 4 ? 4 : lds_bytes, e->debug_filler[at], (long long)(microseconds * 100ull));
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

extern "C" int tc_profile_enable(tc_engine* e, int on) {
    if (!e) return TC_E_INVALID_ARG;
    e->prof_on = on != 0;
    e->prof_used = 0;
    for (int i = 0; i < TC_STAGE_COUNT; ++i) {
        e->prof_ms[i] = 0;
        e->prof_calls[i] = 0;
    }
    return TC_E_OK;
}

extern "C" int tc_profile_read(tc_engine* e, double total_ms[TC_STAGE_COUNT], uint64_t calls[TC_STAGE_COUNT]) {
    if (!e || !total_ms || !calls) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    for (hipStream_t a : e->aux)
        if (a) TC_HIP(e, hipStreamSynchronize(a));
    if (e->key_stream) TC_HIP(e, hipStreamSynchronize(e->key_stream));
    for (size_t i = 0; i < e->prof_used; ++i) {
        const int st = e->prof_stage[i];
        if (st < 0) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]) == hipSuccess) {
            e->prof_ms[st] += ms;
            e->prof_calls[st] += 1;
        }
    }
    e->prof_used = 0;
    for (int i = 0; i < TC_STAGE_COUNT; ++i) {
        total_ms[i] = e->prof_ms[i];
        calls[i] = e->prof_calls[i];
    }
    return TC_E_OK;
}

extern "C" int tc_counters_device_ptr(tc_engine* e, void** dptr) {
    if (!e || !dptr) return TC_E_INVALID_ARG;
    *dptr = e->counters;
    return TC_E_OK;
}

// Test hook: the n-th staging copy (host <-> device, any batch path) from now returns an error instead of
// being issued.  0 disarms.
extern "C" int tc_debug_break_wait(tc_engine* e, uint32_t on) {
    if (!e) return TC_E_INVALID_ARG;
    e->debug_break_wait = on != 0;
    return TC_E_OK;
}

extern "C" int tc_debug_fail_copy(tc_engine* e, uint32_t nth) {
    if (!e) return TC_E_INVALID_ARG;
    e->fault_countdown = nth;
    return TC_E_OK;
}

// Internal invariant violations seen so far (0 unless there is a bug): runs that turned out
// irregular in a batch the host had proved regular (k_eval_sorted<DIRECT>).
extern "C" int tc_engine_info_get(tc_engine* e, tc_engine_info* out) {
    if (!e || !out || out->struct_size < offsetof(tc_engine_info, sweeps_aside)) return TC_E_INVALID_ARG;
    const uint32_t caller_size = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(tc_engine_info));
    tc_engine_info r;
    memset(&r, 0, sizeof r);
    r.struct_size = caller_size;
    const bool probed = e->side_ready && e->side_for == cur_stream(e);
    r.side_streams_probed = probed ? 1u : 0u;
    r.grouping_streams_wanted = e->n_aux_want;
    if (probed) {
        r.grouping_streams = e->n_aux;
        r.key_stream = e->key_stream ? 1u : 0u;
        r.candidates_tried = e->probe_tried;
        r.rejected_same_queue = e->probe_same_queue;
        r.rejected_same_pipe = e->probe_same_pipe;
        r.kept_second_best = e->probe_second_best;
        r.probes_assumed = e->probe_assumed ? 1u : 0u;
        r.pipelining_degraded = (e->n_aux < e->n_aux_want || (e->key_mode && !e->key_stream)) ? 1u : 0u;
    }
    r.scratch_sets = e->depth;
    r.grouping_path = e->last_grouping_path;
    r.range_path_possible = e->range_ok ? 1u : 0u;
    if (e->range_hint_host) {
        const unsigned long long h = *(volatile unsigned long long*)e->range_hint_host;
        r.range_hint_requests = h >> 32;
        r.range_hint_largest = h & 0xFFFFFFFFull;
    }
    r.host_chunk_requests = e->host_chunk;
    r.batches = e->batches;
    r.hot_slots = e->hot.slots.size();
    r.hot_batches = e->hot.batches_hot;
    r.probes_pooled = (probed && e->probe_pooled) ? 1u : 0u;
    r.sweeps_aside = e->sweeps_aside;
    memcpy(out, &r, caller_size);
    return TC_E_OK;
}

extern "C" int tc_selfcheck(tc_engine* e, uint64_t* violations) {
    if (!e || !violations) return TC_E_INVALID_ARG;
    e->api_seq++;
    TC_HIP(e, hipSetDevice(e->device));
    unsigned long long v = 0;
    TC_HIP(e, hipMemcpyAsync(&v, e->counters + (TC_CNT_COUNT + 1) + 3, sizeof v, hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    *violations = v;
    return TC_E_OK;
}
