// tcgpu.hip -- MI355X (gfx950) batched GCRA engine: HIP kernels + the C ABI of
// include/tcgpu.h.  One engine = one GPU-resident key store:
//
//   cells[capacity]    Cell {tat i64, expiry u64}                    16 B / key  (the state)
//   rate_id[capacity]  u16  registered rate plan of the key, 0 = none    2 B / key
//   classes[65536]     RateClass {ei, dvt, burst}   dictionary of the registered plans
//
// A batch is applied with the reference's sequential semantics
// (throttlecrab/src/core/rate_limiter.rs:102-250 applied in index order):
//   unique slots : k_eval_unique   one lane per request
//   duplicates   : rs::k_hist + rs::k_onesweep x passes  (stable (slot,index) sort), then
//                  k_eval_sorted   one `now`/`quantity` per batch: closed form per key run
//                                  (direct stores, or k_commit_list for cells whose run spans waves)
//                  k_eval_general  per-request now/quantity/rate: wave-cooperative runs of denials,
//                                  cross-wave hand-over chain with speculative look-back
//   string keys  : kt::k_probe / k_bind / k_follow resolve keys to slots first (key_table.hpp)
// Batches flagged TC_B_INPUTS_READY are grouped on auxiliary streams while earlier batches are
// evaluated on the engine's stream.  HBM-transaction-bound integer work; MFMA is not used (no
// dense contraction).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <deque>
#include <vector>

#include "../../include/tcgpu.h"
#include "gcra_math.hpp"
#include "key_table.hpp"
#include "radix_sort.hpp"

#include "eval_kernels.hpp"
#include "bucket_path.hpp"
#include "maintenance_kernels.hpp"
#include "route_kernels.hpp"

using tc::Cell;
using tc::RateClass;

using namespace ev;
using namespace mk;

namespace {

// Items per thread of a sort tile.  Alone on the chip, 2048-request tiles (512 blocks per 1 Mi batch) are
// fastest (18.6 vs 21.1 us per pass).  Pipelined, the tiles of up to three sorts spin in their look-back
// next to the evaluation kernel and every one of them holds 4 wave slots: 4096-request tiles halve that
// (evaluation 63 -> 56 us under overlap, 13.9 -> 15.1 G decisions/s).
constexpr int SORT_ITEMS = 8;        // batches that run in order on the engine's stream
constexpr int SORT_ITEMS_PIPED = 16; // TC_B_INPUTS_READY batches (grouped on the auxiliary streams)
constexpr int PIPE_DEPTH_MAX = 8;
constexpr uint32_t BP_BACKOFF = 32;          // batches sorted without trying the bucket path after a skewed one was seen
constexpr int AUX_MAX = 4;                  // auxiliary (grouping) streams
// HIP multiplexes streams onto 4 hardware queues; two ACTIVE streams on one queue serialise each other
// (a barrier packet of one blocks the other: measured 14.7 -> 7.4 G/s with a fifth stream).  Slot mode:
// main + 3 grouping streams; string mode: main + key stream + 2.  TCGPU_AUX_STREAMS / TCGPU_PIPE_DEPTH
// override (bench.py uses 2 when RCCL's stream is in the process too).
constexpr int AUX_SLOT_MODE = 3, AUX_KEY_MODE = 2;
// grouping scratch sets = batches whose sort may be in flight at once: one more than the grouping streams

inline uint32_t nblocks(uint64_t n) { return (uint32_t)((n + BLOCK - 1) / BLOCK); }
inline int bit_width_u64(uint64_t v) {
    int b = 0;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b;
}

} // namespace

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
struct tc_engine {
    int device = 0;
    // The stream evaluations are ordered on: the caller's (tc_engine_set_stream) or a private
    // one that is only created when first needed -- HIP multiplexes streams onto few hardware
    // queues (4 by default), and a main stream that shares a queue with an auxiliary stream
    // serialises the pipeline (measured: 12 -> 5.7 G decisions/s), so no stream is created idly.
    hipStream_t own_stream = nullptr;
    hipStream_t user_stream = nullptr;
    uint64_t capacity = 0, max_batch = 0;
    uint32_t cfg_flags = 0;

    Cell* cells = nullptr;
    int64_t* tat8 = nullptr;                 // TC_CFG_FIXED_PARAMS: one TAT per key instead of cells (gcra_math.hpp)
    bool fixed = false;
    bool sealed = false;                     // fixed layout: a request has been decided, the plans can no longer change
    uint16_t* rate_id = nullptr;
    RateClass* classes = nullptr;            // device, MAX_CLASSES entries, [0] = all zero
    std::vector<RateClass> host_classes;     // host mirror, index = class id
    std::unordered_map<std::string, uint16_t> class_of; // (burst,count,period) bytes -> id
    uint16_t uniform_id = 0;                 // != 0: every slot carries this plan
    uint32_t* denied = nullptr;              // TC_CFG_TRACK_DENIED: denials per slot
    uint32_t* topk_ws = nullptr;             // tc_top_denied scratch: 256 histogram words | 2 counters | lists
    unsigned long long* counters = nullptr; // TC_CNT_COUNT canonical + 1 scratch + NSHARD*SHARD_WORDS shards

    // grouping scratch: a ring of `depth` sets.  A batch flagged TC_B_INPUTS_READY is
    // sorted on its set's auxiliary stream while earlier batches are still being evaluated
    // on `stream` (the sort never touches the resident state); evaluation stays in order.
    struct SortSet {
        uint64_t *elem_a = nullptr, *elem_b = nullptr; // max_batch each
        uint32_t* ws = nullptr;                        // hist x2 | ticket | look-back status
        uint32_t* k_slot = nullptr;                    // key mode: slots resolved for the batch using this set
        uint32_t* h_slot = nullptr;                    // TC_B_ASYNC: the host batch's slot column, staged (lazy)
        int64_t* h_in[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // ... and its request columns (lazy)
        uint8_t* h_key_bytes = nullptr;                // TC_B_ASYNC key batch: its key arena and offsets, staged (lazy)
        size_t h_key_cap = 0;
        uint32_t* h_key_off = nullptr;
        uint32_t hist_parity = 0;
        void* bp_scratch = nullptr;    // bucket path: tile histograms, offsets, partitioned elements, gate
        bp::Work bpw;
        hipEvent_t sorted = nullptr;   // recorded on the auxiliary stream after the last pass
        hipEvent_t consumed = nullptr; // recorded on `stream` after the evaluation that read this set
        bool in_use = false;
        bool grouped_aside = false;    // the set's last batch was grouped on an auxiliary stream (`sorted` says when)
        // the caller-owned slot columns this set's batches read (the latest few: a set is reused every `depth` batches).
        // tc_route_batch (TC_ROUTE_AHEAD) orders itself behind the readers of the buffer it overwrites: waiting for the
        // set's CURRENT grouping covers every earlier batch of the set too (a set's next grouping waits for the
        // evaluation of its previous batch, which waited for that batch's grouping)
        struct Reader {
            const uint32_t* ptr = nullptr;
            size_t n = 0;
        } readers[4];
        uint32_t next_reader = 0;
    } sets[PIPE_DEPTH_MAX];
    uint32_t depth = 0; // sets actually allocated
    hipStream_t aux[AUX_MAX] = {};       // set k groups on aux[k % n_aux]
    uint32_t n_aux = 0;
    uint32_t next_aux = 0;
    uint32_t n_aux_want = 0;             // configured number of grouping streams
    int sort_items_piped = SORT_ITEMS_PIPED; // requests per thread of a sort tile, pipelined batches (TCGPU_SORT_ITEMS_PIPED = 8 | 16 | 32)
    hipStream_t side_for = nullptr;      // main stream the side streams were probed against
    bool side_ready = false;
    int aux_priority = 0;
    uint32_t* probe_ws = nullptr;        // {flag, saw}
    uint32_t next_set = 0;
    uint32_t sort_max_tiles = 0;
    PendEntry* pend = nullptr;
    ChainRec* chain = nullptr; // k_eval_general: per-wave hand-over records
    uint32_t chain_seq = 0;
    uint32_t* loaded = nullptr; // k_eval_sorted<DIRECT>: per-wave "cells read" flags
    int eval_items = 0;         // sorted positions per lane in k_eval_sorted (0: chosen per batch)
    bool eval_lean = true;      // k_eval_sorted_lean for decisions-only batches (TCGPU_EVAL_LEAN=0: the general kernel)
    bool stop_events = true;    // events ride on kernels' completion signals instead of marker packets (TCGPU_STOP_EVENTS=0: hipEventRecord)
    bool prefill_on = true;     // TC_B_OUTPUTS_IDLE batches: decision bytes preset on the grouping stream (TCGPU_PREFILL=0: off)
    uint32_t* fill_hint_host = nullptr; // pinned: "most decisions of a recent batch were allowed", written by the evaluation, read here without waiting
    uint32_t* fill_hint_dev = nullptr;  // the same word as the device addresses it
    bool debug_nostore = false; // TCGPU_DEBUG_NO_DECISION_STORE=1: MEASUREMENT ONLY -- the lean kernel skips its decision bytes (wrong results)
    uint32_t loaded_seq = 0;
    // bounds over the registered rate plans (for all_runs_regular)
    int64_t cls_min_ei = INT64_MAX, cls_max_ei = 0, cls_min_dvt = INT64_MAX, cls_max_dvt = 0;
    uint32_t* pend_count = nullptr;
    uint8_t* allowed_tmp = nullptr;
    StoreOpResult* op_result = nullptr;
    OneResult* one_result = nullptr; // tc_rate_limit: the single request's result

    // staging for host-pointer batches (lazy)
    struct Stage {
        uint32_t* slot = nullptr;
        int64_t* in[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        uint8_t* allowed = nullptr;
        uint64_t* bits = nullptr;
        int64_t* out[4] = {nullptr, nullptr, nullptr, nullptr};
        int64_t* result4 = nullptr;
        tc_decision* decisions = nullptr;
        uint32_t* order = nullptr;
        uint8_t* status = nullptr;
    } stage;

    // small host-pointer batches (k_small_batch): one pinned block the kernel reads its inputs from and writes
    // its results to (the caller's arrays are copied in and out by the host)
    uint8_t* small_io = nullptr;  // host address
    uint8_t* small_io_dev = nullptr; // the same block as the device addresses it
    size_t small_io_bytes = 0;
    bool small_off = false;       // TCGPU_NO_SMALL_BATCH=1: always take the big pipeline

    // TC_B_ASYNC host batches still in flight, oldest first: one event per batch, recorded behind its last copy
    std::deque<hipEvent_t> async_done;
    std::vector<hipEvent_t> async_pool;

    uint64_t batches = 0; // TC_CNT_BATCHES is kept on the host
    uint32_t* poison_host = nullptr; // pinned: raised by tc::invariant_failed; every ABI call checks it (TC_E_INVARIANT)
    bool debug_break_wait = false;   // tc_debug_break_wait
    uint32_t fault_countdown = 0; // tc_debug_fail_copy: the n-th staging copy from now fails (error-path tests)
    uint32_t* route_ws = nullptr; // tc_route_batch scratch (lazy): per stream it may run on: tile counts per destination
    size_t route_ws_words = 0;    // words of one of them
    uint32_t next_route = 0;
    uint32_t route_seq = 0;       // sequence number of the one-pass router's look-back words

    // bucket path (bucket_path.hpp): uniform batches are partitioned by key range and ranked per bucket instead of
    // sorted.  Such a batch is enqueued on BOTH paths; the partition's largest bucket (a word in device memory,
    // the gate) decides on the device which of the two runs, so a skewed batch never waits for the host.
    bool bp_ok = false;      // the key space fits the path (<= bp::MAX_BUCKETS buckets)
    int bp_lb = 0;           // log2(slots per bucket)
    uint32_t bp_nbk = 0;
    uint32_t bp_max_n = 0;   // largest batch the scratch is sized for (<= bp::MAX_N)
    uint32_t bp_min_n = 0;   // smaller batches are sorted (TCGPU_BUCKET_MIN_N)
    uint32_t bp_skew = 1024; // longest bucket the path takes on (TCGPU_BUCKET_SKEW)
    bool bp_piped = false;   // also for TC_B_INPUTS_READY batches (TCGPU_BUCKET_PIPED=1; measured slower)
    uint32_t* bp_gate_host = nullptr; // pinned: the gate of a recent batch, mirrored by the device (a hint, never waited for)
    uint32_t bp_backoff = 0;  // batches left during which the bucket path is not even tried (the stream looked skewed)
    uint32_t bp_backoff_len = BP_BACKOFF; // (TCGPU_BUCKET_BACKOFF; 0: always try)
    PendEntry* bp_park = nullptr; // parked cell stores of long buckets, one entry per request

    // string-key mode (TC_CFG_KEY_MODE): device hash table + per-batch resolution scratch
    bool key_mode = false;
    kt::Table kt;
    void* kt_block = nullptr;        // one allocation backing every kt.* array
    kt::RetiredRec* retired = nullptr; // key mode + TC_CFG_TRACK_DENIED: denial counts of keys without a slot
    uint32_t *k_slot = nullptr, *k_state = nullptr, *k_aux = nullptr; // max_batch each
    uint32_t* k_claim = nullptr;     // keys first seen in the batch, per k_probe block
    uint64_t* k_hash = nullptr;
    // Key stages mutate the key table and share the scratch above, so they run one after another in
    // call order: on the key stream for TC_B_INPUTS_READY device batches (overlapping the grouping and
    // evaluation of earlier batches), else on the main stream; the two events hand the order across.
    hipStream_t key_stream = nullptr;
    hipEvent_t k_done = nullptr, m_done = nullptr;
    bool k_busy = false, m_busy = false;
    hipEvent_t wait_before_sort = nullptr; // piped key batch: the auxiliary sort waits for its key stage
    uint8_t* k_stage_bytes = nullptr; // host-pointer batches: staged key arena
    size_t k_stage_bytes_cap = 0;
    uint32_t* k_stage_off = nullptr;  // max_batch + 1

    // optional per-stage HIP-event timing (tc_profile_*): a (begin, end) event pair per
    // kernel, recorded on the stream the kernel is launched on
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev; // 2 per record
    std::vector<int> prof_stage;
    size_t prof_used = 0;            // records
    double prof_ms[TC_STAGE_COUNT] = {0};
    uint64_t prof_calls[TC_STAGE_COUNT] = {0};

    std::string err;
};

static hipStream_t cur_stream(tc_engine* e) {
    if (e->user_stream) return e->user_stream;
    if (!e->own_stream && hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return e->own_stream;
}

// begin / end of one kernel of `stage` on stream `s` (no-ops unless profiling)
static void prof_begin(tc_engine* e, int stage, hipStream_t s) {
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
static void prof_end(tc_engine* e, hipStream_t s) {
    if (!e->prof_on || e->prof_used == e->prof_stage.size()) return;
    (void)hipEventRecord(e->prof_ev[2 * e->prof_used + 1], s);
    e->prof_used++;
}

// every host <-> device copy of a batch goes through here, so that tests can make one of them fail
static hipError_t copy_async(tc_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (e->fault_countdown && --e->fault_countdown == 0) return hipErrorInvalidValue;
    return hipMemcpyAsync(dst, src, bytes, kind, st);
}

#define TC_HIP(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _rc = (call);                                                                     \
        if (_rc != hipSuccess) {                                                                     \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_rc);                           \
            return TC_E_HIP;                                                                         \
        }                                                                                            \
    } while (0)

// A kernel launch whose completion is `stop` (hipExtLaunchKernelGGL: the event rides on the dispatch packet's own
// completion signal -- no marker packet behind the kernel, which the next kernel of the stream would wait for), or a
// plain launch when stop == nullptr.
#define TC_LAUNCH(stop, kernel, grid, block, lds, stream, ...)                                               \
    do {                                                                                                     \
        if (stop) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, stop, 0, __VA_ARGS__);    \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                              \
    } while (0)

static int fail(tc_engine* e, int code, const char* msg) {
    if (e) e->err = msg;
    return code;
}

extern "C" uint32_t tc_abi_version(void) { return TCGPU_ABI_VERSION; }

extern "C" const char* tc_last_error(const tc_engine* e) { return e ? e->err.c_str() : "null engine"; }

static size_t sort_ws_words(uint32_t max_tiles) { return rs::workspace_words(max_tiles); }

// the device-side address of the poison word goes into the counter block, POISON_PTR_WORDS behind the violation counter
// (tc::invariant_failed finds it there: no kernel carries an extra argument for something that never happens)
static int publish_poison_ptr(tc_engine* e) {
    void* dv = nullptr;
    TC_HIP(e, hipHostGetDevicePointer(&dv, e->poison_host, 0));
    const unsigned long long v = (unsigned long long)(uintptr_t)dv;
    TC_HIP(e, hipMemcpy(e->counters + (TC_CNT_COUNT + 1) + 3 + tc::POISON_PTR_WORDS, &v, sizeof v, hipMemcpyHostToDevice));
    return TC_E_OK;
}
#define TC_TRY_EARLY(call)                \
    do {                                  \
        int _trc = (call);                \
        if (_trc != TC_E_OK) return _trc; \
    } while (0)
// sticky: once a kernel has flagged a broken invariant the engine answers nothing else
static int poisoned(tc_engine* e) {
    if (e->poison_host && *(volatile uint32_t*)e->poison_host != 0u) {
        e->err = "an internal invariant failed on the device (a wait gave up, or a closed form met a state it was proven not to meet): "
                 "results and state are undefined -- destroy the engine";
        return TC_E_INVARIANT;
    }
    return TC_E_OK;
}
#define TC_CHECK_POISON(e)              \
    do {                                \
        int _prc = poisoned(e);         \
        if (_prc != TC_E_OK) return _prc; \
    } while (0)

static int engine_alloc(tc_engine* e) {
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
    if (const char* d = getenv("TCGPU_DEBUG_NO_DECISION_STORE")) e->debug_nostore = atoi(d) != 0;
    if (const char* d = getenv("TCGPU_PREFILL")) e->prefill_on = atoi(d) != 0;
    {
        TC_HIP(e, hipHostMalloc((void**)&e->fill_hint_host, 64, hipHostMallocDefault));
        *e->fill_hint_host = 1u;
        void* dv = nullptr;
        TC_HIP(e, hipHostGetDevicePointer(&dv, e->fill_hint_host, 0));
        e->fill_hint_dev = (uint32_t*)dv;
    }
    if (const char* d = getenv("TCGPU_SORT_ITEMS_PIPED")) {
        const int v = atoi(d);
        if (v == 8 || v == 16 || v == 32) e->sort_items_piped = v;
    }
    if (const char* d = getenv("TCGPU_NO_SMALL_BATCH")) e->small_off = atoi(d) != 0;
    const char* pe = getenv("TCGPU_AUX_PRIORITY");
    const bool aux_high = pe && atoi(pe) != 0; // default: lowest priority (measured ~1 % better: the evaluation kernel is the critical path)
    // the evaluation kernel on the main stream is the critical path of the pipeline: grouping runs at
    // the lowest priority and fills what the evaluation leaves free (TCGPU_AUX_PRIORITY=1 flips it)
    e->n_aux_want = e->n_aux;
    e->aux_priority = aux_high ? prio_hi : prio_lo;
    TC_HIP(e, hipMalloc(&e->probe_ws, 2 * sizeof(uint32_t)));
    // (the grouping streams themselves are created by ensure_side_streams, against the actual main stream)
    for (uint32_t si = 0; si < e->depth; ++si) {
        tc_engine::SortSet& ss = e->sets[si];
        TC_HIP(e, hipMalloc(&ss.elem_a, mb * sizeof(uint64_t)));
        TC_HIP(e, hipMalloc(&ss.elem_b, mb * sizeof(uint64_t)));
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
    TC_HIP(e, hipMalloc(&e->op_result, sizeof(StoreOpResult)));
    TC_HIP(e, hipMalloc(&e->one_result, sizeof(OneResult)));
    TC_HIP(e, hipStreamSynchronize((hipStream_t)0)); // set-up runs on the null stream: no private stream yet
    return TC_E_OK;
}

// ---- string-key mode ----------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int key_mode_alloc(tc_engine* e, uint64_t key_arena_bytes) {
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
                 o_free = take(cap * 4), o_misc = take(64), o_tombs = take(kt::TOMB_SHARDS * 4);
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
    // (+32, +40: the overflow compaction's flag words)
    t.free_slots = (uint32_t*)(base + o_free);
    t.capacity = (uint32_t)cap;
    t.retired = nullptr;
    t.denied = e->denied;
    if (e->denied) { // denial counts of keys that lose their slot (kt::RetiredRec)
        TC_HIP(e, hipMalloc(&e->retired, (size_t)kt::RETIRED_CAP * sizeof(kt::RetiredRec)));
        TC_HIP(e, hipMemsetAsync(e->retired, 0, (size_t)kt::RETIRED_CAP * sizeof(kt::RetiredRec), (hipStream_t)0));
        t.retired = e->retired;
    }
    hipLaunchKernelGGL(kt::k_init_free, dim3(std::min<uint64_t>(nblocks(cap), 2048)), dim3(kt::THREADS), 0, (hipStream_t)0,
                       t.free_slots, t.bound, (uint32_t)cap);
    const int top = (int)cap;
    TC_HIP(e, hipMemcpyAsync(t.free_top, &top, sizeof top, hipMemcpyHostToDevice, (hipStream_t)0));
    TC_HIP(e, hipMalloc(&e->k_slot, mb * 4));
    for (uint32_t si = 0; si < e->depth; ++si) TC_HIP(e, hipMalloc(&e->sets[si].k_slot, mb * 4));
    TC_HIP(e, hipEventCreateWithFlags(&e->k_done, hipEventDisableTiming));
    TC_HIP(e, hipEventCreateWithFlags(&e->m_done, hipEventDisableTiming));
    TC_HIP(e, hipMalloc(&e->k_state, mb * 4));
    TC_HIP(e, hipMalloc(&e->k_aux, mb * 4));
    TC_HIP(e, hipMalloc(&e->k_claim, ((size_t)nblocks(mb) + 1) * 4));
    TC_HIP(e, hipMalloc(&e->k_hash, mb * 8));
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

// The grouping streams (and the key stream) must not share a hardware queue with the main stream or
// with each other: two active streams on one queue serialise, which costs the pipeline half its
// throughput.  Which queue a new stream lands on depends on everything the
// process created before (torch, RCCL, other engines), so candidates are created and PROBED against
// the main stream in use; the ones that do not run concurrently are dropped.  ~0.5 ms, once per main stream.
static int ensure_side_streams(tc_engine* e) {
    hipStream_t m = cur_stream(e);
    if (e->side_ready && e->side_for == m) return TC_E_OK;
    TC_HIP(e, hipStreamSynchronize(m));
    for (hipStream_t& a : e->aux)
        if (a) {
            TC_HIP(e, hipStreamSynchronize(a));
            (void)hipStreamDestroy(a);
            a = nullptr;
        }
    if (e->key_stream) {
        TC_HIP(e, hipStreamSynchronize(e->key_stream));
        (void)hipStreamDestroy(e->key_stream);
        e->key_stream = nullptr;
    }
    const uint32_t want = e->n_aux_want + (e->key_mode ? 1u : 0u);
    std::vector<hipStream_t> good, bad;
    for (int c = 0; c < 16 && good.size() < want; ++c) {
        hipStream_t s = nullptr;
        int rc = TC_E_OK;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, e->aux_priority) != hipSuccess) rc = fail(e, TC_E_HIP, "hipStreamCreateWithPriority failed");
        bool ok = false;
        if (rc == TC_E_OK) rc = streams_concurrent(e, m, s, &ok);
        for (size_t g = 0; rc == TC_E_OK && ok && g < good.size(); ++g) rc = streams_concurrent(e, good[g], s, &ok);
        if (rc != TC_E_OK) {
            if (s) (void)hipStreamDestroy(s);
            for (hipStream_t x : good) (void)hipStreamDestroy(x);
            for (hipStream_t x : bad) (void)hipStreamDestroy(x);
            return rc;
        }
        (ok ? good : bad).push_back(s);
    }
    for (hipStream_t s : bad) (void)hipStreamDestroy(s);
    size_t gi = 0;
    if (e->key_mode && !good.empty()) e->key_stream = good[gi++];
    e->n_aux = 0;
    for (; gi < good.size() && e->n_aux < (uint32_t)AUX_MAX; ++gi) e->aux[e->n_aux++] = good[gi];
    e->next_aux = 0;
    e->side_for = m;
    e->side_ready = true;
    return TC_E_OK; // n_aux == 0: no free hardware queue, TC_B_INPUTS_READY batches run in order on the main stream
}

// keys (device arena) -> out_slot[0..n): found slot, freshly bound slot, or NO_SLOT.
// on_key_stream: issue on the key stream (the caller vouched for the inputs), else on the main stream.
static int resolve_keys_device(tc_engine* e, const uint8_t* d_bytes, const uint32_t* d_off, uint32_t n, bool insert,
                               uint32_t* out_slot, bool on_key_stream) {
    hipStream_t s = on_key_stream ? e->key_stream : cur_stream(e);
    if (on_key_stream) {
        if (e->m_busy) TC_HIP(e, hipStreamWaitEvent(s, e->m_done, 0));
    } else {
        if (e->k_busy) TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
    }
    const dim3 grid(nblocks(n)), block(kt::THREADS);
    prof_begin(e, TC_STAGE_HASH, s);
    if (insert) {
        hipLaunchKernelGGL(kt::k_probe<true>, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, e->k_state, e->k_aux,
                           e->k_hash, e->k_claim);
        hipLaunchKernelGGL(kt::k_claim_scan, dim3(1), dim3(1024), 0, s, e->k_claim, grid.x);
        hipLaunchKernelGGL(kt::k_bind, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, e->k_state, e->k_aux, e->k_hash,
                           e->k_claim);
        hipLaunchKernelGGL(kt::k_follow, grid, block, 0, s, n, out_slot, e->k_state, e->k_aux, e->kt, e->k_claim, grid.x,
                           e->counters + TC_CNT_KEYS_INSERTED);
    } else {
        hipLaunchKernelGGL(kt::k_probe<false>, grid, block, 0, s, e->kt, d_bytes, d_off, n, out_slot, e->k_state,
                           e->k_aux, e->k_hash, (uint32_t*)nullptr);
    }
    prof_end(e, s);
    TC_HIP(e, hipGetLastError());
    if (on_key_stream) {
        TC_HIP(e, hipEventRecord(e->k_done, s));
        e->k_busy = true;
        e->m_busy = false;
    } else {
        TC_HIP(e, hipEventRecord(e->m_done, s));
        e->m_busy = true;
        e->k_busy = false;
    }
    return TC_E_OK;
}

// host key arena -> staged on device
static int stage_keys(tc_engine* e, const uint8_t* key_bytes, const uint32_t* key_off, uint64_t n,
                      const uint8_t** d_bytes, const uint32_t** d_off) {
    const size_t total = key_off[n];
    if (total > e->k_stage_bytes_cap) {
        if (e->k_stage_bytes) (void)hipFree(e->k_stage_bytes);
        e->k_stage_bytes = nullptr;
        const size_t want = std::max<size_t>(total * 2, 1 << 16);
        TC_HIP(e, hipMalloc(&e->k_stage_bytes, want));
        e->k_stage_bytes_cap = want;
    }
    if (total) TC_HIP(e, copy_async(e, e->k_stage_bytes, key_bytes, total, hipMemcpyHostToDevice, cur_stream(e)));
    TC_HIP(e, copy_async(e, e->k_stage_off, key_off, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream(e)));
    *d_bytes = e->k_stage_bytes ? e->k_stage_bytes : (const uint8_t*)e->k_stage_off;
    *d_off = e->k_stage_off;
    return TC_E_OK;
}

// one key (host bytes) -> slot on the host; insert binds a fresh slot if unseen
static int resolve_one_key(tc_engine* e, const uint8_t* key, size_t key_len, bool insert, uint32_t* slot) {
    if (key_len > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "key too long");
    const uint32_t off[2] = {0u, (uint32_t)key_len};
    const uint8_t* d_bytes;
    const uint32_t* d_off;
    int rc = stage_keys(e, key, off, 1, &d_bytes, &d_off);
    if (rc != TC_E_OK) return rc;
    rc = resolve_keys_device(e, d_bytes, d_off, 1, insert, e->k_slot, false);
    if (rc != TC_E_OK) return rc;
    TC_HIP(e, hipMemcpyAsync(slot, e->k_slot, sizeof(uint32_t), hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    return TC_E_OK;
}

// rebuild the key table if tombstones fill more than 1/4 of it (checked on the device)
static int rebuild_key_table_if_due(tc_engine* e) {
    kt::Table& t = e->kt;
    hipStream_t s = cur_stream(e);
    uint32_t* flag = t.error_flag + 1; // spare word of the table's misc block
    const dim3 grid(std::min<uint64_t>(nblocks(t.nb_mask + 1), 4096)), block(kt::THREADS);
    hipLaunchKernelGGL(kt::k_rebuild_decide, dim3(1), dim3(64), 0, s, t, flag);
    hipLaunchKernelGGL(kt::k_rebuild_clear, grid, block, 0, s, t, flag);
    hipLaunchKernelGGL(kt::k_reinsert, dim3(std::min<uint64_t>(nblocks(t.capacity), 2048)), block, 0, s, t, flag);
    // ... and compact the overflow arena (keys longer than 48 bytes) once more than half of it is handed out
    unsigned long long* oflag = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(t.overflow_used) + 32);
    hipLaunchKernelGGL(kt::k_overflow_decide, dim3(1), dim3(64), 0, s, t, oflag);
    hipLaunchKernelGGL(kt::k_overflow_compact, dim3(std::min<uint64_t>(nblocks(t.capacity), 2048)), block, 0, s, t, oflag);
    hipLaunchKernelGGL(kt::k_overflow_swap, dim3(1), dim3(64), 0, s, t, oflag);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
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
    for (hipStream_t a : e->aux)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    for (tc_engine::SortSet& ss : e->sets) {
        if (ss.sorted) (void)hipEventDestroy(ss.sorted);
        if (ss.consumed) (void)hipEventDestroy(ss.consumed);
        void* sp[] = {ss.bp_scratch, ss.elem_a, ss.elem_b, ss.ws, ss.h_slot, ss.h_in[0], ss.h_in[1], ss.h_in[2], ss.h_in[3], ss.h_in[4], ss.h_key_bytes, ss.h_key_off};
        for (void* p : sp)
            if (p) (void)hipFree(p);
    }
    void* ptrs[] = {e->route_ws, e->bp_park, e->cells, e->tat8, e->rate_id, e->classes, e->denied, e->topk_ws, e->probe_ws, e->counters, e->pend, e->chain, e->loaded, e->pend_count,
                    e->allowed_tmp, e->op_result, e->one_result, e->stage.slot, e->stage.in[0], e->stage.in[1], e->stage.in[2],
                    e->stage.in[3], e->stage.in[4], e->stage.allowed, e->stage.bits, e->stage.out[0],
                    e->stage.out[1], e->stage.out[2], e->stage.out[3], e->stage.status, e->stage.result4, e->stage.decisions, e->stage.order};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->key_stream) {
        (void)hipStreamSynchronize(e->key_stream);
        (void)hipStreamDestroy(e->key_stream);
    }
    if (e->bp_gate_host) (void)hipHostFree(e->bp_gate_host);
    if (e->fill_hint_host) (void)hipHostFree(e->fill_hint_host);
    if (e->poison_host) (void)hipHostFree(e->poison_host);
    if (e->k_done) (void)hipEventDestroy(e->k_done);
    if (e->m_done) (void)hipEventDestroy(e->m_done);
    for (tc_engine::SortSet& ss : e->sets)
        if (ss.k_slot) (void)hipFree(ss.k_slot);
    void* kptrs[] = {e->retired, e->kt_block, e->k_slot, e->k_state, e->k_aux, e->k_claim, e->k_hash, e->k_stage_bytes, e->k_stage_off};
    for (void* p : kptrs)
        if (p) (void)hipFree(p);
    if (e->small_io) (void)hipHostFree(e->small_io);
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->async_done) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->async_pool) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

extern "C" int tc_engine_set_stream(tc_engine* e, void* hip_stream) {
    if (!e) return TC_E_INVALID_ARG;
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
static int intern_class(tc_engine* e, int64_t burst, int64_t count, int64_t period, bool* grew) {
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

static int upload_classes(tc_engine* e) {
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
    for (uint64_t i = 0; i < n; ++i) {
        if (i && max_burst[i] == max_burst[i - 1] && count_per_period[i] == count_per_period[i - 1] && period[i] == period[i - 1]) {
            ids[i] = (uint16_t)last_id; // runs of one plan are the common case
            continue;
        }
        last_id = intern_class(e, max_burst[i], count_per_period[i], period[i], &grew);
        if (last_id < 0) return fail(e, TC_E_UNSUPPORTED, "more than 65535 distinct rate plans registered");
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

// device staging for host-pointer batches, allocated per array on first use
template <class T>
static int stage_need(tc_engine* e, T*& p, size_t count) {
    if (!p) TC_HIP(e, hipMalloc(&p, count * sizeof(T)));
    return TC_E_OK;
}
#define TC_TRY(call)                  \
    do {                              \
        int _trc = (call);            \
        if (_trc != TC_E_OK) return _trc; \
    } while (0)

// output staging for the arrays `b` asks for; `d` gets the device pointers
static int stage_outputs(tc_engine* e, const tc_batch& b, tc_batch& d) {
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
static int copy_outputs_back(tc_engine* e, const tc_batch& b, hipStream_t s, bool by_kernel = false) {
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

// stable sort of (slot, index) by slot in scratch set `ss`, issued on stream `s`;
// returns the buffer holding the result
static const uint64_t* sort_by_slot(tc_engine* e, tc_engine::SortSet& ss, hipStream_t s, const uint32_t* d_slot, uint32_t n,
                                    bool piped, const uint32_t* gate = nullptr, uint32_t gate_min = 0, hipEvent_t stop_last = nullptr,
                                    uint8_t* fill = nullptr, uint32_t fill_value = 0) {
    const uint32_t cap = (uint32_t)e->capacity;
    const int bits = std::max(1, bit_width_u64(e->capacity)); // the sentinel key `capacity` must fit
    const int passes = (bits + 7) / 8;
    const int items = piped ? e->sort_items_piped : SORT_ITEMS;
    const uint32_t tile = rs::THREADS * (uint32_t)items;
    const uint32_t tiles = (n + tile - 1) / tile;
    rs::Workspace ws = rs::carve(ss.ws, ss.hist_parity, e->sort_max_tiles);
    ws.violations = e->counters + (TC_CNT_COUNT + 1) + 3;
    ss.hist_parity ^= 1u;
    prof_begin(e, TC_STAGE_PREP, s);
    hipLaunchKernelGGL(rs::k_hist<rs::HIST_THREADS>, dim3(rs::HIST_BLOCKS), dim3(rs::HIST_THREADS), 0, s, d_slot, n, cap, passes, ws, tiles, gate, gate_min, fill, fill_value);
    prof_end(e, s);
    uint64_t* bufs[2] = {ss.elem_a, ss.elem_b};
    const uint64_t* in = nullptr;
    for (int p = 0; p < passes; ++p) {
        uint64_t* out = bufs[p & 1];
        prof_begin(e, TC_STAGE_SORT, s); // one record per pass: the stage average is per kernel launch
        hipEvent_t stop = (p + 1 == passes && !e->prof_on) ? stop_last : nullptr;
#define TC_PASS(IT, FI) \
    TC_LAUNCH(stop, (rs::k_onesweep<IT, FI>), dim3(tiles), dim3(rs::THREADS), 0, s, (FI) ? d_slot : (const uint32_t*)nullptr, \
              (FI) ? (const uint64_t*)nullptr : in, out, n, cap, p, ws, gate, gate_min)
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
        prof_end(e, s);
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
        if constexpr (ITEMS <= 2) {
            TC_LAUNCH(stop, (k_eval_sorted_lean<ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->loaded, seq, gate, gate_min, e->fill_hint_dev);
            return;
        }
    }
    if (full && direct) TC_LAUNCH(stop, (k_eval_sorted<true, true, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else if (full) TC_LAUNCH(stop, (k_eval_sorted<true, false, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else if (direct) TC_LAUNCH(stop, (k_eval_sorted<false, true, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
    else TC_LAUNCH(stop, (k_eval_sorted<false, false, ITEMS, FIXED>), grid, block, 0, s, p, sorted, e->pend, e->pend_count, e->loaded, seq, gate, gate_min);
}
static int eval_items_of(const tc_engine* e, bool piped) { return e->eval_items ? e->eval_items : (piped ? 2 : 4); }
// LEAN: decisions only, direct stores, nothing asked for but the `allowed` bytes in request order (k_eval_sorted_lean)
static bool lean_applies(const tc_engine* e, bool full, bool direct, bool piped, const Params& p) {
    return e->eval_lean && !full && direct && eval_items_of(e, piped) <= 2 && p.allowed && !p.status && !p.limit && !p.order && !p.row_bits;
}
static void launch_eval_sorted(tc_engine* e, bool full, bool direct, bool piped, uint32_t n, hipStream_t s, const Params& p,
                               const uint64_t* sorted, uint32_t seq, const uint32_t* gate = nullptr, uint32_t gate_min = 0,
                               hipEvent_t stop = nullptr) {
    const int items = eval_items_of(e, piped);
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

// input arrays of a TC_B_ASYNC host batch, staged inside run_slots_device on the stream that groups the batch
struct HostIn {
    const uint32_t* slot = nullptr;
    const int64_t* col[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // burst, count, period, quantity, now
};

// host arrays -> the scratch set's staging columns, enqueued on `st`; `p` gets the device pointers
static int stage_host_inputs(tc_engine* e, tc_engine::SortSet& ss, const HostIn& hin, uint32_t n, hipStream_t st, Params& p,
                             const uint32_t** d_slot) {
    const uint64_t mb = e->max_batch;
    if (hin.slot) { // (key batches arrive with their slots already resolved on the device)
        if (!ss.h_slot) TC_HIP(e, hipMalloc(&ss.h_slot, mb * sizeof(uint32_t)));
        TC_HIP(e, copy_async(e, ss.h_slot, hin.slot, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st)); // SDMA when pinned
        *d_slot = ss.h_slot;
        p.slot = ss.h_slot;
    }
    const int64_t** dst[5] = {&p.burst, &p.count, &p.period, &p.q, &p.now};
    for (int j = 0; j < 5; ++j) {
        if (!hin.col[j]) continue;
        if (!ss.h_in[j]) TC_HIP(e, hipMalloc(&ss.h_in[j], mb * sizeof(int64_t)));
        TC_HIP(e, copy_async(e, ss.h_in[j], hin.col[j], (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
        *dst[j] = ss.h_in[j];
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
static int run_slots_device(tc_engine* e, const tc_batch& b, const HostIn* hin = nullptr) {
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
    if (e->debug_break_wait) {
        p.flags |= F_DEBUG_NO_ANNOUNCE;
        e->debug_break_wait = false; // one batch
    }
    p.rate_id = e->rate_id;
    p.classes = e->classes;
    p.uniform_class = e->uniform_id;
    p.denied = e->denied;
    p.row_bits = nullptr;
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
        }
        // (the per-request columns of a TC_B_ASYNC host batch are still to be staged: they are in `hin`)
        auto column = [&](const int64_t* dev, int j) { return dev != nullptr || (hin && hin->col[j] != nullptr); };
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
        bool eligible = direct && e->bp_ok && n >= e->bp_min_n && n <= e->bp_max_n && !p.order && (!piped || e->bp_piped);
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
            if (e->wait_before_sort) TC_HIP(e, hipStreamWaitEvent(ax, e->wait_before_sort, 0)); // its key stage
            if (hin) TC_TRY(stage_host_inputs(e, ss, *hin, n, ax, p, &d_slot)); // PCIe transfer overlaps earlier evaluations
            if (b.n_segments) TC_TRY(gather_segments(e, ss, b, n, ax, p, &d_slot));
            if (bucketed && !bucket_partition(e, ss, ax, d_slot, n)) return fail(e, TC_E_HIP, "bucket partition launch failed");
            const bool ride = e->stop_events && !e->prof_on; // `sorted` rides on the last pass's own completion signal
            // TC_B_OUTPUTS_IDLE: nothing enqueued earlier touches this call's `allowed` bytes, so they are preset here, on
            // the grouping stream, to what most decisions of a recent batch were, and the evaluation only stores the
            // others -- 1 Mi one-byte stores scattered over the batch were 6 of its 53 us
            uint8_t* fill = nullptr;
            uint32_t fill_value = 0;
            if ((b.flags & TC_B_OUTPUTS_IDLE) && e->prefill_on && !hin && !bucketed && uniform && lean_applies(e, full, direct, true, p) &&
                p.allowed == b.allowed) {
                fill = b.allowed;
                fill_value = *(volatile uint32_t*)e->fill_hint_host & 1u;
                p.flags |= fill_value ? F_PREFILL1 : F_PREFILL0;
            }
            sorted = sort_by_slot(e, ss, ax, d_slot, n, true, gate, e->bp_skew, ride ? ss.sorted : nullptr, fill, fill_value);
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
            sorted = sort_by_slot(e, ss, s, d_slot, n, false, gate, e->bp_skew);
        }
        if (bucketed) bucket_eval(e, ss, s, p, full);
        prof_begin(e, TC_STAGE_EVAL, s);
        bool consumed_rides = false;
        if (uniform) {
            uint32_t seq = 0u;
            if (direct) {
                if (++e->loaded_seq == 0u) e->loaded_seq = 1u;
                seq = e->loaded_seq;
            }
            // (direct: the evaluation is the last reader of the set -- `consumed` can ride on its completion signal)
            consumed_rides = direct && e->stop_events && !e->prof_on;
            launch_eval_sorted(e, full, direct, piped, n, s, p, sorted, seq, gate, e->bp_skew, consumed_rides ? ss.consumed : nullptr);
            prof_end(e, s);
            if (!direct) {
                prof_begin(e, TC_STAGE_COMMIT, s);
                hipLaunchKernelGGL(k_commit_list, dim3(64), block, 0, s, e->pend, e->pend_count, e->cells, e->tat8);
                prof_end(e, s);
            }
        } else {
            if (++e->chain_seq >= 0x7FFFFFFFu) e->chain_seq = 1u; // 0 = "never written"; 31 bits: the spec word carries a flag
            if (full) hipLaunchKernelGGL(k_eval_general<true>, grid, block, 0, s, p, sorted, e->chain, e->chain_seq);
            else hipLaunchKernelGGL(k_eval_general<false>, grid, block, 0, s, p, sorted, e->chain, e->chain_seq);
            prof_end(e, s);
        }
        e->wait_before_sort = nullptr;
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
static int run_slots_host_staged(tc_engine* e, const tc_batch& b) {
    const uint64_t n = b.n;
    hipStream_t s = cur_stream(e);
    tc_batch d = b;
    d.flags |= TC_B_DEVICE_PTRS;
    d.flags &= ~(TC_B_INPUTS_READY | TC_B_ASYNC);
    d.slot = e->stage.slot;
    d.key_bytes = nullptr;
    d.key_off = nullptr;
    const int64_t* hin[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    const int64_t** din[5] = {&d.max_burst, &d.count_per_period, &d.period, &d.quantity, &d.now_ns};
    for (int j = 0; j < 5; ++j) {
        if (hin[j]) {
            TC_TRY(stage_need(e, e->stage.in[j], e->max_batch));
            TC_HIP(e, copy_async(e, e->stage.in[j], hin[j], n * sizeof(int64_t), hipMemcpyHostToDevice, s));
            *din[j] = e->stage.in[j];
        }
    }
    TC_TRY(stage_outputs(e, b, d));
    TC_TRY(run_slots_device(e, d));
    TC_TRY(copy_outputs_back(e, b, s));
    TC_HIP(e, hipStreamSynchronize(s));
    return poisoned(e); // (a synchronous batch whose kernels flagged an invariant fails itself, not the next call)
}

// results back to the caller's arrays behind the evaluation + the batch's completion event
static int finish_async(tc_engine* e, const tc_batch& b) {
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
    TC_TRY(stage_outputs(e, b, d));
    TC_TRY(run_slots_device(e, d, &hin));
    return finish_async(e, b);
}

// ---- small host-pointer batches: one launch (k_small_batch) ----------------------------------------
constexpr size_t SMALL_KEY_BYTES = 128 * 1024; // string mode: batches with a larger key arena take the big pipeline

static bool small_batch_applies(const tc_engine* e, const tc_batch& b) {
    if (e->small_off || b.n > (uint64_t)SMALL_MAX) return false;
    if (b.flags & (TC_B_DEVICE_PTRS | TC_B_ASYNC | TC_B_GROUPED_OUTPUT)) return false;
    if (b.allowed_bits) return false; // (packed bits come out of the big pipeline only)
    if (e->key_mode && b.key_off[b.n] > SMALL_KEY_BYTES) return false;
    return true;
}

// Host-pointer batch of at most SMALL_MAX requests, slot or string mode.  Same results as the big pipeline
// (the run walker is the plain sequence); ordering against key stages on the key stream as tc_rate_limit.
static int run_small_batch(tc_engine* e, const tc_batch& b) {
    const size_t n = b.n;
    if (!e->small_io) {
        const size_t bytes = 64 + SMALL_KEY_BYTES + (size_t)SMALL_MAX * (4 + 4 + 5 * 8 + 1 + 1 + 4 * 8 + 32 + 32) + 16 * 16;
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
    if (e->key_mode) {
        const size_t total = b.key_off[n];
        const size_t o_off = take((n + 1) * 4), o_bytes = take(total + 16);
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
                       (uint32_t*)d, e->counters + TC_CNT_KEYS_INSERTED);
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
    if (b.flags & TC_B_DEVICE_PTRS) {
        if (b.flags & TC_B_ASYNC) return fail(e, TC_E_INVALID_ARG, "TC_B_ASYNC is for host-pointer batches (device-pointer batches are asynchronous anyway)");
        rc = run_slots_device(e, b);
    } else if (b.flags & TC_B_ASYNC) {
        rc = run_slots_host_async(e, b);
    } else if (small_batch_applies(e, b)) {
        rc = run_small_batch(e, b);
    } else {
        // host pointers: stage in, run, stage out, synchronise
        TC_TRY(stage_need(e, e->stage.slot, e->max_batch));
        TC_HIP(e, copy_async(e, e->stage.slot, b.slot, b.n * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream(e)));
        rc = run_slots_host_staged(e, b);
    }
    // fixed layout: once a request has been decided the plans can no longer change (a batch that was rejected, or
    // whose staging failed, applied nothing and seals nothing)
    if (e->fixed && rc == TC_E_OK) e->sealed = true;
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
    const size_t total = b.key_off[n];
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
    if (total) TC_HIP(e, copy_async(e, ss.h_key_bytes, b.key_bytes, total, hipMemcpyHostToDevice, ks));
    TC_HIP(e, copy_async(e, ss.h_key_off, b.key_off, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ks));
    TC_TRY(resolve_keys_device(e, ss.h_key_bytes, ss.h_key_off, n, true, ss.k_slot, piped));
    tc_batch d = b;
    d.flags = (b.flags & ~(TC_B_ASYNC | TC_B_INPUTS_READY)) | TC_B_DEVICE_PTRS | (piped ? TC_B_INPUTS_READY : 0u);
    d.slot = ss.k_slot;
    d.key_bytes = nullptr;
    d.key_off = nullptr;
    d.max_burst = d.count_per_period = d.period = d.quantity = d.now_ns = nullptr;
    HostIn hin;
    hin.col[0] = b.max_burst, hin.col[1] = b.count_per_period, hin.col[2] = b.period, hin.col[3] = b.quantity, hin.col[4] = b.now_ns;
    if (piped) e->wait_before_sort = e->k_done;
    TC_TRY(stage_outputs(e, b, d));
    TC_TRY(run_slots_device(e, d, &hin));
    return finish_async(e, b);
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
    TC_HIP(e, hipSetDevice(e->device));
    if (b.flags & TC_B_ASYNC) {
        if (b.flags & TC_B_DEVICE_PTRS) return fail(e, TC_E_INVALID_ARG, "TC_B_ASYNC is for host-pointer batches (device-pointer batches are asynchronous anyway)");
        return run_keys_host_async(e, b);
    }
    const uint8_t* d_bytes = b.key_bytes;
    const uint32_t* d_off = b.key_off;
    const bool dev = (b.flags & TC_B_DEVICE_PTRS) != 0;
    if (!dev && small_batch_applies(e, b)) return run_small_batch(e, b);
    if (!dev) {
        int rc = stage_keys(e, b.key_bytes, b.key_off, b.n, &d_bytes, &d_off);
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
        return run_slots_device(e, s);
    }
    int rc = resolve_keys_device(e, d_bytes, d_off, (uint32_t)b.n, true, e->k_slot, false);
    if (rc != TC_E_OK) return rc;
    // host pointers for everything else: reuse the slot path's staging, with the
    // slot column already on the device
    TC_TRY(stage_need(e, e->stage.slot, e->max_batch));
    TC_HIP(e, hipMemcpyAsync(e->stage.slot, e->k_slot, b.n * sizeof(uint32_t), hipMemcpyDeviceToDevice, cur_stream(e)));
    rc = run_slots_host_staged(e, b);
    if (rc != TC_E_OK) return rc;
    return check_key_errors(e);
}

extern "C" int tc_rate_limit(tc_engine* e, const uint8_t* key, size_t key_len, int64_t max_burst,
                             int64_t count_per_period, int64_t period, int64_t quantity, int64_t now_ns,
                             tc_result* out) {
    if (!e || !out || (!key && key_len)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (key_len > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "key too long");
    if (e->fixed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: single calls carry their own rate; use a registered batch");
    TC_HIP(e, hipSetDevice(e->device));
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
    OneResult r;
    TC_HIP(e, hipMemcpyAsync(&r, e->one_result, sizeof r, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    e->batches++;
    if (r.table_full) {
        uint32_t zero = 0;
        TC_HIP(e, hipMemcpyAsync(e->kt.error_flag, &zero, sizeof zero, hipMemcpyHostToDevice, s));
        TC_HIP(e, hipStreamSynchronize(s));
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

extern "C" int tc_sweep_expired(tc_engine* e, int64_t now_ns, uint64_t* removed) {
    if (!e) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    TC_HIP(e, hipSetDevice(e->device));
    hipStream_t s = cur_stream(e);
    unsigned long long* scratch = e->counters + TC_CNT_COUNT;
    if (e->k_busy) { // key stages still in flight on the key stream come first
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    TC_HIP(e, hipMemsetAsync(scratch, 0, sizeof(unsigned long long), s));
    TC_HIP(e, hipMemsetAsync(e->counters + TC_CNT_LIVE_SLOTS, 0, sizeof(unsigned long long), s));
    if (e->key_mode)
        hipLaunchKernelGGL(k_sweep_keys, dim3(std::min<uint64_t>(nblocks(e->capacity), 2048)), dim3(BLOCK), 0, s,
                           e->cells, e->kt, now_ns, e->counters, scratch, e->denied);
    else if (e->fixed)
        hipLaunchKernelGGL(k_sweep_fixed, dim3(std::min<uint64_t>(nblocks(e->capacity), 2048)), dim3(BLOCK), 0, s, e->tat8, e->rate_id,
                           e->classes, (uint32_t)e->uniform_id, e->capacity, now_ns, e->counters, scratch);
    else
        hipLaunchKernelGGL(k_sweep, dim3(std::min<uint64_t>(nblocks(e->capacity), 2048)), dim3(BLOCK), 0, s,
                           e->cells, e->capacity, now_ns, e->counters, scratch);
    TC_HIP(e, hipGetLastError());
    if (e->key_mode) {
        // tombstones lengthen probe chains: rebuild the table once they fill 1/4 of it
        int rc = rebuild_key_table_if_due(e);
        if (rc != TC_E_OK) return rc;
        // the sweep (and a rebuild) changed the key table: later key stages on the key stream wait for it
        TC_HIP(e, hipEventRecord(e->m_done, s));
        e->m_busy = true;
    }
    if (!removed) return TC_E_OK; // asynchronous: the count goes to TC_CNT_SWEPT
    unsigned long long r = 0;
    TC_HIP(e, hipMemcpyAsync(&r, scratch, sizeof r, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    *removed = r;
    return TC_E_OK;
}

extern "C" int tc_counters_refresh(tc_engine* e) {
    if (!e) return TC_E_INVALID_ARG;
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
    TC_HIP(e, hipMemcpyAsync(out, e->counters, TC_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    out[TC_CNT_BATCHES] = e->batches;
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

static int store_op(tc_engine* e, uint64_t slot, int op, int64_t a, int64_t b, uint64_t ttl, int64_t now,
                    StoreOpResult* r) {
    if (e->fixed) return fail(e, TC_E_UNSUPPORTED, "TC_CFG_FIXED_PARAMS: the 8-byte layout cannot hold a free ttl (Store operations need the 16-byte cell)");
    if (now < 0) return fail(e, TC_E_INVALID_ARG, "now_ns < 0");
    TC_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_store_op, dim3(1), dim3(64), 0, cur_stream(e), e->cells, slot, op, a, b, ttl, now, e->op_result);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipMemcpyAsync(r, e->op_result, sizeof *r, hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
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
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, false, &slot);
    if (rc != TC_E_OK) return rc;
    if (slot == kt::NO_SLOT) { // adaptive_cleanup.rs:242: None => Ok(false)
        *swapped = 0;
        return TC_E_OK;
    }
    StoreOpResult r;
    rc = store_op(e, slot, 1, old_value, new_value, ttl_ns, now_ns, &r);
    if (rc != TC_E_OK) return rc;
    *swapped = r.flag;
    return TC_E_OK;
}

extern "C" int tc_store_set_if_not_exists_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t value,
                                                   uint64_t ttl_ns, int64_t now_ns, int* was_set) {
    if (!e || !was_set) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, true, &slot);
    if (rc != TC_E_OK) return rc;
    StoreOpResult r;
    rc = store_op(e, slot, 2, value, 0, ttl_ns, now_ns, &r);
    if (rc != TC_E_OK) return rc;
    *was_set = r.flag;
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
    if (!e->denied) return fail(e, TC_E_UNSUPPORTED, "engine was created without TC_CFG_TRACK_DENIED");
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipMemsetAsync(e->denied, 0, e->capacity * sizeof(uint32_t), cur_stream(e)));
    if (e->retired) TC_HIP(e, hipMemsetAsync(e->retired, 0, (size_t)kt::RETIRED_CAP * sizeof(kt::RetiredRec), cur_stream(e)));
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
    // keys that hold a slot
    // (keys over 256 bytes are not tracked by the reference, metrics.rs:36-39: they are filtered out below, so a few
    // more candidates than k are fetched)
    const uint32_t k_live = std::min<uint32_t>(k + 64u, TOPK_MAX);
    std::vector<uint32_t> slots(k_live);
    std::vector<uint64_t> cnt(k_live);
    uint32_t n_live = 0;
    TC_TRY(tc_top_denied(e, k_live, slots.data(), cnt.data(), &n_live));
    std::vector<std::pair<std::string, uint64_t>> all;
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
    // keys that lost theirs (a key's denials are in exactly one place: see kt::RetiredRec)
    std::vector<kt::RetiredRec> rt(kt::RETIRED_CAP);
    hipStream_t s = cur_stream(e);
    TC_HIP(e, hipMemcpyAsync(rt.data(), e->retired, rt.size() * sizeof(kt::RetiredRec), hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    std::vector<std::pair<std::string, uint64_t>> retired;
    for (const kt::RetiredRec& r : rt)
        if ((r.tag & kt::RT_VALID) && r.count) retired.emplace_back(std::string((const char*)r.bytes, r.len), (uint64_t)r.count);
    auto by_count = [](const std::pair<std::string, uint64_t>& a, const std::pair<std::string, uint64_t>& b) {
        return a.second != b.second ? a.second > b.second : a.first < b.first;
    };
    if (retired.size() > 3u * TOPK_MAX) {
        // TopDeniedKeys::cleanup (metrics.rs:52-64): past 3 x the limit only the most denied `limit` keys are kept.
        // The table is rewritten from the host (the engine is drained: this is a synchronous call).
        std::sort(retired.begin(), retired.end(), by_count);
        retired.resize(TOPK_MAX);
        std::fill(rt.begin(), rt.end(), kt::RetiredRec{});
        for (const auto& kv : retired) {
            const uint64_t h = kt::hash_key((const uint8_t*)kv.first.data(), (uint32_t)kv.first.size());
            uint32_t pos = (uint32_t)(h >> 17) & (kt::RETIRED_CAP - 1u);
            while (rt[pos].tag != kt::RT_EMPTY) pos = (pos + 1u) & (kt::RETIRED_CAP - 1u);
            rt[pos].tag = h | kt::RT_VALID;
            rt[pos].count = (uint32_t)kv.second;
            rt[pos].len = (uint32_t)kv.first.size();
            memcpy(rt[pos].bytes, kv.first.data(), kv.first.size());
        }
        TC_HIP(e, hipMemcpyAsync(e->retired, rt.data(), rt.size() * sizeof(kt::RetiredRec), hipMemcpyHostToDevice, s));
        TC_HIP(e, hipStreamSynchronize(s));
    }
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

extern "C" int tc_slot_keys(tc_engine* e, uint32_t n, const uint32_t* slots, uint8_t* key_bytes, size_t key_bytes_cap,
                            uint32_t* key_off) {
    if (!e || (n && (!slots || !key_off)) || (key_bytes_cap && !key_bytes)) return TC_E_INVALID_ARG;
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

// ---- snapshot / restore (the reference keeps its state in memory only and loses it on restart;
//      columnar state makes a checkpoint a handful of device-to-host copies) -------------------
namespace {
struct SnapHeader {
    char magic[8]; // "TCGPUSN1"
    uint32_t version, key_mode;
    uint64_t capacity, nb, overflow_bytes, n_classes;
    uint64_t batches;
    uint64_t counters[TC_CNT_COUNT];
    uint64_t payload_bytes; // everything after the header
    uint64_t checksum;      // FNV-1a 64 over the payload: a truncated, corrupt or foreign file is refused BEFORE the engine is touched
};
constexpr uint32_t SNAP_VERSION = 3; // 3: sharded tombstone count
// FNV-1a over 8-byte words of the byte stream (a byte-wise FNV over gigabytes of state would take seconds);
// independent of how the stream is cut into pieces
struct StreamSum {
    uint64_t h = 0xcbf29ce484222325ull;
    uint8_t tail[8];
    size_t ntail = 0;
    void add(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        while (n && ntail) { // finish the word left over from the previous piece
            tail[ntail++] = *p++;
            --n;
            if (ntail == 8) {
                uint64_t w;
                memcpy(&w, tail, 8);
                h = (h ^ w) * 0x100000001b3ull;
                ntail = 0;
            }
        }
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, p + i, 8);
            h = (h ^ w) * 0x100000001b3ull;
        }
        for (; i < n; ++i) tail[ntail++] = p[i];
    }
    uint64_t done() const {
        uint64_t r = h;
        for (size_t i = 0; i < ntail; ++i) r = (r ^ tail[i]) * 0x100000001b3ull;
        return r;
    }
};
struct Section {
    void* dev;
    size_t bytes;
};
std::vector<Section> snapshot_sections(tc_engine* e) {
    std::vector<Section> v;
    if (e->fixed) v.push_back({e->tat8, e->capacity * sizeof(int64_t)});
    else v.push_back({e->cells, e->capacity * sizeof(Cell)});
    v.push_back({e->rate_id, e->capacity * sizeof(uint16_t)});
    if (e->denied) v.push_back({e->denied, e->capacity * sizeof(uint32_t)});
    if (e->key_mode) {
        const kt::Table& t = e->kt;
        v.push_back({t.ktab, (t.nb_mask + 1) * sizeof(kt::Entry)});
        v.push_back({t.rec, (size_t)t.capacity * sizeof(kt::KeyRec)});
        v.push_back({t.bound, (size_t)t.capacity});
        v.push_back({t.overflow, (size_t)t.overflow_bytes * 2});
        v.push_back({t.free_slots, (size_t)t.capacity * 4});
        v.push_back({t.overflow_used, 64}); // overflow_used | free_top | error_flag | overflow_half
        v.push_back({t.tombs, kt::TOMB_SHARDS * 4});
        if (e->retired) v.push_back({e->retired, (size_t)kt::RETIRED_CAP * sizeof(kt::RetiredRec)});
    }
    return v;
}
int drain_for_snapshot(tc_engine* e) {
    TC_HIP(e, hipSetDevice(e->device));
    hipStream_t s = cur_stream(e);
    if (e->k_busy) {
        TC_HIP(e, hipStreamWaitEvent(s, e->k_done, 0));
        e->k_busy = false;
    }
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(NSHARD), 0, s, e->counters);
    TC_HIP(e, hipStreamSynchronize(s));
    return TC_E_OK;
}
} // namespace

extern "C" int tc_snapshot_save(tc_engine* e, const char* path) {
    if (!e || !path) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    int rc = drain_for_snapshot(e);
    if (rc != TC_E_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) return fail(e, TC_E_INVALID_ARG, "tc_snapshot_save: cannot open file");
    SnapHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "TCGPUSN1", 8);
    h.version = SNAP_VERSION;
    h.key_mode = e->key_mode ? 1u : 0u;
    h.capacity = e->capacity;
    h.nb = e->key_mode ? e->kt.nb_mask + 1 : 0;
    h.overflow_bytes = e->key_mode ? e->kt.overflow_bytes : 0;
    h.n_classes = e->host_classes.size();
    h.batches = e->batches;
    h.key_mode |= e->denied ? 2u : 0u;
    h.key_mode |= e->fixed ? 4u : 0u;
    if (hipMemcpy(h.counters, e->counters, sizeof h.counters, hipMemcpyDeviceToHost) != hipSuccess) {
        fclose(f);
        return fail(e, TC_E_HIP, "tc_snapshot_save: counter copy failed");
    }
    bool ok = fwrite(&h, sizeof h, 1, f) == 1; // (rewritten with the payload's size and checksum at the end)
    StreamSum sum;
    uint64_t bytes = 0;
    auto put = [&](const void* p, size_t n) {
        if (!ok || n == 0) return;
        ok = fwrite(p, 1, n, f) == n;
        sum.add(p, n);
        bytes += n;
    };
    // rate plans: the (burst,count,period) keys in id order (the dictionary is rebuilt from them)
    std::vector<int64_t> plans(3 * e->host_classes.size(), 0);
    for (const auto& kv : e->class_of) memcpy(&plans[3 * kv.second], kv.first.data(), 24);
    put(plans.data(), plans.size() * 8);
    put(&e->uniform_id, sizeof e->uniform_id);
    const size_t CHUNK = 64u << 20;
    std::vector<uint8_t> buf(CHUNK);
    for (const Section& sec : snapshot_sections(e)) {
        for (size_t off = 0; ok && off < sec.bytes; off += CHUNK) {
            const size_t nbytes = std::min(CHUNK, sec.bytes - off);
            if (hipMemcpy(buf.data(), (const uint8_t*)sec.dev + off, nbytes, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
            put(buf.data(), nbytes);
        }
    }
    h.payload_bytes = bytes;
    h.checksum = sum.done();
    ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof h, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? TC_E_OK : fail(e, TC_E_INVALID_ARG, "tc_snapshot_save: write failed");
}

extern "C" int tc_snapshot_load(tc_engine* e, const char* path) {
    if (!e || !path) return TC_E_INVALID_ARG;
    int rc = drain_for_snapshot(e);
    if (rc != TC_E_OK) return rc;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(e, TC_E_INVALID_ARG, "tc_snapshot_load: cannot open file");
    SnapHeader h;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, "TCGPUSN1", 8) == 0;
    if (ok && h.version != SNAP_VERSION) {
        // the on-disk layout follows the resident layout and changes with it; there is no migration (README.md,
        // "Snapshots"): a snapshot is a restart aid for ONE build, the state itself expires within minutes anyway
        fclose(f);
        e->err = "tc_snapshot_load: snapshot format version " + std::to_string(h.version) + ", this build reads version " +
                 std::to_string(SNAP_VERSION) + " only (no migration: take a new snapshot with this build)";
        return TC_E_UNSUPPORTED;
    }
    const uint32_t mode = (e->key_mode ? 1u : 0u) | (e->denied ? 2u : 0u) | (e->fixed ? 4u : 0u);
    if (!ok || h.key_mode != mode || h.capacity != e->capacity || (e->key_mode && (h.nb != e->kt.nb_mask + 1 ||
                                                                                  h.overflow_bytes != e->kt.overflow_bytes)) ||
        h.n_classes == 0 || h.n_classes > MAX_CLASSES) {
        fclose(f);
        return fail(e, TC_E_INVALID_ARG, "tc_snapshot_load: not a snapshot of an engine with this configuration");
    }
    {   // first pass over the file: size and checksum -- nothing of the engine is touched before they hold
        StreamSum sum;
        uint64_t bytes = 0;
        std::vector<uint8_t> tmp(16u << 20);
        size_t got;
        while ((got = fread(tmp.data(), 1, tmp.size(), f)) > 0) {
            sum.add(tmp.data(), got);
            bytes += got;
        }
        if (bytes != h.payload_bytes || sum.done() != h.checksum || fseek(f, (long)sizeof h, SEEK_SET) != 0) {
            fclose(f);
            return fail(e, TC_E_INVALID_ARG, "tc_snapshot_load: truncated or corrupt snapshot (the engine was not touched)");
        }
    }
    std::vector<int64_t> plans(3 * h.n_classes);
    ok = fread(plans.data(), 8, plans.size(), f) == plans.size();
    uint16_t uniform_id = 0;
    ok = ok && fread(&uniform_id, sizeof uniform_id, 1, f) == 1;
    if (ok) { // rebuild the plan dictionary in id order
        e->host_classes.assign(1, RateClass{0, 0, 0, 0});
        e->class_of.clear();
        e->cls_min_ei = e->cls_min_dvt = INT64_MAX;
        e->cls_max_ei = e->cls_max_dvt = 0;
        for (uint64_t id = 1; id < h.n_classes && ok; ++id) {
            bool grew = false;
            ok = intern_class(e, plans[3 * id], plans[3 * id + 1], plans[3 * id + 2], &grew) == (int)id;
        }
        e->uniform_id = uniform_id;
        ok = ok && hipMemset(e->classes, 0, (size_t)MAX_CLASSES * sizeof(RateClass)) == hipSuccess &&
             hipMemcpy(e->classes, e->host_classes.data(), e->host_classes.size() * sizeof(RateClass), hipMemcpyHostToDevice) == hipSuccess;
    }
    const size_t CHUNK = 64u << 20;
    std::vector<uint8_t> buf(CHUNK);
    for (const Section& sec : snapshot_sections(e)) {
        for (size_t off = 0; ok && off < sec.bytes; off += CHUNK) {
            const size_t nbytes = std::min(CHUNK, sec.bytes - off);
            ok = fread(buf.data(), 1, nbytes, f) == nbytes &&
                 hipMemcpy((uint8_t*)sec.dev + off, buf.data(), nbytes, hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    fclose(f);
    if (!ok) return fail(e, TC_E_INVALID_ARG, "tc_snapshot_load: truncated or unreadable snapshot (engine state is now undefined)");
    // counters: canonical block back, shards cleared (the folded totals live in shard 0)
    const size_t cnt_words = (TC_CNT_COUNT + 1) + (size_t)NSHARD * SHARD_WORDS;
    std::vector<unsigned long long> c(cnt_words, 0ull);
    for (int i = 0; i < TC_CNT_COUNT; ++i) c[i] = h.counters[i];
    c[(TC_CNT_COUNT + 1) + 0] = h.counters[TC_CNT_ALLOWED];
    c[(TC_CNT_COUNT + 1) + 1] = h.counters[TC_CNT_DENIED];
    c[(TC_CNT_COUNT + 1) + 2] = h.counters[TC_CNT_ERRORS];
    TC_HIP(e, hipMemcpy(e->counters, c.data(), cnt_words * sizeof(unsigned long long), hipMemcpyHostToDevice));
    TC_TRY(publish_poison_ptr(e));
    e->batches = h.batches;
    e->sealed = true; // (fixed layout: the loaded state was written under the loaded plans)
    return TC_E_OK;
}

// ---- routing of a global stream (route_kernels.hpp) --------------------------------------------------
extern "C" int tc_route_batch(tc_engine* e, const tc_route* rp) {
    if (!e || !rp || rp->struct_size < offsetof(tc_route, stream)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    tc_route r; // (callers built against the struct without `stream` get the engine's stream)
    memset(&r, 0, sizeof(r));
    memcpy(&r, rp, std::min<size_t>(rp->struct_size, sizeof(r)));
    rt::Map m;
    if (!rt::make_map(r.world, r.keys_per_shard, &m)) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: world must be 1..64 and keys_per_shard 1..2^32");
    if (r.only >= (int32_t)r.world || r.only < -1) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: `only` is not a destination");
    if (r.out_dst && r.only != -1) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: out_dst goes with only = -1");
    if (!r.global_id || (!r.out_slot && !r.out_dst) || !r.out_count) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: NULL array");
    rt::SplitOut split;
    memset(&split, 0, sizeof split);
    if (r.out_dst)
        for (uint32_t d = 0; d < r.world; ++d) {
            if (!r.out_dst[d]) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: NULL destination in out_dst");
            split.ptr[d] = r.out_dst[d];
        }
    if (r.n == 0 || r.n > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "tc_route_batch: n out of range");
    TC_HIP(e, hipSetDevice(e->device));
    hipStream_t s = r.stream ? (hipStream_t)r.stream : cur_stream(e);
    uint32_t lane = 0; // which scratch
    if ((r.flags & TC_ROUTE_AHEAD) && !r.stream) {
        TC_TRY(ensure_side_streams(e));
        if (e->n_aux) {
            // Beside the evaluations: on a grouping stream, behind the GROUPING of every batch enqueued so far -- a
            // slot column is read by the grouping kernels only, so that is the last reader of whatever buffer this
            // call overwrites; waiting for the evaluations too would chain the next batches' grouping, queued behind
            // the router on this stream, to them.  Every grouping stream has its own router scratch.
            lane = 1 + e->next_route++ % e->n_aux;
            s = e->aux[lane - 1];
            // Behind the batches that read the very buffer being overwritten, if the engine still knows them (every set
            // remembers the slot columns of its last four batches; waiting for a set's current grouping covers its
            // earlier batches too, see SortSet::readers), else behind every batch in flight.  TC_ROUTE_NO_READERS: the
            // caller vouches that out_slot is no batch's slot column (it copies the segments elsewhere: the exchange).
            if (!(r.flags & TC_ROUTE_NO_READERS)) {
                bool known = false;
                for (int pass = 0; pass < 2 && !known; ++pass)
                    for (uint32_t si = 0; si < e->depth; ++si) {
                        tc_engine::SortSet& ss = e->sets[si];
                        if (!ss.in_use) continue;
                        bool reads = false;
                        for (const auto& rd : ss.readers)
                            reads = reads || (rd.ptr && r.out_slot && rd.ptr < r.out_slot + r.n && r.out_slot < rd.ptr + rd.n);
                        if (pass == 0 && !reads) continue;
                        if (pass == 0) known = true;
                        TC_HIP(e, hipStreamWaitEvent(s, ss.grouped_aside ? ss.sorted : ss.consumed, 0));
                    }
            }
        }
    }
    const uint32_t n = (uint32_t)r.n, tiles = (n + rt::TILE - 1) / rt::TILE;
    const size_t words = (size_t)tiles * r.world + 4 + 2 * (size_t)rt::ONE_PASS_TILES;
    if (words > e->route_ws_words) {
        if (e->route_ws) {
            TC_HIP(e, hipDeviceSynchronize()); // (whatever streams earlier routers ran on)
            (void)hipFree(e->route_ws);
            e->route_ws = nullptr;
            e->route_ws_words = 0;
        }
        // one scratch per stream a router may run on (the caller's / the engine's, each grouping stream): routers on
        // different streams never wait for one another
        const size_t each = (words * 2 + 1) & ~(size_t)1;
        TC_HIP(e, hipMalloc(&e->route_ws, each * (1 + AUX_MAX) * sizeof(uint32_t)));
        TC_HIP(e, hipMemset(e->route_ws, 0, each * (1 + AUX_MAX) * sizeof(uint32_t)));
        e->route_ws_words = each;
    }
    rt::Work w;
    uint32_t* scratch = e->route_ws + (size_t)lane * e->route_ws_words;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(scratch); // [ONE_PASS_TILES] look-back words of the one-pass router
    w.tile_cnt = scratch + 2 * (size_t)rt::ONE_PASS_TILES;
    w.totals = r.out_count;
    w.tiles = tiles;
    w.host_totals = r.out_count_host;
    w.tag = r.tag;
    // one destination: count, offset and scatter in ONE pass over the ids for grids of at most ONE_PASS_TILES (tiles of 4096
    // or 8192 ids); every destination, or a bigger batch: count | scan | scatter
    const uint32_t tiles32 = (n + 2 * rt::TILE - 1) / (2 * rt::TILE);
    if (r.only >= 0 && tiles32 <= rt::ONE_PASS_TILES && !getenv("TCGPU_ROUTE_3PASS")) {
        if (++e->route_seq == 0u) e->route_seq = 1u;
        unsigned long long* viol = e->counters + (TC_CNT_COUNT + 1) + 3;
        if (tiles <= rt::ONE_PASS_TILES) {
            hipLaunchKernelGGL((rt::k_route_one<rt::ITEMS>), dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (uint32_t)r.only, status,
                               e->route_seq, r.out_slot, r.out_pos, viol);
        } else {
            w.tiles = tiles32;
            hipLaunchKernelGGL((rt::k_route_one<2 * rt::ITEMS>), dim3(tiles32), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (uint32_t)r.only,
                               status, e->route_seq, r.out_slot, r.out_pos, viol);
        }
        hipLaunchKernelGGL(rt::k_route_scan, dim3(r.world), dim3(rt::THREADS), 0, s, w); // (the totals)
    } else {
        hipLaunchKernelGGL(rt::k_route_count, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w);
        hipLaunchKernelGGL(rt::k_route_scan, dim3(r.world), dim3(rt::THREADS), 0, s, w);
        if (r.out_dst) hipLaunchKernelGGL(rt::k_route_scatter<true>, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (int)r.only, r.out_slot, r.out_pos, split);
        else hipLaunchKernelGGL(rt::k_route_scatter<false>, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (int)r.only, r.out_slot, r.out_pos, split);
    }
    if (r.out_count_host) hipLaunchKernelGGL(rt::k_route_publish, dim3(1), dim3(rt::MAX_WORLD), 0, s, w, r.world);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

extern "C" int tc_forward_segments(tc_engine* e, const tc_forward* f) {
    if (!e || !f || f->struct_size < sizeof(tc_forward)) return TC_E_INVALID_ARG;
    TC_CHECK_POISON(e);
    if (f->world == 0 || f->world > 64 || !f->src || !f->count || !f->dst) return fail(e, TC_E_INVALID_ARG, "tc_forward_segments: world 1..64, arrays given");
    TC_HIP(e, hipSetDevice(e->device));
    mk::Destinations ds;
    memset(&ds, 0, sizeof ds);
    ds.n = f->world;
    uint64_t at = 0;
    for (uint32_t d = 0; d < f->world; ++d) {
        if (f->count[d] && !f->dst[d]) return fail(e, TC_E_INVALID_ARG, "tc_forward_segments: NULL destination");
        ds.ptr[d] = f->dst[d];
        ds.start[d] = (uint32_t)at;
        at += f->count[d];
    }
    if (at > 0x7FFFFFFFull) return fail(e, TC_E_INVALID_ARG, "tc_forward_segments: too many requests");
    ds.start[f->world] = (uint32_t)at;
    if (at == 0) return TC_E_OK;
    hipStream_t s = f->stream ? (hipStream_t)f->stream : cur_stream(e);
    hipLaunchKernelGGL(mk::k_forward, dim3(std::min<uint32_t>(nblocks(at), 512u)), dim3(BLOCK), 0, s, f->src, ds);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}

extern "C" int tc_route_host(uint32_t world, uint64_t keys_per_shard, uint64_t n, const uint32_t* global_id, uint32_t* owner,
                             uint32_t* slot) {
    rt::Map m;
    if (!rt::make_map(world, keys_per_shard, &m) || (n && !global_id)) return TC_E_INVALID_ARG;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t o, sl;
        rt::route_of_host(m, global_id[i], o, sl);
        if (owner) owner[i] = o;
        if (slot) slot[i] = sl;
    }
    return TC_E_OK;
}
extern "C" int tc_route_inverse(uint32_t world, uint64_t keys_per_shard, uint64_t n, const uint32_t* owner, const uint32_t* slot,
                                uint64_t* global_id) {
    rt::Map m;
    if (!rt::make_map(world, keys_per_shard, &m) || (n && (!owner || !slot || !global_id))) return TC_E_INVALID_ARG;
    for (uint64_t i = 0; i < n; ++i) {
        if (owner[i] >= world || slot[i] >= keys_per_shard) return TC_E_INVALID_ARG;
        global_id[i] = rt::route_inverse_host(m, owner[i], slot[i]);
    }
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
extern "C" int tc_selfcheck(tc_engine* e, uint64_t* violations) {
    if (!e || !violations) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    unsigned long long v = 0;
    TC_HIP(e, hipMemcpyAsync(&v, e->counters + (TC_CNT_COUNT + 1) + 3, sizeof v, hipMemcpyDeviceToHost, cur_stream(e)));
    TC_HIP(e, hipStreamSynchronize(cur_stream(e)));
    *violations = v;
    return TC_E_OK;
}
