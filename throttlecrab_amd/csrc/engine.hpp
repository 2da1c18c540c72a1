// engine.hpp -- what the translation units of libtcgpu.so share: the engine's state (struct tc_engine), the error /
// launch macros and the handful of host helpers that more than one unit calls.  The kernels live in the *_kernels.hpp /
// radix_sort.hpp / bucket_path.hpp / key_table.hpp headers (internal linkage: every unit compiles the ones it launches).
//   engine.hip    create / destroy, allocation, streams, rate plans, counters, profiling, hooks
//   slots.hip     tc_rate_limit_batch_slots: staging, grouping (sort / bucket path), evaluation, small batches
//   keys.hip      string keys: key stages, tc_rate_limit_batch_keys, tc_rate_limit, the `trait Store` shims
//   maint.hip     sweep, top denied keys, state introspection
//   snapshot.hip  tc_snapshot_save / load
//   route.hip     multi-GPU: tc_route_batch, tc_forward_segments, the host mirror of the map
//   autosweep.hip self-cleaning: tc_set_sweep_policy, the policy feed, auto_sweep_before / auto_sweep_after
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/tcgpu.h"
#include "../../include/throttlecrab_sweep.hpp"
#include "gcra_math.hpp"
#include "key_table.hpp"
#include "radix_sort.hpp"
#include "range_part.hpp"

#include "eval_kernels.hpp"
#include "bucket_path.hpp"
#include "maintenance_kernels.hpp"
#include "route_kernels.hpp"

using tc::Cell;
using tc::RateClass;

using namespace ev;
using namespace mk;

namespace tcg {


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

} // namespace tcg
using namespace tcg;

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
    bool own_pooled = false;             // own_stream belongs to a set of the process's stream pool (engine.hip): never destroyed, handed on
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
        uint64_t* elem_c = nullptr;                    // range path: scratch of a range that does not fit LDS (min(max_batch, range_max_n))
        uint32_t* range_totals = nullptr;              // range path: 2 x rp::NB_HOT words, the buckets' sizes (ranges, then hot ids) of this / the next batch of the set
        uint32_t range_parity = 0;
        uint32_t* part_table = nullptr;                // round 6: rp::k_tile_part's table, rp::NB_HOT words per tile of rp::PT_TILE requests
        rp::HotDev* hot_dev = nullptr;                 // ... and the set's hot table, as of list version hot_version (0: none installed)
        uint32_t hot_version = 0;
        unsigned long long publish_due = 0;            // a copy of the evaluations' notes to be enqueued when the set is next grouped aside (its sequence word; 0: none)
        uint32_t* hot_info = nullptr;                  // rank form: hot id << 16 | rank inside its tile per request (max_batch words)
        uint32_t* hot_P = nullptr;                     // ... requests of a hot id in the tiles before (tiles x rp::HOT_MAX words)
        uint32_t* hot_n = nullptr;                     // ... requests per hot id in the batch; [rp::HOT_MAX]: requests the ranges hold
        ev::HotEval* hot_eval = nullptr;               // ... the addresses above as the lean evaluation's hot role reads them (device memory, written once)
        uint32_t* ws = nullptr;                        // hist x2 | ticket | look-back status
        uint32_t* k_slot = nullptr;                    // key mode: slots resolved for the batch using this set
        uint32_t* h_slot = nullptr;                    // TC_B_ASYNC: the host batch's slot column, staged (lazy)
        int64_t* h_in[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // ... and its request columns (lazy)
        int64_t* h_dict = nullptr;                     // TC_B_PLAN_DICT: the batch's dictionary, plan ids and u32 quantities as they arrived (lazy)
        uint16_t* h_plan_id = nullptr;
        uint32_t* h_q32 = nullptr;
        uint8_t* h_key_bytes = nullptr;                // TC_B_ASYNC key batch: its key arena and offsets, staged (lazy)
        size_t h_key_cap = 0;
        uint32_t* h_key_off = nullptr;
        uint32_t hist_parity = 0;
        void* bp_scratch = nullptr;    // bucket path: tile histograms, offsets, partitioned elements, gate
        bp::Work bpw;
        hipEvent_t sorted = nullptr;   // recorded on the auxiliary stream after the last pass
        hipEvent_t consumed = nullptr; // recorded on `stream` after the evaluation that read this set
        bool in_use = false;
        uint32_t k_n = 0;              // string mode: requests of the pipelined key batch whose slots are in k_slot (the sweep beside evaluations)
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
    void* probe_stamps = nullptr;        // {start of k_probe_occupy, k_probe_stamp's clock}
    bool pipe_probe = true;              // TCGPU_PIPE_PROBE=0: keep candidates that share a dispatch pipe with the main stream
    // what the last probe found (tc_engine_info_get)
    uint32_t probe_tried = 0, probe_same_queue = 0, probe_same_pipe = 0, probe_second_best = 0;
    bool probe_assumed = false;
    bool probe_pooled = false;           // the streams came from the process's pool: the verdict of an earlier probe, re-checked (tc_engine_info: probes_pooled)
    bool side_pooled = false;            // ... and go back to it when the engine lets go of them
    uint32_t last_grouping_path = 0;     // 1 range path, 2 LSD passes, 3 bucket path, 4 no grouping (small batch / unique slots), 5 range path with hot slots peeled
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
    // range path (radix_sort.hpp): one partition pass by key range + an in-LDS finish instead of three LSD passes
    int range_mode = 2;                  // TCGPU_RANGE: 0 off, 1 pipelined batches only, 2 every batch
    bool range_ok = false;               // the key space fits (more than 65536 keys, widest range <= 65536 slots)
    uint32_t range_mul = 0;
    bool range_ilv = false;     // string mode (round 6, late): the ranges interleaved chunk by chunk (radix_sort.hpp: RANGE_ILV; TCGPU_RANGE_ILV=0: contiguous)
    int range_sub_passes = 0;            // 8-bit digits of a slot's offset inside its range
    uint32_t range_max_n = 0;            // batches above this are sorted (TCGPU_RANGE_MAX_N)
    unsigned long long* range_hint_host = nullptr; // pinned: n << 32 | largest range of a recent batch, written by the first pass
    unsigned long long* range_hint_dev = nullptr;
    static constexpr uint32_t RANGE_HINTS = 8;       // looks at the hint the path goes by (the worst of them)
    uint32_t range_share[RANGE_HINTS] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t range_looks = 0;
    uint32_t range_relook = 0; // in-order batches the range path turned down (every 32nd goes through the sort path, which writes a hint)
    // round 6: hot slots peeled out of the range partition (range_part.hpp), so that skewed streams take the range path too
    struct Hot {
        bool on = true;                  // TCGPU_HOT=0: skewed streams stay on the LSD passes
        unsigned long long* notes_dev = nullptr;      // ev::HEAVY_SLOTS notes of the evaluations (ev::heavy_note)
        unsigned long long* notes_host = nullptr;     // pinned: a copy of them + the copy's sequence word (mk::k_heavy_publish)
        unsigned long long* notes_host_dev = nullptr; // (the same block as the device addresses it)
        uint64_t seq_seen = 0;           // the copy the list below was made from
        uint64_t published = 0;          // copies enqueued so far
        uint64_t evals = 0;              // evaluations that took notes
        uint32_t heavy_min = 32;         // a run of at least this many requests is noted (TCGPU_HOT_MIN)
        std::vector<uint32_t> slots;     // the hot list, heaviest first (<= rp::HOT_MAX)
        uint32_t version = 0;            // bumped whenever the list changes (the sets install it when they next need it)
        uint32_t backoff = 0;            // batches left during which the hot form is not tried (peeling did not make the ranges fit)
        unsigned long long* hint_cold_host = nullptr; // pinned: n << 32 | largest range of a recent batch grouped in the hot form (without its hot slots)
        unsigned long long* hint_cold_dev = nullptr;
        uint64_t batches_hot = 0;        // batches grouped in the hot form so far
        std::atomic<uint32_t> stable_looks{0}; // looks in a row that kept the list
        bool rank_on = true;             // TCGPU_HOT_RANK=0: lean batches also take the gather form (A/B)
        uint32_t* done = nullptr;        // rank form: the evaluation's count of finished hot-role blocks (zero between launches); word 8: k_heavy_publish's
        const ev::HotEval* he_dev = nullptr; // ... what the next lean evaluation is handed (nullptr: no hot role)
        std::vector<unsigned long long> scratch, found; // hot_make's workspace
        // The list is MADE on a thread of the engine's own (slots.hip: hot_worker): reading 8 192 notes, deduplicating and ranking
        // them took the caller's thread 60-90 us whenever a new copy had arrived -- seven times in an engine's first 24 batches,
        // which is where the driver's 20-batch region lies (Zipf 47 us per batch there against 38 in later regions,
        // tools/host_calls.py).  The caller's thread only TAKES a finished list (a swap under the mutex).
        bool threaded = true;            // TCGPU_HOT_THREAD=0: made on the caller's thread, as before
        std::vector<uint32_t> made;      // the list as last made (the maker's side; == slots once taken)
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        bool worker_on = false, stop = false;     // (stop: under mu)
        std::vector<uint32_t> next;      // a finished list the caller has not taken yet (under mu)
        std::atomic<bool> next_ready{false};
        std::atomic<bool> asleep{false}; // the worker waits on cv: no call for a while (the next call wakes it)
        std::atomic<int64_t> last_call_ns{0};
        uint64_t lists_made = 0;         // (worker's)
    } hot;
    uint32_t* fill_hint_host = nullptr; // pinned: "most decisions of a recent batch were allowed", written by the evaluation, read here without waiting
    uint32_t* fill_hint_dev = nullptr;  // the same word as the device addresses it
    bool general_earlier = true;     // TCGPU_GENERAL_EARLIER=0: k_eval_general without the earlier-state rule (A/B)
    bool general_runs = true;        // TCGPU_GENERAL_RUNS=0: k_eval_general settles one allowed request per round (A/B)
    bool general_lean = false;       // TCGPU_GENERAL_LEAN=1: the variant compiled without the allowed-runs rule for drained streams (A/B: measured no better, profiles/r06_v19_general_lean_ab.txt)
    bool debug_nostore = false; // builds with -DTCGPU_DEBUG_KNOBS only: TCGPU_DEBUG_NO_DECISION_STORE=1, MEASUREMENT ONLY -- the lean kernel skips its decision bytes (wrong results)
    uint32_t loaded_seq = 0;
    bool kernels_preloaded = false; // slots.hip: preload_pipelined_kernels has run for this engine's device
    // bounds over the registered rate plans (for all_runs_regular)
    int64_t cls_min_ei = INT64_MAX, cls_max_ei = 0, cls_min_dvt = INT64_MAX, cls_max_dvt = 0;
    uint32_t* pend_count = nullptr;
    uint8_t* allowed_tmp = nullptr;
    // Results of the single-request calls land in PINNED host memory, written by the kernel itself (round 5: they used to be
    // fetched with a copy behind the kernel -- a second launch, 7 us after the first: 12-15 of a call's 33 us).  The *_result
    // pointers are the device's view of the block (what the kernels are given), *_host the caller's.
    // A synchronous host batch in PAGEABLE memory, small enough to be bound by latency: the engine copies the arrays through a pinned
    // block of its own and takes the pinned path (slots.hip: bounce_in / bounce_out).  bounce_max = most bytes, in + out, that go
    // this way (TCGPU_BOUNCE_MAX; 0: never)
    uint8_t* bounce = nullptr;
    size_t bounce_max = 4u << 20;
    uint32_t async_copy_kernel_n = 32768;  // TC_B_ASYNC host batches up to this many requests are staged by one copy launch (TCGPU_ASYNC_COPY_KERNEL_N)
    uint8_t* host_results = nullptr;       // hipHostMalloc: OneResult | StoreOpResult | one resolved slot
    StoreOpResult* op_result = nullptr;    // (device view)
    OneResult* one_result = nullptr;       // (device view) tc_rate_limit: the single request's result
    uint32_t* one_slot = nullptr;          // (device view) resolve_one_key
    const StoreOpResult* op_result_host = nullptr;
    const OneResult* one_result_host = nullptr;
    const uint32_t* one_slot_host = nullptr;

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
        int64_t* dict = nullptr;     // TC_B_PLAN_DICT (lazy): 65 536 x 3
        uint16_t* plan_id = nullptr;
        uint32_t* q32 = nullptr;
    } stage;
    std::vector<int64_t> dict_wide[4]; // TC_B_PLAN_DICT batches that take a path without device-side decoding: the wide columns, made on the host

    // small host-pointer batches (k_small_batch): one pinned block the kernel reads its inputs from and writes
    // its results to (the caller's arrays are copied in and out by the host)
    uint8_t* small_io = nullptr;  // host address
    uint8_t* small_io_dev = nullptr; // the same block as the device addresses it
    size_t small_io_bytes = 0;
    size_t small_slots_at = 0;    // key mode: where the last small batch's resolved slots are in it (retry of rejected requests)
    bool small_off = false;       // TCGPU_NO_SMALL_BATCH=1: always take the big pipeline
    bool copy_kernel_off = false; // TCGPU_COPY_KERNEL=0: pinned host inputs of synchronous batches by hipMemcpyAsync, one array at a time (A/B)
    uint64_t host_chunk = 0;      // requests per chunk of a pipelined synchronous host batch (HOST_CHUNK_DEFAULT; TCGPU_HOST_CHUNK, 0: never;
                                  // a multiple of 64: packed decision bits are written in whole words)

    // TC_B_ASYNC host batches still in flight, oldest first: one event per batch, recorded behind its last copy
    std::deque<hipEvent_t> async_done;
    std::vector<hipEvent_t> async_pool;

    // self-cleaning (autosweep.hip): AdaptiveStore::maybe_clean_expired in front of the engine's own mutating calls
    struct AutoSweep {
        uint32_t kind = TC_SWEEP_NONE;
        throttlecrab::sweep::AdaptiveSweep adaptive{0, 0};
        throttlecrab::sweep::PeriodicSweep periodic{0};
        throttlecrab::sweep::ProbabilisticSweep probabilistic{1000};
        // the feed: what the device tells the host after every mutating call, in pinned memory, never waited for (mk::PolicyFeed)
        void* feed_host = nullptr;
        void* feed_dev = nullptr;
        uint64_t issued = 0;              // feeds enqueued so far (the feed of call k carries seq k)
        uint64_t seen = 0;                // seq of the newest feed the host has read
        uint64_t allowed_accounted = 0;   // device total of allowed requests already handed to the policy as operations
        uint64_t swept_accounted = 0;     // device total of removed entries already handed to the policy
        uint64_t host_ops = 0;            // operations only the host knows of (tc_store_* calls: one each), not yet handed over
        uint64_t entries = 0;             // the store's size as of feed `seen` (string mode: bound keys; slot mode: live slots of the last sweep)
        uint64_t free_slots = 0;          // string mode: free slots as of feed `seen`
        int64_t last_now = 0;             // newest timestamp the engine has been shown (a call's own, or the feed's for device columns)
        std::deque<std::pair<uint64_t, uint64_t>> keys_after; // (seq, n) of the key batches issued: those > seen may have taken n slots each
        // how many of a batch's requests were NEW keys lately: the share the feeds reported (inserted keys / requests) over the last
        // looks, and the worst of them -- what the room check expects of the batches it has no numbers for yet
        uint64_t inserted_seen = 0;
        double new_share[8] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
        uint32_t new_share_at = 0;
        // a sweep the engine started whose result the host has not heard yet
        bool pending = false;
        uint64_t pending_seq = 0;         // the first feed that reflects it
        uint64_t pending_entries = 0;     // the store's size when it was started
        tc_sweep_info stats{};
        bool in_retry = false;            // a rejected sub-batch is being applied again: no second retry
        uint64_t chunk_slots_at = UINT64_MAX; // a chunk of a pipelined host key batch: its resolved slots are also kept at e->k_slot + this
        int64_t min_interval_ns = 0;
        int64_t room_quiet_until = INT64_MIN; // an unproductive room sweep is not repeated before the stream's clock gets here
    } as;

    uint64_t batches = 0; // TC_CNT_BATCHES is kept on the host
    uint32_t* poison_host = nullptr; // pinned: raised by tc::invariant_failed; every ABI call checks it (TC_E_INVARIANT)
    bool debug_break_wait = false;   // tc_debug_break_wait
    hipStream_t debug_filler[4] = {};    // tc_debug_occupy: filler streams, one per distinct CU mask seen (up to 4)
    uint32_t debug_filler_mask[4][8] = {};
    uint32_t debug_fillers = 0;
    uint32_t fault_countdown = 0; // tc_debug_fail_copy: the n-th staging copy from now fails (error-path tests)
    uint32_t* route_ws = nullptr; // tc_route_batch scratch (lazy): per stream it may run on: tile counts per destination
    size_t route_ws_words = 0;    // words of one of them
    uint32_t next_route = 0;
    uint32_t route_seq = 0;       // sequence number of the one-pass router's look-back words
    hipEvent_t route_l0_done = nullptr; // the last router that used scratch lane 0 (the caller's stream / the engine's stream) ...
    hipStream_t route_l0_stream = nullptr; // ... and the stream it ran on: a router on ANOTHER stream waits for it (one scratch)
    bool route_l0_used = false;

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
    bool spread_free = false;        // TCGPU_SPREAD_FREE=1: the free stack hands out slots from all over the key space (kt::spread_position), so that key
                                     // batches can take the range path -- measured slower on configs[4] (profiles/r05_v9_spread_free_ab.txt): off
    kt::Table kt;
    void* kt_block = nullptr;        // one allocation backing every kt.* array
    kt::RetiredRec* retired = nullptr; // key mode + TC_CFG_TRACK_DENIED: denial counts of keys without a slot
    uint32_t* sweep_part = nullptr;               // [3][SWEEP_GRID] per-block counts of a sweep (every mode)
    mk::SweepWork sweep_work{};                   // key-mode sweep: the unbound slots per block, those counts, offsets
    uint32_t *k_slot = nullptr, *k_aux = nullptr; // max_batch each
    uint8_t* k_state = nullptr;      // max_batch
    uint32_t* k_claim = nullptr;     // keys first seen in the batch, per k_probe block (+ the batch's total behind them)
    // Key stages mutate the key table and share the scratch above, so they run one after another in
    // call order: on the key stream for TC_B_INPUTS_READY device batches (overlapping the grouping and
    // evaluation of earlier batches), else on the main stream; the two events hand the order across.
    hipStream_t key_stream = nullptr;
    hipEvent_t k_done = nullptr, m_done = nullptr;
    bool k_busy = false, m_busy = false;
    // Round 6 (late): an explicit sweep right behind pipelined key batches can run on the KEY stream, beside the newest batches'
    // evaluations instead of behind them (keys.hip: sweep_keys_device) -- the slots those batches asked for are marked, left out of the
    // scan and looked at once the newest evaluation is done (mk::k_sweep_fixup).  Exact (tests/test_gpu_sweep_aside.py) and OFF by
    // default: the key stream's idle time around a sweep stays what it was (238 -> 242 us in the traces, configs[4] 0-2 % either way,
    // profiles/r06_v45_sweep_aside_ab.txt) -- beside two batches' grouping and evaluation the sweep's kernels take twice as long.
    bool sweep_aside = false;       // TCGPU_SWEEP_ASIDE=1: on
    uint8_t* touched = nullptr;     // [capacity rounded up to 16, + 16] the marks (zero outside such a sweep)
    int aside_set = -1;             // the scratch set of the newest pipelined key batch (its slot column, its evaluation's event);
                                    //   -1: something else has touched the cells or the table since
    uint32_t aside_streak = 0;      // pipelined key batches in a row with nothing else between them (a sweep goes aside from 2: the
                                    // evaluation of the batch before the newest is then the last thing in front of the newest on the engine's stream)
    uint64_t sweeps_aside = 0;
    bool sweep_kernels_preloaded = false; // keys.hip: preload_sweep_kernels
    uint64_t api_seq = 0;           // entry points called so far (TC_CHECK_POISON)
    uint64_t aside_seq = 0;         // ... when the newest pipelined key batch was noted
    hipEvent_t wait_before_sort = nullptr; // piped key batch: the auxiliary sort waits for its key stage
    uint8_t* k_stage_bytes = nullptr; // host-pointer batches: staged key arena
    size_t k_stage_bytes_cap = 0;
    uint32_t* k_stage_off = nullptr;  // max_batch + 1

    // optional per-stage HIP-event timing (tc_profile_*): a (begin, end) event pair per
    // kernel, recorded on the stream the kernel is launched on
    bool prof_on = false;
    bool prof_markers = false;       // TCGPU_PROF_MARKERS=1: time every stage with marker events around its kernels
    std::vector<hipEvent_t> prof_ev; // 2 per record
    std::vector<int> prof_stage;
    size_t prof_used = 0;            // records
    double prof_ms[TC_STAGE_COUNT] = {0};
    uint64_t prof_calls[TC_STAGE_COUNT] = {0};

    std::string err;
};
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

// The same, timed: while per-stage profiling is on (tc_profile_enable) the launch carries a (start, stop) event pair ON
// THE DISPATCH PACKET ITSELF (hipExtLaunchKernelGGL), so the stage's time is the kernel's own execution time -- what
// rocprofv3 --kernel-trace reports for it -- and no marker packet is put around the kernel (hipEventRecord before and
// after every kernel stretched the pipelined run: 37.7 us "per evaluation" where the trace showed 30.0).
#define TC_LAUNCH_T(e, stage, stop, kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                                     \
        hipEvent_t _pa = nullptr, _pb = nullptr;                                                             \
        if ((e)->prof_on && prof_pair((e), (stage), &_pa, &_pb))                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _pa, _pb, 0, __VA_ARGS__);               \
        else TC_LAUNCH(stop, kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)

#define TC_TRY_EARLY(call)                \
    do {                                  \
        int _trc = (call);                \
        if (_trc != TC_E_OK) return _trc; \
    } while (0)

// (every entry point that takes an engine comes through here: `api_seq` lets the sweep beside an evaluation -- keys.hip:
// sweep_keys_device -- see that nothing was called between the newest pipelined key batch and itself)
#define TC_CHECK_POISON(e)              \
    do {                                \
        int _prc = poisoned(e);         \
        if (_prc != TC_E_OK) return _prc; \
        (e)->api_seq++;                 \
    } while (0)

#define TC_TRY(call)                  \
    do {                              \
        int _trc = (call);            \
        if (_trc != TC_E_OK) return _trc; \
    } while (0)

// input arrays of a TC_B_ASYNC host batch, staged inside run_slots_device on the stream that groups the batch
struct HostIn {
    const uint32_t* slot = nullptr;
    const int64_t* col[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // burst, count, period, quantity, now
    // TC_B_PLAN_DICT: the first three columns as a dictionary + 16-bit indices, the quantities as u32 (round 6)
    const int64_t* plan_dict = nullptr;
    const uint16_t* plan_id = nullptr;
    const uint32_t* q32 = nullptr;
    uint32_t n_plans = 0;
};

// ---- helpers shared by the units (defined in the unit named) ----------------------------------------------------
// engine.hip
hipStream_t cur_stream(tc_engine* e);
void prof_begin(tc_engine* e, int stage, hipStream_t s);
void prof_end(tc_engine* e, hipStream_t s);
bool prof_pair(tc_engine* e, int stage, hipEvent_t* start, hipEvent_t* stop); // TC_LAUNCH_T
// stages whose kernels are launched with TC_LAUNCH_T: marker events only with TCGPU_PROF_MARKERS=1 (the old way)
inline void prof_begin_m(tc_engine* e, int stage, hipStream_t s);
inline void prof_end_m(tc_engine* e, hipStream_t s);
hipError_t copy_async(tc_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
int fail(tc_engine* e, int code, const char* msg);
int publish_poison_ptr(tc_engine* e);
int poisoned(tc_engine* e);
int ensure_side_streams(tc_engine* e);
int upload_classes(tc_engine* e);
int intern_class(tc_engine* e, int64_t burst, int64_t count, int64_t period, bool* grew);
// slots.hip
// A large synchronous host-pointer batch whose arrays are PINNED (tc_host_alloc) is pipelined in chunks through the TC_B_ASYNC
// machinery: chunk k + 1's inputs cross PCIe (SDMA, grouping / key stream) while chunk k is evaluated and chunk k - 1's results go
// back -- a sequence of batches in index order is the batch (tcgpu.h: "exactly as if applied one by one"), so results are
// unchanged.  VERDICT r4 #3: the reference-shaped call (host arrays, 55 B in / 32 B out per request) was copy in, compute, copy
// out in a row: 1.93 ms per 1 Mi requests where the transfers alone, overlapped, need 1.1.  Measured (us per 1 Mi reference-
// shaped requests, pinned arrays): uniform chunks of 32 Ki / 64 Ki / 128 Ki / 256 Ki requests 2930 / 2560 / 1670 / 1560, chunks
// that grow from 64 Ki to 256 Ki and shrink again 1670-1720, one piece 1930 -- a chunk costs ~75 us of host time to enqueue and
// its seven transfers are the less efficient the smaller they are, so: chunks of 256 Ki, for batches of at least two of them.
// (Pageable arrays: the runtime stages those copies itself and blocks the caller meanwhile -- nothing to overlap: one piece.)
constexpr uint64_t HOST_CHUNK_DEFAULT = 256 * 1024;
inline std::vector<uint64_t> host_chunk_plan(const tc_engine* e, uint64_t n) {
    std::vector<uint64_t> plan;
    for (uint64_t left = n; left;) {
        const uint64_t c = std::min<uint64_t>(left, e->host_chunk);
        plan.push_back(c);
        left -= c;
    }
    return plan;
}
bool host_arrays_pinned(const tc_batch& b); // slots.hip: every array the batch names is device-visible host memory
// bytes that cross PCIe per request of a host batch, both ways.  Chunking pays by what it overlaps: the reference-shaped call
// (87 B) gains 370 us per 1 Mi requests, a slot batch that asks for `allowed` only (5 B) LOSES 30 (217 us against 187, tools/
// host_slots.py: four enqueues for 5 MB of transfers) -- so only batches that move at least HOST_CHUNK_MIN_BYTES per request.
constexpr uint64_t HOST_CHUNK_MIN_BYTES = 16;
inline uint64_t host_bytes_per_request(const tc_batch& b) {
    uint64_t in = b.slot ? 4 : 0;
    if (b.key_off && b.key_bytes && b.n) in += 4 + (uint64_t(b.key_off[b.n]) - b.key_off[0]) / b.n;
    for (const void* c : {(const void*)b.max_burst, (const void*)b.count_per_period, (const void*)b.period, (const void*)b.quantity, (const void*)b.now_ns})
        in += c ? 8 : 0;
    if (b.flags & TC_B_PLAN_DICT) in += 2 + (b.quantity32 ? 4 : 0);
    uint64_t out = (b.allowed ? 1 : 0) + (b.status ? 1 : 0) + (b.result4 ? 32 : 0) + (b.decisions ? 32 : 0);
    for (const void* c : {(const void*)b.limit, (const void*)b.remaining, (const void*)b.reset_after_ns, (const void*)b.retry_after_ns}) out += c ? 8 : 0;
    return in + out;
}
inline bool host_chunking_applies(const tc_engine* e, const tc_batch& b) {
    return e->host_chunk && b.n >= 2 * e->host_chunk && !(b.flags & (TC_B_DEVICE_PTRS | TC_B_ASYNC | TC_B_GROUPED_OUTPUT | TC_B_UNIQUE_SLOTS)) &&
           host_arrays_pinned(b) && host_bytes_per_request(b) >= HOST_CHUNK_MIN_BYTES;
}
// requests [at, at + cn) of host batch b as a batch of their own (every column and output advanced; key_off stays absolute)
void host_sub_batch(const tc_batch& b, uint64_t at, uint64_t cn, tc_batch& c);
// wait for the last `mine` TC_B_ASYNC batches (the ones this call issued) and hand their events back to the pool
int wait_own_async(tc_engine* e, size_t mine);
// host arrays -> device staging on stream s: ONE copy kernel (mk::k_copy_multi) when every source is pinned host memory, else one
// hipMemcpyAsync each (count <= 8; a failed copy -- tc_debug_fail_copy -- fails the call before anything was applied, as before)
int stage_in_multi(tc_engine* e, const void* const* src, void* const* dst, const size_t* bytes, uint32_t count, hipStream_t s);
// TC_B_PLAN_DICT (round 6).  dict_check: the batch's compact columns are consistent (TC_E_INVALID_ARG otherwise);
// dict_expand_on_host: b's wide columns made on the host (e->dict_wide), the flag cleared -- the paths that do not decode on the
// device (one-launch small batches, retries); dict_expand_device: a DEVICE-pointer batch's, by a kernel on the engine's stream;
// stage_compact: host arrays -> device (dict, ids, u32 quantities into the buffers given, allocated on first use) on stream s,
// then mk::k_expand_plans into out[0..2] (and out[3] if there are u32 quantities)
int dict_check(tc_engine* e, const tc_batch& b);
void dict_expand_on_host(tc_engine* e, tc_batch& b);
int dict_expand_device(tc_engine* e, tc_batch& b);
int stage_compact(tc_engine* e, const int64_t* dict, uint32_t n_plans, const uint16_t* plan_id, const uint32_t* q32, uint32_t n, hipStream_t s, int64_t*& d_dict,
                  uint16_t*& d_id, uint32_t*& d_q32, int64_t* const out[4]);
struct Bounced {
    tc_batch bb;
    struct Out {
        void* user;
        const void* pinned;
        size_t bytes;
    } outs[12];
    int n_out = 0;
    bool on = false;
};
bool bounce_in(tc_engine* e, const tc_batch& b, Bounced& bo); // false: does not apply (bo.on stays false), the batch runs as it is
void bounce_out(const Bounced& bo);                          // the results, from the pinned block into the caller's arrays
int stage_outputs(tc_engine* e, const tc_batch& b, tc_batch& d);
int copy_outputs_back(tc_engine* e, const tc_batch& b, hipStream_t s, bool by_kernel = false);
int run_slots_device(tc_engine* e, const tc_batch& b, const HostIn* hin = nullptr);
void hot_worker_stop(tc_engine* e); // slots.hip: the hot list's maker thread (joined by tc_engine_destroy)
int run_slots_host_staged(tc_engine* e, const tc_batch& b, uint32_t* key_error_flag = nullptr, const uint32_t* d_slot = nullptr, bool columns_staged = false);
int finish_async(tc_engine* e, const tc_batch& b);
bool small_batch_applies(const tc_engine* e, const tc_batch& b);
int run_small_batch(tc_engine* e, const tc_batch& b);
// device staging for host-pointer batches, allocated per array on first use
template <class T>
inline int stage_need(tc_engine* e, T*& p, size_t count) {
    if (!p) TC_HIP(e, hipMalloc(&p, count * sizeof(T)));
    return TC_E_OK;
}
// autosweep.hip
// in front of a mutating call: n requests (key_batch: each may take a free slot), first timestamp `now_ns` (now_known false: the
// timestamps are a device column -- the newest one the feed has shown stands in).  ops_now: operations the call itself is known
// to be (tc_store_* calls: one, counted BEFORE should_clean like adaptive_cleanup.rs:206; the allowed requests of a batch are
// only known afterwards and reach the policy through the feed).  May enqueue a sweep on the engine's stream.
int auto_sweep_before(tc_engine* e, uint64_t n, bool key_batch, bool now_known, int64_t now_ns, uint64_t ops_now = 0);
// behind it: enqueue the feed on the engine's stream (now_last: device address of the call's last timestamp, or nullptr and
// now_scalar)
int auto_sweep_after(tc_engine* e, uint64_t n, bool key_batch, const int64_t* now_last_dev, int64_t now_scalar);
// a synchronous call ran out of slots: sweep at now_ns and wait (the caller applies the rejected requests again)
int auto_sweep_for_retry(tc_engine* e, int64_t now_ns);
inline bool auto_sweep_on(const tc_engine* e) { return e->as.kind != TC_SWEEP_NONE; }
// The sweep beside an evaluation (keys.hip: sweep_keys_device).  `aside_streak` pipelined key batches in a row, with nothing between
// them on the engine's stream but sweeps that went aside themselves (those run on the key stream): the evaluation of the batch before
// the newest is then the last thing in front of the newest batch's on the engine's stream.  `aside_seq` = api_seq when the chain was
// last extended: the call being served continues it only if it is the very next one.
inline void aside_reset(tc_engine* e) { e->aside_set = -1, e->aside_streak = 0; }
inline bool aside_fresh(const tc_engine* e) { return e->aside_streak != 0u && e->api_seq == e->aside_seq + 1; }
inline uint32_t aside_chain(const tc_engine* e) { return aside_fresh(e) ? e->aside_streak : 0u; } // (taken before the batch resets it)
inline void aside_note(tc_engine* e, uint32_t chain, int set_index, uint32_t n) {
    e->aside_set = set_index, e->sets[set_index].k_n = n, e->aside_streak = chain + 1u, e->aside_seq = e->api_seq;
}
// a sweep went aside: the chain goes on, but a second sweep behind the same batch goes behind everything
inline void aside_swept(tc_engine* e) { e->aside_set = -1, e->aside_seq = e->api_seq; }
// maint.hip
int sweep_enqueue(tc_engine* e, int64_t now_ns, bool may_go_aside = false); // tc_sweep_expired without the wait
// keys.hip
int resolve_keys_device(tc_engine* e, const uint8_t* d_bytes, const uint32_t* d_off, uint32_t n, bool insert, uint32_t* out_slot, bool on_key_stream);
int stage_keys(tc_engine* e, const uint8_t* key_bytes, const uint32_t* key_off, uint64_t n, const uint8_t** d_bytes, const uint32_t** d_off,
               const tc_batch* cols = nullptr);
int resolve_one_key(tc_engine* e, const uint8_t* key, size_t key_len, bool insert, uint32_t* slot);
int sweep_keys_device(tc_engine* e, int64_t now_ns, unsigned long long* removed_scratch, bool aside = false);

inline void prof_begin_m(tc_engine* e, int stage, hipStream_t s) {
    if (e->prof_markers) prof_begin(e, stage, s);
}
inline void prof_end_m(tc_engine* e, hipStream_t s) {
    if (e->prof_markers) prof_end(e, s);
}
