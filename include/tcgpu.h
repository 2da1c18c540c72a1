/*
 * tcgpu.h -- C ABI of the MI355X-native batched GCRA engine (libtcgpu.so).
 *
 * This is the drop-in boundary for ONE hot path of lazureykis/throttlecrab:
 *     RateLimiter<AdaptiveStore>::rate_limit
 *         throttlecrab/src/core/rate_limiter.rs:102-250      (decision + advance)
 *         throttlecrab/src/core/rate/mod.rs:164-176          (emission interval)
 *         throttlecrab/src/core/store/mod.rs:85-133          (Store trait)
 *         throttlecrab/src/core/store/adaptive_cleanup.rs:138-279 (AdaptiveStore)
 * The reference has no FFI layer (it is 100 % Rust); the seam is the generic
 * `RateLimiter<S: Store>` that the server calls from
 *     throttlecrab-server/src/actor.rs:178-215 (StoreType::rate_limit).
 * A Rust maintainer binds these symbols with `extern "C"` (INTEGRATION.md
 * shows the stub) and gets `rate_limit` + the new `rate_limit_batch`.
 *
 * Semantics: a batch behaves EXACTLY as if its requests were applied one by
 * one, in index order, through the reference's `rate_limit` (duplicate keys
 * inside a batch included).  Integer results are bit-exact with the reference
 * inside the validated domain; outside it (where the reference panics or is
 * build-mode dependent) the request gets TC_INTERNAL and state is untouched:
 *     now_ns < 0                       (rate_limiter.rs:126-144 reads the wall clock)
 *     Duration * u32 overflow          (rate_limiter.rs:122 panics)
 *     now_ns + dvt overflows i64       (rate_limiter.rs:217 plain add)
 *
 * Ownership / threading: the handle is owned by the caller and is NOT
 * thread-safe (same contract as `&mut RateLimiter`, store/mod.rs:40-43): one
 * caller at a time; results are ordered on one HIP stream per engine (the
 * engine may group and stage batches on internal streams of its own).
 * Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Environment variables.  The library reads a number of TCGPU_* variables when an engine is created.  They are UNSUPPORTED TUNING
 * SWITCHES of the measurement scripts under tools/ (A/B of a kernel variant, a stream count, a threshold): none of them changes a
 * result, none is part of this ABI, any may disappear.  As of round 6:
 *   pipeline     TCGPU_AUX_STREAMS  TCGPU_PIPE_DEPTH  TCGPU_AUX_PRIORITY  TCGPU_PIPE_PROBE  TCGPU_ASSUME_CONCURRENT  TCGPU_STREAM_POOL
 *                TCGPU_STOP_EVENTS  TCGPU_PROF_MARKERS
 *   grouping     TCGPU_RANGE  TCGPU_RANGE_ILV  TCGPU_RANGE_MAX_N  TCGPU_SORT_ITEMS_PIPED  TCGPU_HOT  TCGPU_HOT_MIN  TCGPU_HOT_RANK  TCGPU_HOT_THREAD  TCGPU_BUCKET
 *                TCGPU_BUCKET_PIPED  TCGPU_BUCKET_BACKOFF  TCGPU_BUCKET_MIN_N  TCGPU_BUCKET_SKEW  TCGPU_ROUTE_3PASS
 *   evaluation   TCGPU_EVAL_ITEMS  TCGPU_EVAL_LEAN  TCGPU_PREFILL  TCGPU_GENERAL_EARLIER  TCGPU_GENERAL_RUNS  TCGPU_GENERAL_LEAN
 *                TCGPU_NO_SMALL_BATCH
 *   host batches TCGPU_HOST_CHUNK  TCGPU_BOUNCE_MAX  TCGPU_COPY_KERNEL  TCGPU_ASYNC_COPY_KERNEL_N  TCGPU_SYNC_COPY_MAX
 *   string keys  TCGPU_SPREAD_FREE  TCGPU_SWEEP_ASIDE        multi-GPU  TCGPU_EXCHANGE_WAIT_S
 * One more, TCGPU_DEBUG_NO_DECISION_STORE (the lean kernel skips its decision bytes: wrong results, for timing only), exists
 * only in libraries built with `make DEBUG_KNOBS=1` (-DTCGPU_DEBUG_KNOBS); the library this header ships with ignores it.
 */
#ifndef TCGPU_H
#define TCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCGPU_ABI_VERSION 1

typedef struct tc_engine tc_engine;

/* Per-request status = CellError (throttlecrab/src/core/mod.rs:49-56); 0 = Ok. */
enum {
    TC_OK = 0,
    TC_NEGATIVE_QUANTITY = 1,  /* CellError::NegativeQuantity  (rate_limiter.rs:111-113) */
    TC_INVALID_RATE_LIMIT = 2, /* CellError::InvalidRateLimit  (rate_limiter.rs:115-117) */
    TC_INTERNAL = 3            /* CellError::Internal          (outside validated domain) */
};

/* Call-level return codes (0 = success).  A negative return means the whole
 * call failed and NO request of the batch was applied -- except TC_E_TABLE_FULL:
 * the requests whose keys had or got a slot were applied, the others carry
 * status TC_INTERNAL. */
enum {
    TC_E_OK = 0,
    TC_E_INVALID_ARG = -1,
    TC_E_HIP = -2,            /* HIP runtime error; see tc_last_error().  Raised while a batch's inputs were being staged:
                               * nothing was applied and the engine is usable.  Raised by a RESULT copy (after the
                               * evaluation was enqueued): the batch was applied but its results are lost -- treat the
                               * engine as ahead of the caller (reload a snapshot, or drop it) */
    TC_E_NOMEM = -3,
    TC_E_BATCH_TOO_LARGE = -4,
    TC_E_TABLE_FULL = -5,     /* string mode: no free slot / arena space for a new key */
    TC_E_NO_DEVICE = -6,
    TC_E_UNSUPPORTED = -7,
    TC_E_AGAIN = -9,          /* tc_exchange_* with TC_X_NONBLOCKING: the phase's turn has not come (another rank, or the device,
                               * has to get further first); nothing was done, call again */
    TC_E_INVARIANT = -8       /* a kernel flagged a broken internal invariant (a wait on another workgroup gave up after
                               * 2 s, a closed form met a state it was proven not to meet): results and resident state
                               * since the last call that returned TC_E_OK from tc_synchronize() are UNDEFINED.  Sticky:
                               * every later call on the engine returns it; destroy the engine (reload a snapshot).
                               * Never seen unless there is a bug or the device lost forward progress. */
};

/* tc_config.flags */
#define TC_CFG_KEY_MODE 0x1u /* enable the on-device key->slot hash table (string keys) */
#define TC_CFG_TRACK_DENIED 0x2u /* keep a denial counter per key (4 B/slot) for tc_top_denied: the device-side
                                  * TopDeniedKeys of throttlecrab-server/src/metrics.rs:24-76 */

#define TC_CFG_FIXED_PARAMS 0x4u /* 8-byte resident state: one TAT per key instead of the 16-byte {tat, expiry} cell (half the
                                  * table, twice the keys per memory line).  Sound while a key's (burst, count, period)
                                  * never change -- then every write leaves expiry == tat + dvt (rate_limiter.rs:179-183,
                                  * adaptive_cleanup.rs:237) and the expiry column is redundant.  Hence, in this mode:
                                  *  - plans are registered BEFORE the first request (tc_register_params*; later calls
                                  *    return TC_E_UNSUPPORTED) and every batch carries TC_B_REGISTERED_PARAMS;
                                  *  - a plan needs burst >= 2 after the reference's `as u32` truncation and an emission
                                  *    interval / tolerance below 2^60 ns (else TC_E_UNSUPPORTED at registration);
                                  *  - timestamps >= 2^62 ns (year 2116) get TC_INTERNAL;
                                  *  - tc_rate_limit, the tc_store_* operations (free ttl) and TC_CFG_KEY_MODE (string
                                  *    keys carry their rate with every request) return TC_E_UNSUPPORTED.
                                  * Results and tc_read_state are bit-identical to the 16-byte layout otherwise. */

typedef struct tc_config {
    uint32_t struct_size;     /* = sizeof(tc_config) */
    uint32_t flags;           /* TC_CFG_* */
    int32_t device_id;        /* HIP device ordinal */
    int32_t reserved0;
    uint64_t capacity;        /* number of key slots resident in HBM */
    uint64_t max_batch;       /* largest n accepted by one rate_limit_batch call */
    uint64_t key_arena_bytes; /* string mode: overflow arena for keys longer than 112 bytes (shorter keys are
                               * stored inside their slot's 128-byte record); 0 = max(1 MiB, 4 B/slot) */
} tc_config;

/* tc_batch.flags */
#define TC_B_DEVICE_PTRS 0x1u       /* every pointer in the batch is a device pointer (async on the stream) */
#define TC_B_REGISTERED_PARAMS 0x2u /* use the per-slot (burst,count,period) set by tc_register_params */
#define TC_B_UNIQUE_SLOTS 0x4u      /* caller guarantees no slot occurs twice (skips grouping) */
#define TC_B_INPUTS_READY 0x8u      /* device-pointer batch whose `slot` column is already complete in device memory at
                                     * call time and stays untouched until the call's results are ready: the engine may
                                     * group (sort) it on an internal stream while earlier batches are still being
                                     * evaluated.  Evaluation order, results and visibility on the engine's stream are
                                     * unchanged; without the flag the whole batch runs in order on the engine's stream */

#define TC_B_GROUPED_OUTPUT 0x10u    /* write every output ROW in the order the engine evaluates in (grouped by key,
                                     * index order inside a key) and the request index of each row to `order`:
                                     * output row k belongs to request order[k].  Saves the engine one scattered
                                     * store per request; for consumers that walk all results anyway (a reply
                                     * fan-out).  Without the flag row i belongs to request i, as everywhere else. */

#define TC_B_OUTPUTS_IDLE 0x40u      /* with TC_B_INPUTS_READY: nothing enqueued before this call reads or writes the call's
                                     * output arrays (every batch in flight has arrays of its own, as a consumer that
                                     * reads results later needs anyway): the engine may initialise them on an internal
                                     * stream ahead of the evaluation.  Decisions-only batches then cost one store per
                                     * MINORITY decision instead of one per request.  Results are unchanged. */

/* One batch of requests = the argument list of RateLimiter::rate_limit
 * (rate_limiter.rs:102-110), columnar.  A NULL input column means "use the
 * scalar of the same name for every request". */
struct tc_decision;
#define TC_B_ASYNC 0x20u             /* HOST-pointer slot batch that only enqueues: the inputs are copied to the device,
                                     * evaluated and the outputs copied back asynchronously, overlapping the PCIe
                                     * transfers of one batch with the evaluation of others.  Input and output
                                     * arrays must stay untouched / are not valid until tc_wait_batches() says the
                                     * batch is done (or tc_synchronize()).  For the copies to be asynchronous the
                                     * arrays must be pinned (tc_host_alloc); with pageable memory the call is
                                     * correct but waits for its transfers.  Results are those of the in-order
                                     * sequence of calls, as always. */

#define TC_B_PLAN_DICT 0x80u         /* Round 6: the (max_burst, count_per_period, period) of every request as a 16-bit index into a
                                      * DICTIONARY the batch carries: plan_dict[n_plans][3] + plan_id[n] replace the three 8-byte
                                      * columns (which must be NULL), and quantity32[n], if given, replaces `quantity` -- 6 bytes per
                                      * request instead of 32.  A server's requests carry a handful of distinct triples; the
                                      * reference's signature rate_limit(key, max_burst, count_per_period, period, quantity, now)
                                      * (rate_limiter.rs:102-110) stays what the SHIMS take (Rust rate_limit_batch(&[Request]), the
                                      * C++ RateLimiter, the actor): they encode, falling back to the wide columns for a batch of more
                                      * than 65 536 distinct triples.  Same results as the wide form, request by request.  An index
                                      * >= n_plans decodes to (0, 0, 0): status TC_INVALID_RATE_LIMIT.  Host- and device-pointer
                                      * batches, slots and keys; not with TC_B_REGISTERED_PARAMS. */

typedef struct tc_batch {
    uint32_t struct_size; /* = sizeof(tc_batch) */
    uint32_t flags;       /* TC_B_* */
    uint64_t n;           /* number of requests */

    /* key: slots (tc_rate_limit_batch_slots) or a key arena (…_keys) */
    const uint32_t* slot;      /* [n] slot ids in [0, capacity) */
    const uint8_t* key_bytes;  /* key arena */
    const uint32_t* key_off;   /* [n+1] offsets into key_bytes */

    const int64_t* max_burst;        /* [n] or NULL */
    const int64_t* count_per_period; /* [n] or NULL */
    const int64_t* period;           /* [n] or NULL (seconds) */
    const int64_t* quantity;         /* [n] or NULL */
    const int64_t* now_ns;           /* [n] or NULL; SystemTime as ns since UNIX_EPOCH */
    int64_t max_burst_scalar;
    int64_t count_per_period_scalar;
    int64_t period_scalar;
    int64_t quantity_scalar;
    int64_t now_ns_scalar;

    /* outputs = (bool, RateLimitResult) (rate_limiter.rs:13-22) + status; any may be NULL */
    uint8_t* allowed;        /* [n] 0/1 */
    uint64_t* allowed_bits;  /* [(n+63)/64] bit i%64 of word i/64 = allowed[i] */
    int64_t* limit;          /* [n] */
    int64_t* remaining;      /* [n] */
    int64_t* reset_after_ns; /* [n] Duration as ns */
    int64_t* retry_after_ns; /* [n] Duration as ns */
    uint8_t* status;         /* [n] TC_OK / TC_NEGATIVE_QUANTITY / ... */
    /* RateLimitResult as ONE 32-byte record per request, {limit, remaining, reset_after_ns,
     * retry_after_ns}: a request's result then costs one scattered store instead of four
     * (use it instead of the four columns above when all fields are wanted). 16-byte aligned. */
    int64_t* result4;        /* [n][4] */
    struct tc_decision* decisions; /* [n] 32-byte records (see tc_decision below); 16-byte aligned */
    uint32_t* order;         /* [n] TC_B_GROUPED_OUTPUT: request index of each output row */

    /* A slot column that arrives in PIECES (tc_rate_limit_batch_slots, TC_B_DEVICE_PTRS): n_segments > 0 replaces
     * `slot` by seg_slot[0..n_segments) -- HOST arrays of device pointers and lengths, sum(seg_n) == n -- taken one
     * after the other: request i of the batch is the i-th entry of the concatenation.  What a shard of a multi-GPU
     * deployment receives: one segment per source GPU, in source order, so that a key's requests keep the order
     * of the global stream (tc_route_batch with only = -1 produces the segments).  The engine gathers the pieces on
     * the stream that groups the batch; with TC_B_INPUTS_READY every piece is complete in device memory at call time.
     * At most TC_MAX_SEGMENTS pieces. */
    uint32_t n_segments;
    uint32_t reserved_seg;
    const uint32_t* const* seg_slot; /* host array [n_segments] of device pointers */
    const uint32_t* seg_n;           /* host array [n_segments] */

    /* TC_B_PLAN_DICT (round 6; a tc_batch of an older struct_size simply ends before these) */
    const int64_t* plan_dict;   /* [n_plans][3]: max_burst, count_per_period, period (seconds) */
    const uint16_t* plan_id;    /* [n] index into plan_dict */
    const uint32_t* quantity32; /* [n] or NULL: the quantities as u32 (then `quantity` must be NULL) */
    uint32_t n_plans;           /* 1 .. 65536 */
    uint32_t reserved_dict;
} tc_batch;
#define TC_MAX_SEGMENTS 64

/* Everything rate_limit returns for one request except `limit` (== the request's max_burst, resp.
 * the key's registered burst), as ONE 32-byte record: a full result then costs a single scattered
 * store per request instead of three (result4 + allowed + status). */
typedef struct tc_decision {
    int64_t remaining;
    int64_t reset_after_ns;
    int64_t retry_after_ns;
    uint8_t allowed; /* 0/1 */
    uint8_t status;  /* TC_OK / TC_NEGATIVE_QUANTITY / ... */
    uint8_t pad[6];  /* zero */
} tc_decision;

/* Single-request result (the tuple rate_limit returns). */
typedef struct tc_result {
    int64_t limit;
    int64_t remaining;
    int64_t reset_after_ns;
    int64_t retry_after_ns;
    uint8_t allowed;
    uint8_t status;
} tc_result;

/* Counter block (the payload all-gathered across GPUs; cf. the reference's
 * Metrics counters, throttlecrab-server/src/metrics.rs:84-94). */
enum {
    TC_CNT_TOTAL = 0,    /* requests decided */
    TC_CNT_ALLOWED = 1,
    TC_CNT_DENIED = 2,
    TC_CNT_ERRORS = 3,   /* status != TC_OK */
    TC_CNT_SWEPT = 4,    /* entries removed by tc_sweep_expired */
    TC_CNT_BATCHES = 5,
    TC_CNT_KEYS_INSERTED = 6,
    TC_CNT_LIVE_SLOTS = 7, /* the store's size, AdaptiveStore::len(): string mode -- the keys that hold a slot, exact in tc_counters();
                            * slot mode (and the device-resident block) -- occupied slots as of the last tc_sweep_expired */
    TC_CNT_COUNT = 8
};

uint32_t tc_abi_version(void);

/* RateLimiter::new(AdaptiveStore::with_capacity(..)) (rate_limiter.rs:56-58,
 * adaptive_cleanup.rs:93-106).  Returns NULL and sets *err on failure. */
tc_engine* tc_engine_create(const tc_config* cfg, int* err);
void tc_engine_destroy(tc_engine* e);

/* Issue all work on `hip_stream` (a hipStream_t) instead of the engine's own
 * stream; NULL restores the engine's stream. */
int tc_engine_set_stream(tc_engine* e, void* hip_stream);
/* Block until everything issued so far has finished. */
int tc_synchronize(tc_engine* e);
/* Message for the last negative return code (owned by the engine). */
const char* tc_last_error(const tc_engine* e);

/* Per-slot rate parameters kept as columns in HBM (emission interval, burst
 * tolerance, burst capacity).  slots == NULL registers slots [0, n).  Host
 * pointers.  Invalid parameter triples are rejected (TC_E_INVALID_ARG). */
int tc_register_params(tc_engine* e, uint64_t n, const uint32_t* slots, const int64_t* max_burst,
                       const int64_t* count_per_period, const int64_t* period);
/* Same triple for every slot. */
int tc_register_params_uniform(tc_engine* e, int64_t max_burst, int64_t count_per_period, int64_t period);

/* rate_limit_batch over pre-resolved slots / over string keys.
 * Host-pointer batches return after the results are in the output arrays (up to 1024 requests are served by
 * one kernel launch, ~30-40 us per call; larger ones by the grouping + evaluation pipeline), unless flagged
 * TC_B_ASYNC; TC_B_DEVICE_PTRS batches are asynchronous on the engine's stream. */
int tc_rate_limit_batch_slots(tc_engine* e, const tc_batch* b);
int tc_rate_limit_batch_keys(tc_engine* e, const tc_batch* b);

/* TC_B_ASYNC batches: block until at most `max_in_flight` of them are still incomplete (0 = all done).
 * A caller that cycles through K sets of pinned buffers calls tc_wait_batches(e, K - 1) before it
 * refills the oldest set.  (The reference's actor answers one request per loop turn,
 * throttlecrab-server/src/actor.rs:217-236; this is what lets a batch-draining actor fill the next
 * batch while the previous ones are in flight.) */
int tc_wait_batches(tc_engine* e, uint32_t max_in_flight);
/* Pinned host memory (hipHostMalloc / hipHostFree behind a C signature): what TC_B_ASYNC batches need, and what makes a
 * synchronous host-pointer batch fastest -- its arrays cross PCIe in one copy launch each way, and from 512 Ki requests on the
 * batch is pipelined in chunks.  Arrays in pageable memory are always correct: up to 4 MB of them per call go through a pinned
 * block of the engine's own (two memcpys by the calling thread), larger ones through the runtime's staging copies. */
void* tc_host_alloc(size_t bytes);
void tc_host_free(void* p);

/* RateLimiter::rate_limit (rate_limiter.rs:102-250), one request, string key.
 * Returns 0 and fills *out (out->status carries the CellError). */
int tc_rate_limit(tc_engine* e, const uint8_t* key, size_t key_len, int64_t max_burst,
                  int64_t count_per_period, int64_t period, int64_t quantity, int64_t now_ns,
                  tc_result* out);

/* AdaptiveStore::cleanup (adaptive_cleanup.rs:173-203): drop every entry with
 * expiry <= now.  Decision-neutral.  removed == NULL: the sweep is only enqueued on the
 * engine's stream (no host wait; the number removed is added to TC_CNT_SWEPT). */
int tc_sweep_expired(tc_engine* e, int64_t now_ns, uint64_t* removed);

/* ---- self-cleaning: AdaptiveStore::maybe_clean_expired behind the engine's own calls -----------------------------------
 * The reference's stores clean THEMSELVES: every compare_and_swap_with_ttl / set_if_not_exists_with_ttl first runs
 * maybe_clean_expired (adaptive_cleanup.rs:205-211,229,262; periodic.rs:128-142; probabilistic.rs:110-125), so a
 * `RateLimiter<AdaptiveStore>` never fills up with expired keys.  With a policy set, the engine does the same in front of
 * every mutating call (tc_rate_limit, tc_rate_limit_batch_keys / _slots, tc_store_compare_and_swap_with_ttl,
 * tc_store_set_if_not_exists_with_ttl): it counts the reference's operations (one per allowed request / store call), looks
 * at the call's timestamp and at the store's size and, when the reference's should_clean would say yes, enqueues
 * tc_sweep_expired at that timestamp in front of the call -- on the engine's stream, nothing waits.  A batch is one step:
 * the trigger is evaluated once per call with the call's first timestamp, where the reference evaluates it per request.
 * Cleanup never changes a decision while timestamps do not decrease, so results stay bit-exact.
 * String mode adds what a fixed-size table needs where the reference's HashMap would grow:
 *   - room: a key batch that might not find n free slots is preceded by a sweep (and, if the numbers the host holds are
 *     older than that sweep, by one wait for fresh ones);
 *   - retry: a SYNCHRONOUS call that still ran out of slots (TC_E_TABLE_FULL) sweeps at the newest timestamp of the call and
 *     applies the rejected requests once more, in order, before it reports the error.
 * Without a policy (the default of tc_engine_create) the engine never cleans by itself: tc_sweep_expired is the
 * caller's.  The host mirrors (throttlecrab_gpu.hpp GpuStore, rust/throttlecrab-gpu) set TC_SWEEP_ADAPTIVE with the
 * server's defaults (throttlecrab-server/src/config.rs:285-304) when they create their engine. */
enum {
    TC_SWEEP_NONE = 0,
    TC_SWEEP_ADAPTIVE = 1,      /* adaptive_cleanup.rs:138-211 */
    TC_SWEEP_PERIODIC = 2,      /* periodic.rs:128-142 */
    TC_SWEEP_PROBABILISTIC = 3  /* probabilistic.rs:110-125 */
};
typedef struct tc_sweep_policy {
    uint32_t struct_size;         /* = sizeof(tc_sweep_policy) */
    uint32_t kind;                /* TC_SWEEP_* */
    int64_t created_ns;           /* the store's creation time (`SystemTime::now()` in with_capacity): the first time-based
                                   * cleanup is due one interval later (adaptive: 5 s, DEFAULT_CLEANUP_INTERVAL_SECS) */
    int64_t min_interval_ns;      /* adaptive; 0 = 1 s (MIN_CLEANUP_INTERVAL_SECS; the server's default is 5 s) */
    int64_t max_interval_ns;      /* adaptive; 0 = 300 s */
    int64_t interval_ns;          /* periodic; 0 = 60 s */
    uint64_t max_operations;      /* adaptive; 0 = 100 000 (MAX_OPERATIONS_BEFORE_CLEANUP; the server's default is 1 000 000) */
    uint64_t map_capacity;        /* adaptive: the `capacity` of AdaptiveStore::with_capacity -- the memory-pressure trigger
                                   * fires above 3/4 of 1.3 x this many entries; 0 = the engine's capacity / 1.3, i.e. the
                                   * trigger fires when 3/4 of the engine's slots are taken */
    uint64_t cleanup_probability; /* probabilistic; 0 = 1000 */
} tc_sweep_policy;
/* p == NULL or kind == TC_SWEEP_NONE: the engine stops cleaning by itself. */
int tc_set_sweep_policy(tc_engine* e, const tc_sweep_policy* p);
typedef struct tc_sweep_info {
    uint32_t struct_size;          /* = sizeof(tc_sweep_info), set by the caller */
    uint32_t kind;                 /* the policy in force */
    uint64_t sweeps;               /* sweeps the engine started by itself since the policy was set ... */
    uint64_t sweeps_by_time;       /* ... because the interval had passed (adaptive_cleanup.rs:140, periodic.rs:129) */
    uint64_t sweeps_by_operations; /* ... the operation count (:145) or the probabilistic draw (probabilistic.rs:116) */
    uint64_t sweeps_by_size;       /* ... the size of the store (:150-168) */
    uint64_t sweeps_for_room;      /* ... a key batch that might not have found free slots */
    uint64_t retries;              /* synchronous calls that ran out of slots, swept and applied the rejected requests again */
    uint64_t feed_waits;           /* times a call waited for fresh numbers from the device (room checks only) */
    uint64_t operations;           /* operations counted since the last cleanup */
    uint64_t entries;              /* the store's size as the host last saw it (string mode: bound keys) */
    uint64_t last_removed;         /* entries the last cleanup the host has heard of removed */
    int64_t current_interval_ns;   /* adaptive: current_cleanup_interval */
    int64_t next_cleanup_ns;       /* adaptive / periodic: next_cleanup */
} tc_sweep_info;
int tc_sweep_stats(tc_engine* e, tc_sweep_info* out);

/* Copy the counter block to host (refreshes it first). */
int tc_counters(tc_engine* e, uint64_t out[TC_CNT_COUNT]);
/* Fold the engine's sharded decision counters into the device-resident counter
 * block (asynchronous on the stream).  Call before reading the block through
 * tc_counters_device_ptr, e.g. right before the RCCL all-gather. */
int tc_counters_refresh(tc_engine* e);
/* Per-stage timing of the batch pipeline with HIP events recorded on the
 * engine's stream between its kernels (diagnostics for the roofline report;
 * leave off in production -- every event is an extra stream operation). */
enum {
    TC_STAGE_PREP = 0,   /* digit histograms of the batch (rs::k_hist) */
    TC_STAGE_SORT = 1,   /* one radix pass (rs::k_onesweep); counted once per pass */
    TC_STAGE_EVAL = 2,   /* GCRA decide + advance */
    TC_STAGE_COMMIT = 3, /* deferred cell stores */
    TC_STAGE_PACK = 4,   /* decision bit packing */
    TC_STAGE_HASH = 5,   /* string mode: key -> slot resolution */
    /* uniform batches are grouped WITHOUT a sort when their keys are spread evenly (bucket path): */
    TC_STAGE_BUCKET_HIST = 6,    /* requests per key-range bucket, per 4096-request tile (bp::k_tile_hist) */
    TC_STAGE_BUCKET_SCAN = 7,    /* prefix of those counts over the tiles (bp::k_bucket_scan) */
    TC_STAGE_BUCKET_SCATTER = 8, /* stable partition of the batch by bucket (bp::k_scatter) */
    TC_STAGE_BUCKET_EVAL = 9,    /* rank + GCRA decide + advance, one wave per bucket (bp::k_bucket_eval) */
    TC_STAGE_COUNT = 10
};
int tc_profile_enable(tc_engine* e, int on);
/* Accumulated milliseconds and launches per stage since tc_profile_enable. */
int tc_profile_read(tc_engine* e, double total_ms[TC_STAGE_COUNT], uint64_t calls[TC_STAGE_COUNT]);

/* Device address of the counter block (uint64[TC_CNT_COUNT]) for an on-device
 * collective (RCCL all-gather) without a host round trip. */
int tc_counters_device_ptr(tc_engine* e, void** dptr);

/* `trait Store` parity shims (store/mod.rs:85-133), one key per call. */
int tc_store_get(tc_engine* e, const uint8_t* key, size_t key_len, int64_t now_ns, int64_t* value, int* found);
int tc_store_compare_and_swap_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t old_value,
                                       int64_t new_value, uint64_t ttl_ns, int64_t now_ns, int* swapped);
int tc_store_set_if_not_exists_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t value,
                                        uint64_t ttl_ns, int64_t now_ns, int* was_set);

/* Introspection for differential tests: raw (tat, expiry) columns of slots
 * [first, first+n) copied to host.  expiry is ns since epoch saturated to u64;
 * 0 = vacant. */
int tc_read_state(tc_engine* e, uint64_t first, uint64_t n, int64_t* tat, uint64_t* expiry);
/* string mode: slot currently bound to `key`, or -1. */
int tc_lookup_slot(tc_engine* e, const uint8_t* key, size_t key_len, int64_t* slot);

/* Top denied keys (metrics.rs:24-76,296-309: `throttlecrab_top_denied_keys{key,rank}`).  Needs
 * TC_CFG_TRACK_DENIED.  Counts are exact (the reference's capped HashMap forgets keys when it
 * overflows).  Slot mode: a slot IS the key.  Key mode: this call reports the keys that currently
 * hold a slot; tc_top_denied_keys also knows the keys a sweep has unbound (the reference counts by
 * key, whatever its store cleans up).  Returns the k slots with the most denials since creation / tc_denied_reset,
 * most denied first (ties: lower slot first), k is capped at 10 000 like the reference's
 * MAX_DENIED_KEYS_LIMIT; slots never denied are not listed.  *n_out <= k entries are written. */
int tc_top_denied(tc_engine* e, uint32_t k, uint32_t* slots, uint64_t* counts, uint32_t* n_out);
int tc_denied_reset(tc_engine* e);
/* string mode: TopDeniedKeys::get_top (metrics.rs:66-76) -- the k most denied KEYS, most denied first (ties: key
 * bytes ascending), as a key arena (key_off[k + 1]) + counts.  A key's denials follow the key: a sweep that unbinds it
 * moves the count into a side table of 65 536 keys (keys of up to 256 bytes, the reference's MAX_KEY_LENGTH; trimmed
 * to the 10 000 most denied ones once it passes 30 000, like TopDeniedKeys::cleanup), and binding the key again
 * moves it back.  k is capped at 10 000.  TC_E_INVALID_ARG if key_bytes_cap is too small. */
int tc_top_denied_keys(tc_engine* e, uint32_t k, uint8_t* key_bytes, size_t key_bytes_cap, uint32_t* key_off, uint64_t* counts,
                       uint32_t* n_out);
/* string mode: the keys bound to `slots` as an arena (key_off[n+1]); an unbound slot yields an
 * empty key.  TC_E_INVALID_ARG if key_bytes_cap is too small. */
int tc_slot_keys(tc_engine* e, uint32_t n, const uint32_t* slots, uint8_t* key_bytes, size_t key_bytes_cap,
                 uint32_t* key_off);

/* ---- multi-GPU: routing a global request stream to the GPU that owns each key -----------------------------
 * The reference has no distributed mode ("use client-side sharding by key", README.md:247-249).  A key space of
 * world * keys_per_shard GLOBAL ids is sharded without a routing table by a bijection of that range:
 *     q = id div world, r = id mod world:   shard-local slot = q,   owner = (r + mix(q)) mod world
 * (mix(q) = the top bits of q * 0x9E3779B1 scaled to [0, world): ids that share a residue are still spread over all
 * owners) -- every shard gets exactly keys_per_shard dense slots.  tc_route_batch keeps, of a batch of global ids
 * in device memory, the requests a destination owns (stable: the requests of a key keep their order) as
 * shard-local slots ready for tc_rate_limit_batch_slots; asynchronous on the engine's stream (or on r->stream).
 * No collective is involved: each GPU filters the stream it is handed. */
/* tc_route.flags */
#define TC_ROUTE_AHEAD 0x1u   /* run the router on one of the engine's grouping streams, ordered behind the grouping of
                               * every batch whose call has returned (the last reader of a slot column -- except for
                               * TC_B_UNIQUE_SLOTS batches, whose evaluation reads it -- so out_slot may be a buffer an
                               * earlier batch call was given); successive routers rotate over those streams and do
                               * not wait for one another (give each its own out_* buffers): it then
                               * works beside the evaluations of earlier batches instead of between them -- with
                               * out_count_host, the way to route several global batches ahead of the evaluation */
#define TC_ROUTE_NO_READERS 0x2u /* with TC_ROUTE_AHEAD: out_slot is not the slot column of any batch still in flight (the
                               * caller forwards the segments elsewhere, tc_forward_segments): the router waits for nothing */
typedef struct tc_route {
    uint32_t struct_size;     /* = sizeof(tc_route) */
    uint32_t world;           /* number of shards (GPUs), 1..64 */
    uint64_t keys_per_shard;  /* slots of every shard (the engines' capacity) */
    uint64_t n;               /* requests in the global batch (<= the engine's max_batch * world) */
    const uint32_t* global_id; /* [n] device: global key ids in [0, world * keys_per_shard) (others: taken modulo) */
    int32_t only;             /* >= 0: write that destination's requests only, from out_slot[0];
                               * -1: every destination's segment, one after the other (start of d = sum of counts before d) */
    uint32_t flags;           /* TC_ROUTE_* */
    uint32_t* out_slot;       /* [n] device: shard-local slots */
    uint32_t* out_pos;        /* [n] device or NULL: position of each kept request in the global batch */
    uint32_t* out_count;      /* [world] device: requests per destination (all of them, whatever `only` is) */
    void* stream;             /* NULL: the engine's stream.  Else the hipStream_t to enqueue the router on (the caller
                               * orders that stream before the batch call that reads out_slot; successive
                               * tc_route_batch calls must use one stream) */
    uint32_t* out_count_host; /* NULL or [world + 1] PINNED host memory (tc_host_alloc): the router's last block
                               * writes the counts there and, after them, `tag` into out_count_host[world].  A caller
                               * that polls that word needs no stream synchronisation to learn how many requests it
                               * owns: once the tag is there, out_slot / out_pos are complete too. */
    uint32_t tag;             /* any value the previous use of out_count_host did not leave there */
    uint32_t reserved1;
    uint32_t* const* out_dst; /* NULL, or (only == -1) a HOST array [world] of device pointers: segment d is written to
                               * out_dst[d], from its start, instead of into out_slot (which may then be NULL; out_pos is
                               * not written) -- out_dst[d] is the inbox destination d keeps for this source, in d's own
                               * memory (peer memory over xGMI, hipIpcOpenMemHandle): routing and forwarding in one pass.
                               * Once out_count_host carries the tag, every segment has landed. */
} tc_route;
int tc_route_batch(tc_engine* e, const tc_route* r);
/* The exchange step of a sharded deployment in which every GPU routes only ITS slice of the global stream
 * (tc_route_batch with only = -1: one segment per destination, one after the other in out_slot): copy segment d to
 * dst[d] -- the inbox the destination keeps for this source, in its own memory (peer memory over xGMI, mapped
 * through hipIpcOpenMemHandle; dst[own rank] is local) -- in ONE launch on `stream` (NULL: the engine's stream).
 * count[] is the host copy of tc_route.out_count (out_count_host).  Peer copies, no collective: the decision path
 * of the reference has none either (README.md:247-249 shards on the client).  The destination evaluates the
 * inboxes of a step as one batch with a segmented slot column (tc_batch.seg_slot), sources in rank order. */
typedef struct tc_forward {
    uint32_t struct_size;      /* = sizeof(tc_forward) */
    uint32_t world;            /* 1..64 */
    const uint32_t* src;       /* device: the router's out_slot */
    const uint32_t* count;     /* HOST [world]: requests per destination */
    uint32_t* const* dst;      /* HOST [world]: device pointers, dst[d] has room for count[d] entries */
    void* stream;              /* hipStream_t or NULL */
} tc_forward;
int tc_forward_segments(tc_engine* e, const tc_forward* f);

/* ---- the exchange as library calls (one rank's side; `--route exchange` of bench.py) -----------------------------------
 * Every rank owns `ring` inbox slots per source in its own HBM ([ring][world][seg_cap] u32, shared with the peers through
 * hipIpcGetMemHandle / hipIpcOpenMemHandle once, at set-up) and the ranks share one small host-memory block: mail
 * [world][ring][world][2] = (count, step + 1) written by a source once its segment has landed, and done[world] = steps a
 * destination has evaluated (which frees its inbox slots).  Per step i a rank
 *   routes   its slice of global step i + route_ahead straight into the owners' inboxes (tc_route_batch, only = -1, out_dst),
 *   posts    step i + post_ahead: (count, step) into every destination's mailbox once the router's tag is in,
 *   collects the world mailbox words of step i and
 *   evaluates the world inboxes as ONE batch whose slot column comes in pieces (sources in rank order).
 * tc_exchange_step is those four in one call.  No collective, no host synchronisation in steady state (a phase only waits
 * when its inputs are not there yet).  The caller primes the pipeline: routes 0 .. route_ahead - 1, posts 0 .. post_ahead - 1.
 * tmpl: outputs, rate plan / quantity / timestamp and flags of the evaluation (device pointers; TC_B_INPUTS_READY is added);
 * a step larger than the engine's max_batch is evaluated in chunks, each writing behind the previous one.  tmpl->n, if not 0, is
 * the number of requests the output arrays hold: a step (tc_shard_evaluate: this rank's share of it) that is larger fails with
 * TC_E_INVALID_ARG before anything is applied.  */
#define TC_X_NONBLOCKING 0x1u    /* a phase whose turn has not come returns TC_E_AGAIN instead of waiting (ONE thread driving
                                  * several ranks, as the tests do); a repeated phase of a step already done is a no-op */
typedef struct tc_exchange tc_exchange;
typedef struct tc_exchange_config {
    uint32_t struct_size;      /* = sizeof(tc_exchange_config) */
    uint32_t rank;             /* this shard */
    uint32_t world;            /* shards, 1..64 */
    uint32_t ring;             /* inbox slots per source (>= 2; a source may run ring - 1 steps ahead of a destination) */
    uint32_t seg_cap;          /* requests one source may send per step (the slice length) */
    uint32_t flags;            /* TC_X_* */
    uint64_t keys_per_shard;
    uint32_t* const* inbox;    /* HOST array [world]: inbox[d] = destination d's [ring][world][seg_cap] array as THIS process
                                * addresses it (own: device memory; peers: hipIpcOpenMemHandle) */
    uint32_t* mail;            /* host memory shared by all ranks, zero at set-up: [world][ring][world][2] */
    int64_t* done;             /* ... [world] */
} tc_exchange_config;
/* (an exchange points into its engine: destroy it first.  A step may be routed at most 8 steps ahead of its post.) */
int tc_exchange_create(tc_engine* e, const tc_exchange_config* c, tc_exchange** out);
int tc_exchange_destroy(tc_exchange* x);
int tc_exchange_route(tc_exchange* x, uint64_t step, const uint32_t* global_id, uint32_t n);
int tc_exchange_post(tc_exchange* x, uint64_t step);
int tc_exchange_collect(tc_exchange* x, uint64_t step, uint32_t* counts /* [world] or NULL */);
int tc_exchange_evaluate(tc_exchange* x, uint64_t step, const tc_batch* tmpl, uint64_t* decided);
int tc_exchange_step(tc_exchange* x, uint64_t step, const uint32_t* global_id_ahead, uint32_t n_ahead, uint32_t route_ahead,
                     uint32_t post_ahead, const tc_batch* tmpl, uint64_t* decided);
/* publish the steps whose evaluation has completed (frees inbox slots for the sources); never waits */
int tc_exchange_poll(tc_exchange* x);
/* host time (ns) the calls have spent WAITING since the last call of this function: [0] for free inbox slots (route), [1] for
 * the router's tag (post), [2] for the sources' mailbox words (collect) -- what tells a host-bound step from a starved one */
int tc_exchange_wait_ns(tc_exchange* x, uint64_t out[3]);

/* ---- `replicate` mode as library calls (one rank's side; bench.py --route replicate) -------------------------------------------
 * Every rank is handed the WHOLE global batch; per step i it
 *   routes    global step i + route_ahead: keeps what it owns as shard-local slots (tc_route_batch, one destination) on one of the
 *             engine's grouping streams, beside the evaluations, into one of `ring` columns of its own; the router's last block
 *             leaves the counts and the step's tag in pinned host memory,
 *   evaluates step i: polls that tag (routed a few steps ago: no wait in steady state), and decides its share -- in batches of
 *             at most max_batch, each writing its outputs behind the previous one's.
 * tc_shard_step is both in one call: a rank's host cost per step is one call (round 4: four Python calls, 26-38 us).  The caller
 * primes the pipeline: routes steps 0 .. route_ahead - 1.  tmpl: as for tc_exchange_evaluate.  No collective, no peer memory. */
typedef struct tc_shard tc_shard;
typedef struct tc_shard_config {
    uint32_t struct_size;     /* = sizeof(tc_shard_config) */
    uint32_t rank;            /* this shard */
    uint32_t world;           /* shards, 1..64 */
    uint32_t ring;            /* routed batches kept (>= route_ahead + 2, at most 64) */
    uint64_t keys_per_shard;  /* the engines' capacity */
    uint64_t max_global;      /* requests of the largest global batch */
} tc_shard_config;
/* (a shard points into its engine: destroy it first) */
int tc_shard_create(tc_engine* e, const tc_shard_config* c, tc_shard** out);
int tc_shard_destroy(tc_shard* x);
int tc_shard_route(tc_shard* x, uint64_t step, const uint32_t* global_id, uint64_t n);
int tc_shard_evaluate(tc_shard* x, uint64_t step, const tc_batch* tmpl, uint64_t* decided);
int tc_shard_step(tc_shard* x, uint64_t step, const uint32_t* global_id_ahead, uint64_t n_ahead, uint32_t route_ahead, const tc_batch* tmpl,
                  uint64_t* decided);
/* host time (ns) the calls have spent waiting for a router's tag since the last call of this function */
int tc_shard_wait_ns(tc_shard* x, uint64_t* out);

/* The same map on the host: owner and shard-local slot of n global ids (either output may be NULL), and its
 * inverse (global id of slot `slot` of shard `owner`).  No device needed. */
int tc_route_host(uint32_t world, uint64_t keys_per_shard, uint64_t n, const uint32_t* global_id, uint32_t* owner,
                  uint32_t* slot);
int tc_route_inverse(uint32_t world, uint64_t keys_per_shard, uint64_t n, const uint32_t* owner, const uint32_t* slot,
                     uint64_t* global_id);

/* String keys across GPUs (README.md:247-249: "client-side sharding by key"): owner[i] = mix64(hash(key i) ^ salt) mod world --
 * the front door of a sharded deployment of key-mode engines: split a key batch by owner, hand every GPU the keys it owns.
 * key_off[n + 1] delimits the keys in key_bytes.  Host only. */
int tc_route_keys_host(uint32_t world, uint64_t n, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t* owner);

/* ---- pipeline health --------------------------------------------------------------------------------------------------------
 * Pipelined batches (TC_B_INPUTS_READY, TC_B_ASYNC, pinned host batches in chunks) are fast because their grouping runs on
 * internal streams BESIDE the evaluation of earlier batches.  Which hardware queue and dispatch pipe a stream lands on depends on
 * everything the process did with the GPU before, so the engine probes candidates against its main stream
 * and keeps those that really run concurrently -- and when it finds none (GPU_MAX_HW_QUEUES too small, a process full of streams)
 * the batches still give the same results, in order on the main stream, 1.5-2.5 x slower, without any error.  This call says
 * which of the two it is. */
typedef struct tc_engine_info {
    uint32_t struct_size;          /* = sizeof(tc_engine_info), set by the caller */
    uint32_t side_streams_probed;  /* 0: no pipelined batch yet on the current main stream -- nothing below is known */
    uint32_t grouping_streams_wanted;
    uint32_t grouping_streams;     /* kept after the probe; 0: pipelined batches run in order on the main stream */
    uint32_t key_stream;           /* string mode: 1 if key stages have a stream of their own */
    uint32_t candidates_tried;     /* streams created and probed */
    uint32_t rejected_same_queue;  /* ... that did not run while the main stream (or a kept stream) was running */
    uint32_t rejected_same_pipe;   /* ... that ran, but were handed out behind the main stream's workgroups */
    uint32_t kept_second_best;     /* kept although they share a dispatch pipe with ANOTHER kept stream (never with the main one) */
    uint32_t probes_assumed;       /* 1: TCGPU_ASSUME_CONCURRENT -- candidates taken unprobed (profiler counter passes) */
    uint32_t pipelining_degraded;  /* 1: fewer grouping streams than wanted (or no key stream): expect the slower figures */
    uint32_t scratch_sets;         /* batches whose grouping may be in flight at once */
    uint32_t grouping_path;        /* of the last batch: 0 none yet, 1 range path (2 launches), 2 LSD passes, 3 bucket path, 4 small batch / unique,
                                    * 5 range path with the hot slots peeled out (3 launches; skewed streams, round 6) */
    uint32_t range_path_possible;  /* the key space admits the range path */
    uint64_t range_hint_requests;  /* the range hint the next batch goes by: requests of the batch it came from ... */
    uint64_t range_hint_largest;   /* ... and its largest key range (a block finishes up to 4 096 requests in LDS) */
    uint64_t host_chunk_requests;  /* pinned synchronous host batches of at least twice this many requests are pipelined in chunks; 0: never */
    uint64_t batches;              /* batches decided so far */
    uint64_t hot_slots;            /* slots on the hot list right now (made from the evaluations' notes on long runs; 0: none) */
    uint64_t hot_batches;          /* batches grouped with the hot slots peeled out so far */
    uint64_t probes_pooled;        /* 1: the grouping streams came from the process's pool -- streams an earlier engine on this main stream
                                    * had probed, re-checked against the main stream instead of probing sixteen new candidates */
    uint64_t sweeps_aside;         /* string mode, round 6: explicit sweeps (tc_sweep_expired right behind a pipelined key batch) that ran on
                                    * the key stream BESIDE that batch's evaluation instead of behind it (TCGPU_SWEEP_ASIDE=0: never).  A caller
                                    * built against the struct without this field passes its smaller struct_size and gets the fields it knows */
} tc_engine_info;
int tc_engine_info_get(tc_engine* e, tc_engine_info* out);

/* Number of internal invariant violations the kernels have flagged since creation (always 0
 * unless there is a bug; the parity tests assert it). */
int tc_selfcheck(tc_engine* e, uint64_t* violations);

/* Test hook (error paths): the nth host <-> device staging copy from now fails with TC_E_HIP instead of being
 * issued; 0 disarms.  Not for production use. */
int tc_debug_fail_copy(tc_engine* e, uint32_t nth);
/* Test hook (watchdog): the next uniform batch's evaluation withholds the "row 0 has read its cells" announcement,
 * so a key whose requests span more than one row makes its owner wait until the watchdog gives up (2 s) and the
 * engine is poisoned (TC_E_INVARIANT).  Not for production use. */
int tc_debug_break_wait(tc_engine* e, uint32_t on);

/* Test hook (forward progress): a FILLER kernel -- `blocks` workgroups of 256 threads, `lds_bytes` of LDS each, every one of
 * which stays resident for `microseconds` -- enqueued on a stream of its own that is restricted to the compute units of
 * cu_mask (8 words, bit i = CU i as hipExtStreamCreateWithCUMask counts them; NULL: every CU).  Returns at once.  The
 * kernels that wait for other workgroups (radix look-back, direct stores, the general path's chain, the one-pass router)
 * wait only for workgroups dispatched EARLIER, so taking wave slots, LDS or whole CUs away from them must only ever make them
 * slower: tests/test_gpu_robustness.py runs them against fillers and checks results and watchdog.  Not for production use. */
int tc_debug_occupy(tc_engine* e, const uint32_t* cu_mask, uint32_t blocks, uint32_t lds_bytes, uint64_t microseconds);
/* Test hook (string keys): walks the key table both ways after draining the engine -- every bound slot must be reachable
 * through the entry its position column names (the entry binds that slot and carries the hash of the slot's key record),
 * every binding entry must name a bound slot that points back at it, no claim may be left pending, and bound slots + free
 * slots must add up to the capacity.  *inconsistencies = how many of these fail (0 on a
 * healthy table).  TC_E_UNSUPPORTED without TC_CFG_KEY_MODE.  Not for production use (two passes over the table). */
int tc_debug_check_keys(tc_engine* e, uint64_t* inconsistencies);

/* Checkpoint / restore of everything resident (state cells, rate plans, denial counters, in
 * string mode the key table, plus the counter block).  The reference keeps its state in memory
 * only and loses it on restart; here a snapshot is a few device-to-host copies.  Load needs an
 * engine created with the same capacity and flags.  Both calls drain the engine first. */
int tc_snapshot_save(tc_engine* e, const char* path);
int tc_snapshot_load(tc_engine* e, const char* path);

#ifdef __cplusplus
}
#endif
#endif /* TCGPU_H */
