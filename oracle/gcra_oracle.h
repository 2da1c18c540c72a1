/*
 * gcra_oracle.h -- CPU oracle for the throttlecrab GCRA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * `RateLimiter<AdaptiveStore>::rate_limit` path, used as the checker in tests/,
 * in __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing in
 * the product path (throttlecrab_amd/, include/tcgpu.h) may link, import or
 * call it.
 *
 * Parity status: PINNED against the reference's own known-answer tests
 * (tests/golden/reference_kat.json, transcribed from
 *  throttlecrab/src/core/tests.rs, store/store_test_suite.rs,
 *  throttlecrab-server/src/transport/redis_test.rs, actor_tests.rs, grpc.rs);
 * the reference itself (Rust) cannot be compiled in this image (no cargo/rustc),
 * so there is no oracle/_ref build.
 *
 * Reference files restated (paths relative to /root/reference):
 *   throttlecrab/src/core/rate_limiter.rs:102-250     tco_rate_limit
 *   throttlecrab/src/core/rate/mod.rs:164-176         tco_emission_interval
 *   throttlecrab/src/core/store/mod.rs:85-133         tco_store_vt (Store trait)
 *   throttlecrab/src/core/store/adaptive_cleanup.rs   tco_adaptive_* (AdaptiveStore)
 *   throttlecrab/src/core/mod.rs:49-56                TCO_* status codes (CellError)
 */
#ifndef GCRA_ORACLE_H
#define GCRA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CellError taxonomy, throttlecrab/src/core/mod.rs:49-56 (0 = Ok). */
enum {
    TCO_OK = 0,
    TCO_NEGATIVE_QUANTITY = 1,
    TCO_INVALID_RATE_LIMIT = 2,
    TCO_INTERNAL = 3
};

/* (bool, RateLimitResult), rate_limiter.rs:13-22; Durations as u64 ns. */
typedef struct tco_result {
    int64_t limit;
    int64_t remaining;
    uint64_t reset_after_ns;
    uint64_t retry_after_ns;
    uint8_t allowed;
    uint8_t status;
} tco_result;

/* `trait Store`, store/mod.rs:85-133.  `now` = SystemTime as i64 ns since
 * UNIX_EPOCH, `ttl` = Duration as u64 ns.  Return 0 or a negative error
 * (the trait's Err(String), mapped to CellError::Internal by the caller). */
typedef struct tco_store_vt {
    int (*get)(void* self, const uint8_t* key, size_t klen, int64_t now, int64_t* val, int* found);
    int (*cas_ttl)(void* self, const uint8_t* key, size_t klen, int64_t old_v, int64_t new_v,
                   uint64_t ttl, int64_t now, int* ok);
    int (*set_nx_ttl)(void* self, const uint8_t* key, size_t klen, int64_t val, uint64_t ttl,
                      int64_t now, int* ok);
} tco_store_vt;

typedef struct tco_store {
    const tco_store_vt* vt;
    void* self;
} tco_store;

/* Rate::from_count_and_period, rate/mod.rs:164-176: emission interval in ns
 * (u64, saturating f64->u64 cast).  Invalid args -> u64::MAX secs is never
 * reached on the hot path (arguments are validated first). */
uint64_t tco_emission_interval(int64_t count, int64_t period);

/* Derived per-request constants (rate_limiter.rs:119-123,154-155).
 * Returns TCO_OK, or TCO_INTERNAL where the reference panics / is
 * build-mode dependent (Duration*u32 overflow; now<0; now+dvt overflow). */
int tco_derive(int64_t burst, int64_t count, int64_t period, int64_t now, int64_t* ei, int64_t* dvt);

/* RateLimiter::rate_limit, rate_limiter.rs:102-250, over any Store. */
int tco_rate_limit(tco_store* st, const uint8_t* key, size_t klen, int64_t max_burst,
                   int64_t count_per_period, int64_t period, int64_t quantity, int64_t now,
                   tco_result* out);

/* ---- AdaptiveStore (string keyed), adaptive_cleanup.rs ------------------ */
typedef struct tco_adaptive tco_adaptive;

/* AdaptiveStore::with_config (adaptive_cleanup.rs:117-136).  `created_ns`
 * stands in for the SystemTime::now() the reference reads at construction. */
tco_adaptive* tco_adaptive_new(size_t capacity, uint64_t min_interval_ns, uint64_t max_interval_ns,
                               size_t max_operations, int64_t created_ns);
/* AdaptiveStore::with_capacity defaults (adaptive_cleanup.rs:10-16,93-106). */
tco_adaptive* tco_adaptive_with_capacity(size_t capacity, int64_t created_ns);
void tco_adaptive_free(tco_adaptive*);
tco_store tco_adaptive_as_store(tco_adaptive*);
size_t tco_adaptive_len(const tco_adaptive*);
uint64_t tco_adaptive_cleanups(const tco_adaptive*);       /* number of cleanup() runs */
void tco_adaptive_force_cleanup(tco_adaptive*, int64_t now); /* AdaptiveStore::cleanup */
/* Test knob (no reference analogue): on = 0 makes maybe_clean_expired a no-op so
 * that cleanup only happens through tco_adaptive_force_cleanup.  Needed to compare
 * streams whose timestamps go backwards, where the reference's own results depend
 * on when its cleanup heuristics fire. */
void tco_adaptive_set_auto_cleanup(tco_adaptive*, int on);

/* ---- Dense slot store (keys are u32 slot ids; no hashing) --------------- */
typedef struct tco_dense tco_dense;
tco_dense* tco_dense_new(size_t capacity);
void tco_dense_free(tco_dense*);
tco_store tco_dense_as_store(tco_dense*);
/* raw cell access for differential tests: expiry saturated to u64 */
void tco_dense_peek(const tco_dense*, uint32_t slot, int64_t* tat, uint64_t* expiry_sat, int* occupied);
/* ... of slots [first, first + n) at once */
void tco_dense_dump(const tco_dense*, size_t first, size_t n, int64_t* tat, uint64_t* expiry_sat, uint8_t* occupied);
/* cleanup(): vacate every cell with expiry <= now; returns how many */
uint64_t tco_dense_sweep(tco_dense*, int64_t now);
size_t tco_dense_live(const tco_dense*);

/* ---- Batch drivers (sequential, index order) ---------------------------- */
/* Parameter arrays are read at index i*stride (stride 0 = broadcast scalar,
 * 1 = per request).  Output arrays may be NULL. */
typedef struct tco_batch_io {
    size_t n;
    const int64_t* max_burst; size_t burst_stride;
    const int64_t* count_per_period; size_t count_stride;
    const int64_t* period; size_t period_stride;
    const int64_t* quantity; size_t quantity_stride;
    const int64_t* now_ns; size_t now_stride;
    uint8_t* allowed;
    int64_t* limit;
    int64_t* remaining;
    int64_t* reset_after_ns;
    int64_t* retry_after_ns;
    uint8_t* status;
} tco_batch_io;

/* keys: arena bytes + offsets[n+1] */
void tco_batch_keys(tco_store* st, const uint8_t* key_bytes, const uint32_t* key_off,
                    const tco_batch_io* io);
/* slots: key i is the 4-byte little-endian slot id */
void tco_batch_slots(tco_store* st, const uint32_t* slot, const tco_batch_io* io);
/* tco_batch_slots over `threads` threads, requests partitioned by slot (same results; full-size GPU tests) */
void tco_batch_slots_mt(tco_store*, const uint32_t* slot, const tco_batch_io*, int threads);

/* Hash-sharded multi-thread CPU baseline: T AdaptiveStores, key i handled by
 * thread (hash(key) % T); each thread walks the whole stream in index order
 * and serves its own keys.  Returns elapsed seconds (steady clock). */
double tco_batch_keys_mt(int threads, size_t capacity_per_thread, uint64_t max_operations, int64_t created_ns,
                         const uint8_t* key_bytes, const uint32_t* key_off,
                         const tco_batch_io* io);

/* `format!("{prefix}{id}")` key arena for a slot-id stream; returns bytes used
 * (0 if out_cap is too small). */
size_t tco_format_keys(const char* prefix, const uint32_t* ids, size_t n, uint8_t* out_bytes,
                       size_t out_cap, uint32_t* out_off);

/* 64-bit key hash used only to place keys (results are hash independent). */
uint64_t tco_hash_bytes(const uint8_t* p, size_t n);
/* the reference's own store_comparison.rs loop (see gcra_oracle.c); returns seconds */
double tco_reference_shape(size_t num_keys, size_t iterations, uint64_t* allowed, uint64_t* blocked);

#ifdef __cplusplus
}
#endif
#endif
