/*
 * gcra_oracle.c -- CPU oracle for the throttlecrab GCRA hot path (TEST
 * INFRASTRUCTURE ONLY; see gcra_oracle.h for the rules and the parity status).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Rust semantics that matter and how they are kept:
 *   - i64 saturating_add/sub/mul            -> sat_add/sat_sub/sat_mul below
 *   - `x as u64` from f64                    -> saturating, NaN -> 0
 *   - `(max_burst - 1) as u32`               -> low 32 bits of the i64
 *   - `Duration * u32`                       -> checked; the reference PANICS on
 *                                               overflow -> status Internal here
 *   - `Duration::as_nanos() as i64`          -> u128 -> i64 truncation (wrap)
 *   - `now_ns + dvt` (plain add, :217)       -> panics in debug / wraps in
 *                                               release -> status Internal here,
 *                                               decided BEFORE the store is touched
 *   - SystemTime + Duration (expiry)         -> exact, kept as unsigned __int128 ns
 */
#define _POSIX_C_SOURCE 200809L
#include "gcra_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ---- Rust integer semantics -------------------------------------------- */
static inline int64_t sat_add(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_add_overflow(a, b, &r)) return b > 0 ? INT64_MAX : INT64_MIN;
    return r;
}
static inline int64_t sat_sub(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_sub_overflow(a, b, &r)) return b < 0 ? INT64_MAX : INT64_MIN;
    return r;
}
static inline int64_t sat_mul(int64_t a, int64_t b) {
    int64_t r;
    if (__builtin_mul_overflow(a, b, &r)) return ((a < 0) != (b < 0)) ? INT64_MIN : INT64_MAX;
    return r;
}
static inline int64_t max_i64(int64_t a, int64_t b) { return a > b ? a : b; }

static inline uint64_t f64_to_u64_sat(double x) {
    if (!(x > 0.0)) return 0; /* NaN, -x, 0 */
    if (x >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)x;
}

/* rate/mod.rs:164-176 -- `(period as f64 * 1e9 / count as f64) as u64`.
 * Two IEEE-754 roundings (mul, then div); compiled with -ffp-contract=off. */
uint64_t tco_emission_interval(int64_t count, int64_t period) {
    volatile double p = (double)period * 1000000000.0;
    double pn = p / (double)count;
    return f64_to_u64_sat(pn);
}

/* rate_limiter.rs:119-123 (rate, emission_interval, dvt = ei * (burst-1) as u32),
 * :126-144 (now -> ns), :154-155 (as_nanos() as i64). */
int tco_derive(int64_t burst, int64_t count, int64_t period, int64_t now, int64_t* ei, int64_t* dvt) {
    uint64_t ei_u = tco_emission_interval(count, period);
    uint32_t mult = (uint32_t)(uint64_t)(burst - 1);
    /* core::time Duration::checked_mul(u32): secs*rhs + (nanos*rhs)/1e9 must fit u64,
     * otherwise `Duration * u32` panics ("overflow when multiplying duration by scalar"). */
    uint64_t secs = ei_u / 1000000000ull, nanos = ei_u % 1000000000ull;
    uint64_t extra = (nanos * (uint64_t)mult) / 1000000000ull, s;
    if (__builtin_mul_overflow(secs, (uint64_t)mult, &s) || __builtin_add_overflow(s, extra, &s))
        return TCO_INTERNAL;
    u128 dvt128 = (u128)ei_u * (u128)mult;
    *ei = (int64_t)ei_u;               /* as_nanos() as i64: truncating */
    *dvt = (int64_t)(uint64_t)dvt128;  /* as_nanos() as i64: truncating */
    /* :126-144: a pre-1970 `now` makes the reference read the wall clock
     * (non-deterministic) -> outside the validated domain. */
    if (now < 0) return TCO_INTERNAL;
    int64_t tmp;
    /* :217 `now_ns + delay_variation_tolerance_ns` is a plain add. */
    if (__builtin_add_overflow(now, *dvt, &tmp)) return TCO_INTERNAL;
    return TCO_OK;
}

/* rate_limiter.rs:102-250 */
int tco_rate_limit(tco_store* st, const uint8_t* key, size_t klen, int64_t max_burst,
                   int64_t count_per_period, int64_t period, int64_t quantity, int64_t now,
                   tco_result* out) {
    memset(out, 0, sizeof *out);
    if (quantity < 0) { /* :111-113 */
        out->status = TCO_NEGATIVE_QUANTITY;
        return TCO_NEGATIVE_QUANTITY;
    }
    if (max_burst <= 0 || count_per_period <= 0 || period <= 0) { /* :115-117 */
        out->status = TCO_INVALID_RATE_LIMIT;
        return TCO_INVALID_RATE_LIMIT;
    }
    int64_t ei, dvt;
    if (tco_derive(max_burst, count_per_period, period, now, &ei, &dvt) != TCO_OK) {
        out->status = TCO_INTERNAL;
        return TCO_INTERNAL;
    }
    const int64_t limit = max_burst; /* :123 */
    const int64_t now_ns = now;

    int retries = 0; /* :147-149 */
    for (;;) {
        int64_t stored = 0;
        int found = 0;
        if (st->vt->get(st->self, key, klen, now, &stored, &found) != 0) { /* :151 */
            out->status = TCO_INTERNAL;
            return TCO_INTERNAL;
        }
        int64_t tat;
        if (found) { /* :158-161 */
            int64_t min_tat = sat_sub(now_ns, dvt);
            tat = max_i64(stored, min_tat);
        } else { /* :162-166 */
            tat = sat_sub(now_ns, ei);
        }
        int64_t increment = sat_mul(ei, quantity); /* :170 */
        int64_t new_tat = sat_add(tat, increment); /* :171 */
        int64_t allow_at = sat_sub(new_tat, dvt);  /* :174 */
        int allowed = now_ns >= allow_at;          /* :175 */

        if (allowed) { /* :177-205 */
            uint64_t ttl = (uint64_t)sat_add(sat_sub(new_tat, now_ns), dvt); /* :179-183 `as u64` */
            int ok = 0, rc;
            if (found)
                rc = st->vt->cas_ttl(st->self, key, klen, stored, new_tat, ttl, now, &ok); /* :186-189 */
            else
                rc = st->vt->set_nx_ttl(st->self, key, klen, new_tat, ttl, now, &ok); /* :190-195 */
            if (rc != 0) {
                out->status = TCO_INTERNAL;
                return TCO_INTERNAL;
            }
            if (!ok) { /* :197-204 */
                if (++retries >= 10) {
                    out->status = TCO_INTERNAL;
                    return TCO_INTERNAL;
                }
                continue;
            }
        }

        int64_t current_tat = allowed ? new_tat : tat;           /* :208 */
        int64_t burst_limit = now_ns + dvt;                       /* :217 (no overflow: tco_derive) */
        int64_t room = sat_sub(burst_limit, current_tat);         /* :218 */
        int64_t remaining = ei > 0 ? max_i64(room / ei, 0) : 0;  /* :221-225 */
        int64_t ra = max_i64(sat_add(sat_sub(current_tat, now_ns), dvt), 0); /* :227-232 */
        int64_t rt = allowed ? 0 : max_i64(sat_sub(allow_at, now_ns), 0);    /* :234-238 */

        out->allowed = (uint8_t)allowed;
        out->status = TCO_OK;
        out->limit = limit;
        out->remaining = remaining;
        out->reset_after_ns = (uint64_t)ra;
        out->retry_after_ns = (uint64_t)rt;
        return TCO_OK;
    }
}

/* ---- key hash (placement only) ------------------------------------------ */
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
uint64_t tco_hash_bytes(const uint8_t* p, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)n;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = mix64(h ^ w) + 0x9e3779b97f4a7c15ull;
        p += 8; n -= 8;
    }
    if (n) {
        uint64_t w = 0;
        memcpy(&w, p, n);
        h = mix64(h ^ w ^ ((uint64_t)n << 56));
    }
    return mix64(h);
}

/* ---- AdaptiveStore ------------------------------------------------------- */
/* adaptive_cleanup.rs:39-53: HashMap<String,(i64, Option<SystemTime>)> + cleanup
 * bookkeeping.  The map is an open-addressed table of heap-allocated keys so the
 * cost profile (hash + probe per call, key allocation per insert) resembles the
 * reference's; hashbrown's exact capacity growth is approximated (7/8 load) --
 * it only feeds the decision-neutral memory-pressure trigger (:166). */
typedef struct {
    uint8_t* key;   /* NULL = empty */
    uint32_t klen;
    uint64_t hash;
    int64_t val;
    u128 expiry;    /* Some(SystemTime) in ns; AdaptiveStore never stores None */
} ad_entry;

struct tco_adaptive {
    ad_entry* tab;
    size_t buckets; /* power of two */
    size_t len;
    /* cleanup timing (:41-45) */
    u128 next_cleanup;
    uint64_t min_interval, max_interval, cur_interval;
    /* triggers (:46-49) */
    size_t expired_count, ops_since_cleanup, max_ops;
    /* history (:50-52) */
    size_t last_removed, last_total;
    uint64_t cleanups;
    int auto_off; /* test knob: 1 = maybe_clean_expired never cleans (explicit cleanup only) */
};

static size_t cap_to_buckets(size_t cap) { /* hashbrown capacity_to_buckets */
    if (cap < 4) return 4;
    if (cap < 8) return 8;
    size_t adj = cap * 8 / 7, b = 1;
    while (b < adj) b <<= 1;
    return b;
}
static size_t buckets_to_cap(size_t b) { return b < 8 ? b - 1 : (b / 8) * 7; }

tco_adaptive* tco_adaptive_new(size_t capacity, uint64_t min_interval_ns, uint64_t max_interval_ns,
                               size_t max_operations, int64_t created_ns) {
    tco_adaptive* s = (tco_adaptive*)calloc(1, sizeof *s);
    if (!s) return NULL;
    size_t want = (size_t)((double)capacity * 1.3); /* CAPACITY_OVERHEAD_FACTOR :11,:95,:124 */
    s->buckets = cap_to_buckets(want);
    s->tab = (ad_entry*)calloc(s->buckets, sizeof(ad_entry));
    if (!s->tab) { free(s); return NULL; }
    s->cur_interval = 5ull * 1000000000ull; /* DEFAULT_CLEANUP_INTERVAL_SECS :14 */
    s->next_cleanup = (u128)(created_ns < 0 ? 0 : created_ns) + s->cur_interval; /* :96,:125 */
    s->min_interval = min_interval_ns;
    s->max_interval = max_interval_ns;
    s->max_ops = max_operations;
    return s;
}
tco_adaptive* tco_adaptive_with_capacity(size_t capacity, int64_t created_ns) {
    /* :12-15: min 1 s, max 300 s, 100 000 ops */
    return tco_adaptive_new(capacity, 1000000000ull, 300ull * 1000000000ull, 100000, created_ns);
}
void tco_adaptive_free(tco_adaptive* s) {
    if (!s) return;
    for (size_t i = 0; i < s->buckets; i++) free(s->tab[i].key);
    free(s->tab);
    free(s);
}
size_t tco_adaptive_len(const tco_adaptive* s) { return s->len; }
uint64_t tco_adaptive_cleanups(const tco_adaptive* s) { return s->cleanups; }

static ad_entry* ad_find(tco_adaptive* s, const uint8_t* key, size_t klen, uint64_t h) {
    size_t mask = s->buckets - 1, i = (size_t)h & mask;
    for (;;) {
        ad_entry* e = &s->tab[i];
        if (!e->key) return NULL;
        if (e->hash == h && e->klen == klen && memcmp(e->key, key, klen) == 0) return e;
        i = (i + 1) & mask;
    }
}
static void ad_place(ad_entry* tab, size_t buckets, ad_entry ent) {
    size_t mask = buckets - 1, i = (size_t)ent.hash & mask;
    while (tab[i].key) i = (i + 1) & mask;
    tab[i] = ent;
}
static void ad_grow(tco_adaptive* s) {
    size_t nb = s->buckets * 2;
    ad_entry* nt = (ad_entry*)calloc(nb, sizeof(ad_entry));
    for (size_t i = 0; i < s->buckets; i++)
        if (s->tab[i].key) ad_place(nt, nb, s->tab[i]);
    free(s->tab);
    s->tab = nt;
    s->buckets = nb;
}
/* HashMap::insert(key.to_string(), (val, Some(expiry))) -- :238,:270,:275.  The
 * reference allocates a fresh String every time; when the key is already present
 * the map keeps its old key and drops the new one (alloc + free). */
static void ad_insert(tco_adaptive* s, const uint8_t* key, size_t klen, uint64_t h, int64_t val, u128 expiry) {
    uint8_t* copy = (uint8_t*)malloc(klen ? klen : 1);
    memcpy(copy, key, klen);
    ad_entry* e = ad_find(s, key, klen, h);
    if (e) {
        e->val = val;
        e->expiry = expiry;
        free(copy);
        return;
    }
    if (s->len + 1 > buckets_to_cap(s->buckets)) ad_grow(s);
    ad_entry ent;
    ent.key = copy; ent.klen = (uint32_t)klen; ent.hash = h; ent.val = val; ent.expiry = expiry;
    ad_place(s->tab, s->buckets, ent);
    s->len++;
}

/* adaptive_cleanup.rs:138-171 */
static int ad_should_clean(const tco_adaptive* s, int64_t now) {
    if ((u128)now >= s->next_cleanup) return 1;                 /* :140 */
    if (s->ops_since_cleanup >= s->max_ops) return 1;           /* :145 */
    if (s->expired_count > 50) {                                /* :150-163 */
        double ratio = (double)s->expired_count / (double)(s->len ? s->len : 1);
        double thr = (s->last_removed > s->last_total / 4) ? 0.2 / 2.0 : 0.2 * 1.25;
        if (ratio > thr) return 1;
    }
    if (s->len > buckets_to_cap(s->buckets) * 3 / 4) return 1;  /* :166 */
    return 0;
}
/* adaptive_cleanup.rs:173-203 */
static void ad_cleanup(tco_adaptive* s, int64_t now) {
    size_t initial = s->len;
    ad_entry* nt = (ad_entry*)calloc(s->buckets, sizeof(ad_entry));
    size_t kept = 0;
    for (size_t i = 0; i < s->buckets; i++) {
        ad_entry* e = &s->tab[i];
        if (!e->key) continue;
        if (e->expiry > (u128)now) { /* retain(|exp| *exp > now) :176-182 */
            ad_place(nt, s->buckets, *e);
            kept++;
        } else {
            free(e->key);
        }
    }
    free(s->tab);
    s->tab = nt;
    s->len = kept;
    size_t removed = initial - kept;
    if (removed == 0 && s->expired_count == 0) { /* :187-190 */
        uint64_t d = s->cur_interval * 2;
        s->cur_interval = d < s->max_interval ? d : s->max_interval;
    } else if ((double)removed > (double)initial * 0.5) { /* :191-195 */
        uint64_t d = s->cur_interval / 2;
        s->cur_interval = d > s->min_interval ? d : s->min_interval;
    }
    s->last_removed = removed; /* :198-202 */
    s->last_total = initial;
    s->next_cleanup = (u128)now + s->cur_interval;
    s->expired_count = 0;
    s->ops_since_cleanup = 0;
    s->cleanups++;
}
void tco_adaptive_force_cleanup(tco_adaptive* s, int64_t now) { ad_cleanup(s, now); }
/* adaptive_cleanup.rs:205-211 */
static void ad_maybe_clean(tco_adaptive* s, int64_t now) {
    s->ops_since_cleanup++;
    if (!s->auto_off && ad_should_clean(s, now)) ad_cleanup(s, now);
}
void tco_adaptive_set_auto_cleanup(tco_adaptive* s, int on) { s->auto_off = !on; }

/* adaptive_cleanup.rs:246-252 */
static int ad_get(void* self, const uint8_t* key, size_t klen, int64_t now, int64_t* val, int* found) {
    tco_adaptive* s = (tco_adaptive*)self;
    ad_entry* e = ad_find(s, key, klen, tco_hash_bytes(key, klen));
    if (e && e->expiry > (u128)now) {
        *val = e->val;
        *found = 1;
    } else {
        *found = 0;
    }
    return 0;
}
/* adaptive_cleanup.rs:221-244 */
static int ad_cas(void* self, const uint8_t* key, size_t klen, int64_t old_v, int64_t new_v,
                  uint64_t ttl, int64_t now, int* ok) {
    tco_adaptive* s = (tco_adaptive*)self;
    ad_maybe_clean(s, now); /* :229 */
    uint64_t h = tco_hash_bytes(key, klen);
    ad_entry* e = ad_find(s, key, klen, h);
    if (e && e->expiry <= (u128)now) { /* :232-235 */
        s->expired_count++;
        *ok = 0;
    } else if (e && e->val == old_v) { /* :236-240 */
        ad_insert(s, key, klen, h, new_v, (u128)now + ttl);
        *ok = 1;
    } else { /* :241-242 */
        *ok = 0;
    }
    return 0;
}
/* adaptive_cleanup.rs:254-278 */
static int ad_set_nx(void* self, const uint8_t* key, size_t klen, int64_t val, uint64_t ttl,
                     int64_t now, int* ok) {
    tco_adaptive* s = (tco_adaptive*)self;
    ad_maybe_clean(s, now); /* :261 */
    uint64_t h = tco_hash_bytes(key, klen);
    ad_entry* e = ad_find(s, key, klen, h);
    if (e && e->expiry > (u128)now) { /* :264 live -> false */
        *ok = 0;
    } else {
        if (e) s->expired_count++; /* :266-267 */
        ad_insert(s, key, klen, h, val, (u128)now + ttl); /* :268-276 */
        *ok = 1;
    }
    return 0;
}
static const tco_store_vt AD_VT = {ad_get, ad_cas, ad_set_nx};
tco_store tco_adaptive_as_store(tco_adaptive* s) {
    tco_store st;
    st.vt = &AD_VT;
    st.self = s;
    return st;
}

/* ---- Dense slot store ---------------------------------------------------- */
/* Same get/CAS/set_nx rules (adaptive_cleanup.rs:220-279) over a slot-indexed
 * array; the key is the 4-byte little-endian slot id.  Differential partner for
 * the GPU engine's slot mode (no hashing, no cleanup heuristics). */
typedef struct {
    int64_t val;
    u128 expiry;
    uint8_t occupied;
} dn_cell;
struct tco_dense {
    dn_cell* cells;
    size_t capacity;
};
tco_dense* tco_dense_new(size_t capacity) {
    tco_dense* d = (tco_dense*)calloc(1, sizeof *d);
    d->cells = (dn_cell*)calloc(capacity ? capacity : 1, sizeof(dn_cell));
    d->capacity = capacity;
    return d;
}
void tco_dense_free(tco_dense* d) {
    if (!d) return;
    free(d->cells);
    free(d);
}
static dn_cell* dn_at(tco_dense* d, const uint8_t* key, size_t klen) {
    uint32_t slot;
    if (klen != 4) return NULL;
    memcpy(&slot, key, 4);
    if (slot >= d->capacity) return NULL;
    return &d->cells[slot];
}
static int dn_get(void* self, const uint8_t* key, size_t klen, int64_t now, int64_t* val, int* found) {
    dn_cell* c = dn_at((tco_dense*)self, key, klen);
    if (!c) return -1;
    if (c->occupied && c->expiry > (u128)now) {
        *val = c->val;
        *found = 1;
    } else {
        *found = 0;
    }
    return 0;
}
static int dn_cas(void* self, const uint8_t* key, size_t klen, int64_t old_v, int64_t new_v,
                  uint64_t ttl, int64_t now, int* ok) {
    dn_cell* c = dn_at((tco_dense*)self, key, klen);
    if (!c) return -1;
    if (c->occupied && c->expiry <= (u128)now) {
        *ok = 0;
    } else if (c->occupied && c->val == old_v) {
        c->val = new_v;
        c->expiry = (u128)now + ttl;
        *ok = 1;
    } else {
        *ok = 0;
    }
    return 0;
}
static int dn_set_nx(void* self, const uint8_t* key, size_t klen, int64_t val, uint64_t ttl,
                     int64_t now, int* ok) {
    dn_cell* c = dn_at((tco_dense*)self, key, klen);
    if (!c) return -1;
    if (c->occupied && c->expiry > (u128)now) {
        *ok = 0;
    } else {
        c->occupied = 1;
        c->val = val;
        c->expiry = (u128)now + ttl;
        *ok = 1;
    }
    return 0;
}
static const tco_store_vt DN_VT = {dn_get, dn_cas, dn_set_nx};
tco_store tco_dense_as_store(tco_dense* d) {
    tco_store st;
    st.vt = &DN_VT;
    st.self = d;
    return st;
}
void tco_dense_peek(const tco_dense* d, uint32_t slot, int64_t* tat, uint64_t* expiry_sat, int* occupied) {
    const dn_cell* c = &d->cells[slot];
    *tat = c->val;
    *expiry_sat = c->expiry > (u128)UINT64_MAX ? UINT64_MAX : (uint64_t)c->expiry;
    *occupied = c->occupied;
}
/* the same for a range of slots (bulk comparison of the whole resident state at full size) */
void tco_dense_dump(const tco_dense* d, size_t first, size_t n, int64_t* tat, uint64_t* expiry_sat, uint8_t* occupied) {
    for (size_t i = 0; i < n && first + i < d->capacity; i++) {
        const dn_cell* c = &d->cells[first + i];
        tat[i] = c->val;
        expiry_sat[i] = c->expiry > (u128)UINT64_MAX ? UINT64_MAX : (uint64_t)c->expiry;
        occupied[i] = c->occupied;
    }
}
uint64_t tco_dense_sweep(tco_dense* d, int64_t now) {
    uint64_t removed = 0;
    for (size_t i = 0; i < d->capacity; i++) {
        dn_cell* c = &d->cells[i];
        if (c->occupied && !(c->expiry > (u128)now)) { /* retain(exp > now) */
            c->occupied = 0;
            c->val = 0;
            c->expiry = 0;
            removed++;
        }
    }
    return removed;
}
size_t tco_dense_live(const tco_dense* d) {
    size_t n = 0;
    for (size_t i = 0; i < d->capacity; i++) n += d->cells[i].occupied;
    return n;
}

/* ---- Batch drivers -------------------------------------------------------- */
static inline void io_store(const tco_batch_io* io, size_t i, const tco_result* r) {
    if (io->allowed) io->allowed[i] = r->allowed;
    if (io->limit) io->limit[i] = r->limit;
    if (io->remaining) io->remaining[i] = r->remaining;
    if (io->reset_after_ns) io->reset_after_ns[i] = (int64_t)r->reset_after_ns;
    if (io->retry_after_ns) io->retry_after_ns[i] = (int64_t)r->retry_after_ns;
    if (io->status) io->status[i] = r->status;
}
#define IO_ARGS(io, i)                                                                      \
    (io)->max_burst[(i) * (io)->burst_stride], (io)->count_per_period[(i) * (io)->count_stride], \
        (io)->period[(i) * (io)->period_stride], (io)->quantity[(i) * (io)->quantity_stride],   \
        (io)->now_ns[(i) * (io)->now_stride]

void tco_batch_keys(tco_store* st, const uint8_t* key_bytes, const uint32_t* key_off,
                    const tco_batch_io* io) {
    for (size_t i = 0; i < io->n; i++) {
        tco_result r;
        tco_rate_limit(st, key_bytes + key_off[i], key_off[i + 1] - key_off[i], IO_ARGS(io, i), &r);
        io_store(io, i, &r);
    }
}
void tco_batch_slots(tco_store* st, const uint32_t* slot, const tco_batch_io* io) {
    for (size_t i = 0; i < io->n; i++) {
        tco_result r;
        uint8_t k[4];
        memcpy(k, &slot[i], 4);
        tco_rate_limit(st, k, 4, IO_ARGS(io, i), &r);
        io_store(io, i, &r);
    }
}

/* The same over `threads` threads (a CHECKER for full-size GPU tests, not a baseline): thread t serves the requests
 * whose slot is congruent to t, in index order.  Requests of different keys never touch the same cell, so every
 * key still sees its requests one by one in queue order and the results equal tco_batch_slots'. */
typedef struct {
    int tid, threads;
    tco_store* st;
    const uint32_t* slot;
    const tco_batch_io* io;
} ds_arg;
static void* ds_worker(void* p) {
    ds_arg* a = (ds_arg*)p;
    const tco_batch_io* io = a->io;
    for (size_t i = 0; i < io->n; i++) {
        if ((int)(a->slot[i] % (uint32_t)a->threads) != a->tid) continue;
        tco_result r;
        uint8_t k[4];
        memcpy(k, &a->slot[i], 4);
        tco_rate_limit(a->st, k, 4, IO_ARGS(io, i), &r);
        io_store(io, i, &r);
    }
    return NULL;
}
void tco_batch_slots_mt(tco_store* st, const uint32_t* slot, const tco_batch_io* io, int threads) {
    if (threads <= 1) {
        tco_batch_slots(st, slot, io);
        return;
    }
    if (threads > 256) threads = 256;
    pthread_t th[256];
    ds_arg args[256];
    for (int t = 0; t < threads; t++) {
        args[t].tid = t; args[t].threads = threads; args[t].st = st; args[t].slot = slot; args[t].io = io;
        pthread_create(&th[t], NULL, ds_worker, &args[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ---- hash-sharded multi-thread baseline -----------------------------------
 * What the reference's README recommends beyond one core ("client-side sharding by key", README.md:247-249):
 * one AdaptiveStore per thread, keys routed by hash.  Three parallel phases, all inside the timing:
 *   1. every thread routes a contiguous chunk of the stream: owner of each key + how many it sends to every shard
 *   2. every thread writes the indices of its chunk's requests into the shards' lists (offsets from phase 1:
 *      a shard's list is in stream order)
 *   3. every thread serves its own list from its own store
 * (A worker that walks the WHOLE stream looking for its own requests spends most of its time skipping:
 * 64 threads gave 3.7x that way.) */
typedef struct {
    int tid, threads;
    size_t cap;
    uint64_t max_operations;
    int64_t created;
    const uint8_t* key_bytes;
    const uint32_t* key_off;
    const tco_batch_io* io;
    uint16_t* owner;
    size_t* counts;  /* [threads (router)][threads (shard)] -> start of the router's part of the shard's list */
    uint32_t* lists; /* [n] request indices, grouped by shard */
    size_t* shard_begin; /* [threads + 1] */
} mt_arg;

static void* mt_route(void* p) {
    mt_arg* a = (mt_arg*)p;
    const size_t n = a->io->n, T = (size_t)a->threads;
    const size_t lo = n * (size_t)a->tid / T, hi = n * (size_t)(a->tid + 1) / T;
    size_t* mine = a->counts + (size_t)a->tid * T;
    for (size_t i = lo; i < hi; i++) {
        const uint8_t* k = a->key_bytes + a->key_off[i];
        size_t kl = a->key_off[i + 1] - a->key_off[i];
        /* shard by a hash decorrelated from the in-store placement hash */
        const uint16_t o = (uint16_t)(mix64(tco_hash_bytes(k, kl) ^ 0xa5a5a5a5a5a5a5a5ull) % (uint64_t)T);
        a->owner[i] = o;
        mine[o]++;
    }
    return NULL;
}
static void* mt_fill(void* p) {
    mt_arg* a = (mt_arg*)p;
    const size_t n = a->io->n, T = (size_t)a->threads;
    const size_t lo = n * (size_t)a->tid / T, hi = n * (size_t)(a->tid + 1) / T;
    size_t* at = a->counts + (size_t)a->tid * T; /* (now offsets) */
    for (size_t i = lo; i < hi; i++) a->lists[at[a->owner[i]]++] = (uint32_t)i;
    return NULL;
}
static void* mt_worker(void* p) {
    mt_arg* a = (mt_arg*)p;
    /* the server's store: cleanup interval 1 s .. 300 s, max_operations as configured (config.rs:301 default 1e6) */
    tco_adaptive* s = tco_adaptive_new(a->cap, 1000000000ull, 300ull * 1000000000ull, a->max_operations, a->created);
    tco_store st = tco_adaptive_as_store(s);
    const tco_batch_io* io = a->io;
    for (size_t j = a->shard_begin[a->tid]; j < a->shard_begin[a->tid + 1]; j++) {
        const size_t i = a->lists[j];
        const uint8_t* k = a->key_bytes + a->key_off[i];
        size_t kl = a->key_off[i + 1] - a->key_off[i];
        tco_result r;
        tco_rate_limit(&st, k, kl, IO_ARGS(io, i), &r);
        io_store(io, i, &r);
    }
    tco_adaptive_free(s);
    return NULL;
}

double tco_batch_keys_mt(int threads, size_t capacity_per_thread, uint64_t max_operations, int64_t created_ns,
                         const uint8_t* key_bytes, const uint32_t* key_off,
                         const tco_batch_io* io) {
    if (threads < 1) threads = 1;
    if (threads > 60000) threads = 60000;
    const size_t T = (size_t)threads, n = io->n;
    pthread_t* th = (pthread_t*)calloc(T, sizeof *th);
    mt_arg* args = (mt_arg*)calloc(T, sizeof *args);
    uint16_t* owner = (uint16_t*)malloc((n ? n : 1) * sizeof *owner);
    uint32_t* lists = (uint32_t*)malloc((n ? n : 1) * sizeof *lists);
    size_t* counts = (size_t*)calloc(T * T, sizeof *counts);
    size_t* shard_begin = (size_t*)calloc(T + 1, sizeof *shard_begin);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t t = 0; t < T; t++) {
        args[t].tid = (int)t; args[t].threads = threads; args[t].cap = capacity_per_thread;
        args[t].max_operations = max_operations; args[t].created = created_ns; args[t].key_bytes = key_bytes; args[t].key_off = key_off;
        args[t].io = io; args[t].owner = owner; args[t].counts = counts; args[t].lists = lists;
        args[t].shard_begin = shard_begin;
        pthread_create(&th[t], NULL, mt_route, &args[t]);
    }
    for (size_t t = 0; t < T; t++) pthread_join(th[t], NULL);
    /* counts[router][shard] -> where the router's requests start in the shard's list (T*T words: negligible) */
    size_t run = 0;
    for (size_t o = 0; o < T; o++) {
        shard_begin[o] = run;
        for (size_t r = 0; r < T; r++) {
            const size_t c = counts[r * T + o];
            counts[r * T + o] = run;
            run += c;
        }
    }
    shard_begin[T] = run;
    for (size_t t = 0; t < T; t++) pthread_create(&th[t], NULL, mt_fill, &args[t]);
    for (size_t t = 0; t < T; t++) pthread_join(th[t], NULL);
    for (size_t t = 0; t < T; t++) pthread_create(&th[t], NULL, mt_worker, &args[t]);
    for (size_t t = 0; t < T; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(args);
    free(owner);
    free(lists);
    free(counts);
    free(shard_begin);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* The reference's own library benchmark, shape for shape (throttlecrab-server/examples/store_comparison.rs:4-34,
 * the run behind "AdaptiveStore 12.5M req/s" in docs/benchmark-results.md:26-30): `iterations` calls of
 * rate_limit(&format!("key_{}", i % num_keys), 100, 1000, 3600, 1, SystemTime::now()) on
 * RateLimiter::new(AdaptiveStore::with_capacity(num_keys)), one thread, key formatting and the clock read
 * inside the loop.  Returns seconds; *allowed / *blocked as the example counts them. */
double tco_reference_shape(size_t num_keys, size_t iterations, uint64_t* allowed, uint64_t* blocked) {
    struct timespec t0, t1, now;
    clock_gettime(CLOCK_REALTIME, &now);
    tco_adaptive* s = tco_adaptive_with_capacity(num_keys, (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec);
    tco_store st = tco_adaptive_as_store(s);
    uint64_t na = 0, nb = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t i = 0; i < iterations; i++) {
        char key[32];
        const int kl = snprintf(key, sizeof key, "key_%zu", i % num_keys);
        clock_gettime(CLOCK_REALTIME, &now);
        tco_result r;
        tco_rate_limit(&st, (const uint8_t*)key, (size_t)kl, 100, 1000, 3600, 1, (int64_t)now.tv_sec * 1000000000ll + now.tv_nsec, &r);
        if (r.allowed) na++;
        else nb++;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    tco_adaptive_free(s);
    if (allowed) *allowed = na;
    if (blocked) *blocked = nb;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* `format!("{prefix}{id}")` key arena for a slot-id stream (bench helper). */
size_t tco_format_keys(const char* prefix, const uint32_t* ids, size_t n, uint8_t* out_bytes,
                       size_t out_cap, uint32_t* out_off) {
    size_t plen = strlen(prefix), pos = 0;
    for (size_t i = 0; i < n; i++) {
        char tmp[16];
        int len = 0;
        uint32_t v = ids[i];
        do { tmp[len++] = (char)('0' + v % 10); v /= 10; } while (v);
        if (pos + plen + (size_t)len > out_cap) return 0;
        out_off[i] = (uint32_t)pos;
        memcpy(out_bytes + pos, prefix, plen);
        pos += plen;
        while (len) out_bytes[pos++] = (uint8_t)tmp[--len];
    }
    out_off[n] = (uint32_t)pos;
    return pos;
}
