// key_table.hpp -- on-device key -> slot resolution for string keys (gfx950).
//
// Replaces what AHashMap<String, ...> does for the reference's stores
// (throttlecrab/src/core/store/adaptive_cleanup.rs:39-41, get/insert at
// :231-277): find the entry of a key, or bind a fresh one.  The hash only
// PLACES keys; hits are confirmed by a full key comparison, so results never
// depend on the hash function (same as the reference with ahash).
//
// Layout (all in HBM):
//   ktab[NB]        open-addressing table of 32-byte Entry records, linear probing, NB = 2^k >= 2*capacity
//                   w    = val(32) | tag(24) << 32 | len8 << 56;  val: 0 empty, 1 tombstone, >= 2 bound to
//                          slot val-2, bit 31 set = "being inserted by request (val & 0x7fffffff) of the
//                          current batch"; len8 = key length if <= 16, else 255
//                   hash = full hash of the bound key
//                   key  = the key itself, zero padded, when it is at most 16 bytes long
//                   A key of up to 16 bytes ("user:123", "key_1234567") is therefore found and CONFIRMED
//                   with one 32-byte access; longer keys are confirmed against their slot's KeyRec.
//   rec[cap]        one 128-byte KeyRec per slot (ONE memory line): full hash, key length, position of the slot's
//                   ktab entry at binding time and the key bytes inline (<= 112 B; a longer key keeps an 8-byte
//                   offset into the overflow arena).  A confirm requests the length word and the key words of
//                   the line together, so a hit on a 17..112-byte key costs the entry line + this line.
//   bound[cap]      u8 1 = slot has a key (the compact column the expiry sweep scans:
//                   walking the 128-byte records would read 128x the bytes)
//   free_slots[cap] stack of unbound slots, free_top = number of free slots
//   pos_col[cap]    the ktab position of each bound slot's entry (KeyRec::pos as a column of its own: the sweep turns the
//                   entries of ~1 M freed slots into tombstones, and a 4-byte column read in slot order costs a sixteenth of
//                   one record line per key)
//
// Inserting inside a batch is a three-kernel protocol (k_probe, k_bind, k_follow) without spinning (a wave
// cannot wait for its own lanes): k_probe claims an empty entry with the
// REQUEST INDEX, duplicates of the same new key find that claim and compare
// against the claimant's key bytes in the input arena; k_bind (claimants)
// takes a slot, stores the key and publishes the slot; k_follow copies the
// claimant's slot to its duplicates.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kt {

constexpr int THREADS = 256;
constexpr uint32_t VAL_EMPTY = 0u, VAL_TOMB = 1u, VAL_PENDING = 0x80000000u;
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
constexpr uint32_t ST_FOUND = 0u, ST_CLAIMANT = 1u, ST_FOLLOWER = 2u, ST_MISSING = 3u, ST_NOSPACE = 4u; // NOSPACE: claimed an entry, but the overflow arena is full
constexpr uint32_t INLINE_KEY = 112u; // KeyRec: one 128-byte line holds hash, length, entry position and the key
constexpr uint32_t ENTRY_KEY = 16u;   // Entry
constexpr uint32_t LEN8_LONG = 255u;
// The tombstone count only feeds the rebuild decision, but every probe block and every sweep block moves it: one word
// took 16 384 atomics of a 1 Mi-key batch once the table held tombstones to recycle (~12 ns each, serialised: k_probe 60
// -> 390-450 us, and every kernel beside it slowed down with it).  64 shards, one atomic per BLOCK.
constexpr uint32_t TOMB_SHARDS = 64u;

struct __attribute__((aligned(32))) Entry {
    unsigned long long w;
    uint64_t hash;
    uint64_t key[2];
};

struct __attribute__((aligned(128))) KeyRec {
    uint64_t hash;
    uint32_t len; // NO_SLOT = slot not bound
    uint32_t pos; // the entry's position WHEN THE KEY WAS BOUND (a rebuild moves entries: Table::pos_col[] is the one that is kept current)
    uint8_t bytes[INLINE_KEY];
};

// Denial counts outlive a key's slot (the reference counts by KEY, independent of what its store sweeps:
// throttlecrab-server/src/metrics.rs:24-76).  When a sweep unbinds a key whose slot counted denials, key and count move
// into this small open-addressed side table; when the key is bound again, the count moves back to its new slot's
// counter -- a key's denials are always in exactly one place.  RETIRED_CAP entries (3 x the reference's 10 000-key
// limit, metrics.rs:17: the host trims to the top 10 000 once the table passes 30 000, like TopDeniedKeys::cleanup);
// keys over RETIRED_KEY bytes are not kept (the reference ignores keys over 256 bytes, metrics.rs:11,37).
// (round 4: 65 536 entries, so that 30 000 keys are a load of 46 %; the record BEHIND the table -- index RETIRED_CAP -- carries the
// table's statistics: .count = denial counts that found no room within RETIRED_PROBES records, .len = records ever claimed,
// tombstones included; the host compacts the table -- tombstones out, trimmed like the reference's map -- when either says so)
constexpr uint32_t RETIRED_CAP = 65536u, RETIRED_KEY = 256u, RETIRED_PROBES = 128u;
constexpr unsigned long long RT_EMPTY = 0ull, RT_BUSY = 1ull, RT_DELETED = 2ull, RT_VALID = 1ull << 63;
struct RetiredRec {
    unsigned long long tag; // RT_EMPTY / RT_BUSY / RT_DELETED / hash | RT_VALID
    uint32_t count;
    uint32_t len;
    uint8_t bytes[RETIRED_KEY];
};

struct Table {
    Entry* ktab;
    uint64_t nb_mask;
    KeyRec* rec;
    uint8_t* bound;
    uint8_t* overflow;                 // two halves of overflow_bytes each: keys live in half *overflow_half,
    uint64_t overflow_bytes;           // the sweep compacts them into the other one when the arena runs full
    unsigned long long* overflow_used; // bytes handed out in the current half
    uint32_t* overflow_half;           // 0 / 1
    uint32_t* free_slots;
    uint32_t* pos_col;      // [capacity] KeyRec::pos again, as a compact column: what the sweep reads to find the entries of the slots it freed
    int* free_top;
    uint32_t* tombs;        // [TOMB_SHARDS] tombstones currently in ktab, as shards that sum to the count (wrap-around arithmetic)
    uint32_t* error_flag;   // != 0: a key could not be bound (no slot / no overflow space)
    uint32_t capacity;
    RetiredRec* retired;    // TC_CFG_TRACK_DENIED: denial counts of keys that lost their slot (else nullptr)
    uint32_t* denied;       // ... and the per-slot denial counters
};

// upper half of Entry::w for a key: 24 tag bits of the hash + len8
__device__ __forceinline__ unsigned long long entry_meta(uint64_t h, uint32_t len) {
    const unsigned long long tag = (h >> 40) & 0xFFFFFFull;
    const unsigned long long len8 = len <= ENTRY_KEY ? len : LEN8_LONG;
    return (tag << 32) | (len8 << 56);
}

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// 1..7 trailing bytes as a little-endian word, without reading past the key
// (gfx950 runs with unaligned global access, so the 2/4/8-byte loads are single instructions)
__host__ __device__ __forceinline__ uint64_t load_tail(const uint8_t* __restrict__ p, uint32_t r) {
    uint64_t w = 0;
    uint32_t sh = 0;
    if (r & 4u) {
        uint32_t v;
        __builtin_memcpy(&v, p, 4);
        w = v;
        p += 4;
        sh = 32;
    }
    if (r & 2u) {
        uint16_t v;
        __builtin_memcpy(&v, p, 2);
        w |= (uint64_t)v << sh;
        p += 2;
        sh += 16;
    }
    if (r & 1u) w |= (uint64_t)p[0] << sh;
    return w;
}

__host__ __device__ __forceinline__ uint64_t hash_key(const uint8_t* __restrict__ p, uint32_t len) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)len;
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t w;
        __builtin_memcpy(&w, p + i, 8);
        h = mix64(h ^ w) + 0x9e3779b97f4a7c15ull;
    }
    if (i < len) h = mix64(h ^ load_tail(p + i, len - i) ^ ((uint64_t)(len - i) << 56));
    return mix64(h);
}

__device__ __forceinline__ const uint8_t* stored_key(const Table& t, uint32_t slot, uint32_t len) {
    const uint8_t* in = t.rec[slot].bytes;
    if (len <= INLINE_KEY) return in;
    uint64_t off;
    __builtin_memcpy(&off, in, 8);
    return t.overflow + (uint64_t)*t.overflow_half * t.overflow_bytes + off;
}

__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t len) {
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8);
        __builtin_memcpy(&y, b + i, 8);
        if (x != y) return false;
    }
    return i == len || load_tail(a + i, len - i) == load_tail(b + i, len - i);
}

// a key of at most 16 bytes as two zero-padded little-endian words (what Entry::key holds)
__host__ __device__ __forceinline__ void short_key_words(const uint8_t* __restrict__ p, uint32_t len, uint64_t& k0, uint64_t& k1) {
    k0 = k1 = 0;
    if (len >= 8) {
        __builtin_memcpy(&k0, p, 8);
        if (len == 16) __builtin_memcpy(&k1, p + 8, 8);
        else if (len > 8) k1 = load_tail(p + 8, len - 8);
    } else if (len) {
        k0 = load_tail(p, len);
    }
}

// hash_key() of a key of at most 16 bytes, from its two zero-padded words (no further loads)
__host__ __device__ __forceinline__ uint64_t hash_short(uint64_t k0, uint64_t k1, uint32_t len) {
    const uint64_t C = 0x9e3779b97f4a7c15ull;
    uint64_t h = C ^ (uint64_t)len;
    if (len >= 8) {
        h = mix64(h ^ k0) + C;
        if (len == 16) h = mix64(h ^ k1) + C;
        else if (len > 8) h = mix64(h ^ k1 ^ ((uint64_t)(len - 8) << 56));
    } else if (len) {
        h = mix64(h ^ k0 ^ ((uint64_t)len << 56));
    }
    return mix64(h);
}

// the low `nbytes` (0..8) bytes of w
__host__ __device__ __forceinline__ uint64_t keep_bytes(uint64_t w, uint32_t nbytes) {
    return nbytes >= 8 ? w : (nbytes ? (w & ((1ull << (8 * nbytes)) - 1ull)) : 0ull);
}

// ---- keys of 17..64 bytes as eight zero-padded words held in registers ------------------------------------
// hash_key() and bytes_equal() walk a key in 8-byte steps with one load per step whose value the next step needs
// before it can go on: for a 48-byte key that is six memory round trips to hash it and six more for every
// comparison (measured: 860 us per 1 Mi keys of 32..64 bytes against 69 us for keys of up to 16).  Here every
// load of a key is issued before any value is used.
constexpr int KEY_WORDS = 8;
// `safe_end`: bytes readable from p (>= len): a word that starts inside the key but ends beyond it is only
// read whole if it still ends inside readable memory, else byte by byte (the last key of an arena).
__device__ __forceinline__ void load_words(const uint8_t* __restrict__ p, uint32_t len, uint64_t safe_end, uint64_t (&w)[KEY_WORDS]) {
#pragma unroll
    for (int j = 0; j < KEY_WORDS; ++j) {
        const uint32_t at = (uint32_t)j * 8u;
        uint64_t v = 0;
        if (at < len) {
            if ((uint64_t)at + 8u <= safe_end) __builtin_memcpy(&v, p + at, 8);
            else v = load_tail(p + at, len - at); // (< 8 bytes left in readable memory)
        }
        w[j] = v;
    }
#pragma unroll
    for (int j = 0; j < KEY_WORDS; ++j) {
        const uint32_t at = (uint32_t)j * 8u;
        if (at < len && len - at < 8u) w[j] = keep_bytes(w[j], len - at);
    }
}
// hash_key() of a key of at most 64 bytes from its words (the same value, bit for bit)
__device__ __forceinline__ uint64_t hash_words(const uint64_t (&w)[KEY_WORDS], uint32_t len) {
    const uint64_t C = 0x9e3779b97f4a7c15ull;
    uint64_t h = C ^ (uint64_t)len;
#pragma unroll
    for (int j = 0; j < KEY_WORDS; ++j) {
        const uint32_t at = (uint32_t)j * 8u;
        if (at + 8u <= len) h = mix64(h ^ w[j]) + C;
        else if (at < len) h = mix64(h ^ w[j] ^ ((uint64_t)(len - at) << 56));
    }
    return mix64(h);
}
__device__ __forceinline__ bool words_equal(const uint64_t (&a)[KEY_WORDS], const uint64_t (&b)[KEY_WORDS]) {
    uint64_t d = 0;
#pragma unroll
    for (int j = 0; j < KEY_WORDS; ++j) d |= a[j] ^ b[j];
    return d == 0;
}
// key `len` bytes at `other` (readable up to other + other_safe) == the key whose words are `mine` (same length)
__device__ __forceinline__ bool key_equals_words(const uint8_t* __restrict__ other, uint64_t other_safe, uint32_t len, const uint8_t* __restrict__ key,
                                                 const uint64_t (&mine)[KEY_WORDS]) {
    if (len > 8u * KEY_WORDS) return bytes_equal(other, key, len);
    uint64_t o[KEY_WORDS];
    load_words(other, len, other_safe, o);
    return words_equal(o, mine);
}

// ---- denial counts of keys without a slot (RetiredRec) ---------------------------------------------------------
// sweep side: the key of slot `s` (hash h, len bytes at `key`) leaves with `count` denials
__device__ inline void retire_denials(const Table& t, uint64_t h, const uint8_t* key, uint32_t len, uint32_t count) {
    if (t.retired == nullptr || count == 0u || len > RETIRED_KEY) return;
    const unsigned long long mine = h | RT_VALID;
    uint32_t pos = (uint32_t)(h >> 17) & (RETIRED_CAP - 1u);
    for (uint32_t probes = 0; probes < RETIRED_PROBES; ++probes, pos = (pos + 1u) & (RETIRED_CAP - 1u)) {
        RetiredRec* r = &t.retired[pos];
        unsigned long long tag = __hip_atomic_load(&r->tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; tag == RT_BUSY && spin < 100000; ++spin) { // somebody is writing this record: a few hundred cycles
            __builtin_amdgcn_s_sleep(2);
            tag = __hip_atomic_load(&r->tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tag == mine && r->len == len && bytes_equal(r->bytes, key, len)) {
            atomicAdd(&r->count, count);
            return;
        }
        if (tag == RT_EMPTY || tag == RT_DELETED) {
            unsigned long long expected = tag;
            if (__hip_atomic_compare_exchange_strong(&r->tag, &expected, RT_BUSY, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (tag == RT_EMPTY) atomicAdd(&t.retired[RETIRED_CAP].len, 1u);
                r->count = count;
                r->len = len;
                for (uint32_t b = 0; b < len; ++b) r->bytes[b] = key[b];
                __hip_atomic_store(&r->tag, mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            --probes; // lost the record to another key: look at it again
            pos = (pos + RETIRED_CAP - 1u) & (RETIRED_CAP - 1u);
        }
    }
    // no room within RETIRED_PROBES records: the count is dropped (the reference's capped map forgets keys too) -- and counted, so
    // that the host compacts the table before more are
    atomicAdd(&t.retired[RETIRED_CAP].count, 1u);
}
// bind side: a key that gets a slot again takes its retired denials along
__device__ inline void resurrect_denials(const Table& t, uint64_t h, const uint8_t* key, uint32_t len, uint32_t slot) {
    if (t.retired == nullptr || len > RETIRED_KEY) return;
    const unsigned long long mine = h | RT_VALID;
    uint32_t pos = (uint32_t)(h >> 17) & (RETIRED_CAP - 1u);
    for (uint32_t probes = 0; probes < RETIRED_PROBES; ++probes, pos = (pos + 1u) & (RETIRED_CAP - 1u)) {
        RetiredRec* r = &t.retired[pos];
        const unsigned long long tag = __hip_atomic_load(&r->tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (tag == RT_EMPTY) return;
        if (tag == mine && r->len == len && bytes_equal(r->bytes, key, len)) {
            const uint32_t c = atomicExch(&r->count, 0u);
            if (c) atomicAdd(&t.denied[slot], c);
            __hip_atomic_store(&r->tag, RT_DELETED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

// A request's key as (hash, k0, k1).  Keys of at most 16 bytes that do not end within 16 bytes of the
// arena's end are read with two unconditional 8-byte loads and masked -- one round trip instead of the
// branchy 8/4/2/1-byte tail loads -- and hashed from the words.
__device__ __forceinline__ uint64_t load_and_hash(const uint8_t* __restrict__ key_bytes, uint32_t off, uint32_t len, uint32_t arena,
                                                  uint64_t& k0, uint64_t& k1) {
    const uint8_t* key = key_bytes + off;
    k0 = k1 = 0;
    if (len > ENTRY_KEY) return hash_key(key, len);
    if ((uint64_t)off + 16u <= arena) {
        uint64_t a, b;
        __builtin_memcpy(&a, key, 8);
        __builtin_memcpy(&b, key + 8, 8);
        k0 = keep_bytes(a, len);
        k1 = len > 8 ? keep_bytes(b, len - 8) : 0ull;
    } else {
        short_key_words(key, len, k0, k1);
    }
    return hash_short(k0, k1, len);
}

// the hash of request i's key, exactly as probe_request computes it
__device__ __forceinline__ uint64_t request_hash(const uint8_t* __restrict__ key_bytes, const uint32_t* __restrict__ key_off, uint32_t n, uint32_t i) {
    const uint32_t off = key_off[i], len = key_off[i + 1] - off, arena = key_off[n];
    if (len > ENTRY_KEY && len <= 8u * KEY_WORDS) {
        uint64_t kw[KEY_WORDS];
        load_words(key_bytes + off, len, (uint64_t)arena - off, kw);
        return hash_words(kw, len);
    }
    uint64_t k0, k1;
    return load_and_hash(key_bytes, off, len, arena, k0, k1);
}

// ---------------------------------------------------------------------------
// probe: one lane per request.  INSERT: unseen keys claim an entry.
// Request i of a batch of n: returns its state; `slot` (found: the slot; claimant of a long key: the
// reserved overflow offset / 16), `ax` (claimant: ktab position; follower: request index of the claimant)
// and `h` (the key's hash).
// ---------------------------------------------------------------------------
template <bool INSERT>
__device__ __forceinline__ uint32_t probe_request(const Table& t, const uint8_t* __restrict__ key_bytes,
                                                  const uint32_t* __restrict__ key_off, uint32_t n, uint32_t i, uint32_t& slot,
                                                  uint32_t& ax, uint64_t& h, bool& recycled) {
    uint32_t st = ST_MISSING;
    recycled = false; // the claim took a tombstone: the caller takes it off t.tombs (tombs_sub: one atomic per wave)
    const uint32_t off = key_off[i], len = key_off[i + 1] - off, arena = key_off[n];
    const uint8_t* key = key_bytes + off;
    uint64_t k0 = 0, k1 = 0;
    uint64_t kw[KEY_WORDS];
    const bool in_words = len > ENTRY_KEY && len <= 8u * KEY_WORDS; // 17..64 bytes: the key's words stay in registers
    if (in_words) {
        load_words(key, len, (uint64_t)arena - off, kw);
        h = hash_words(kw, len);
    } else {
#pragma unroll
        for (int j = 0; j < KEY_WORDS; ++j) kw[j] = 0;
        h = load_and_hash(key_bytes, off, len, arena, k0, k1);
    }
    const unsigned long long meta = entry_meta(h, len);
    slot = NO_SLOT;
    ax = 0;
    // An unseen key claims the FIRST TOMBSTONE of its probe chain if there is one, else the empty entry that
    // ends the chain (tombstones are recycled by inserts: a table that once ran full does not keep its probe
    // chains long until the next rebuild).  Every request of a key walks the same chain, so all of them aim at
    // the same entry; whoever loses the compare-and-swap walks again and finds the winner's claim.
    for (uint32_t attempt = 0; attempt <= n && st == ST_MISSING; ++attempt) { // (a lost compare-and-swap is somebody else's progress: <= n turns)
        uint64_t pos = h & t.nb_mask;
        uint64_t tomb_pos = ~0ull;
        unsigned long long tomb_word = 0ull;
        bool absent = false;
        for (uint64_t probes = 0; probes <= t.nb_mask; ++probes) {
            Entry* en = &t.ktab[pos];
            // The whole 32-byte entry in one round trip, with plain loads: what they can show is either final
            // for this kernel (bound entries only change in other kernels; a pending entry keeps its claimant
            // until it is bound), a tombstone (which may turn into a claim: settled by the compare-and-swap
            // below), or "empty", which ends the chain.
            const ulonglong2 lo = *reinterpret_cast<const ulonglong2*>(en);      // w, hash
            const ulonglong2 hi = *(reinterpret_cast<const ulonglong2*>(en) + 1); // key[0], key[1]
            const unsigned long long e = lo.x;
            if (e == 0ull) {
                absent = true;
                break;
            }
            const uint32_t val = (uint32_t)e;
            const bool meta_eq = (e & 0xFFFFFFFF00000000ull) == meta;
            if (val == VAL_TOMB) {
                if (INSERT && tomb_pos == ~0ull) {
                    tomb_pos = pos;
                    tomb_word = e;
                }
            } else if (val & VAL_PENDING) {
                if (meta_eq) {
                    const uint32_t j = val & ~VAL_PENDING; // request index of the claimant (this batch)
                    const uint32_t joff = key_off[j], jlen = key_off[j + 1] - joff;
                    if (jlen == len && (in_words ? key_equals_words(key_bytes + joff, (uint64_t)arena - joff, len, key, kw)
                                                 : bytes_equal(key_bytes + joff, key, len))) {
                        st = ST_FOLLOWER;
                        ax = j;
                        break;
                    }
                }
            } else if (meta_eq && lo.y == h) {
                // bound in an earlier batch: hash / key / record are stable
                const uint32_t s = val - 2u;
                bool same;
                if (len <= ENTRY_KEY) same = hi.x == k0 && hi.y == k1;
                else if (in_words) {
                    // 17..64 bytes: the key sits inside its slot's record (one 128-byte line: length and key words are
                    // all requested before any of them is looked at -- a second dependent round trip costs as much as
                    // the first)
                    const KeyRec& kr = t.rec[s];
                    const uint32_t rlen = kr.len;
                    uint64_t o[KEY_WORDS];
                    load_words(kr.bytes, len, INLINE_KEY, o);
                    same = (rlen == len) & words_equal(o, kw);
                }
                else same = t.rec[s].len == len && bytes_equal(stored_key(t, s, len), key, len);
                if (same) {
                    st = ST_FOUND;
                    slot = s;
                    break;
                }
            }
            pos = (pos + 1) & t.nb_mask;
        }
        if (st != ST_MISSING || !INSERT) break;
        if (!absent && tomb_pos == ~0ull) break; // the table has neither an empty entry nor a tombstone left on this chain
        // No free slot when the batch began: do not claim (a claim that cannot be bound ends as a tombstone, and a
        // flood of unseen keys against a full table would turn every empty entry into one).  free_top only moves
        // after the probe, so a batch can still over-claim by its own new keys: bounded, and recycled later.
        if (*t.free_top <= 0) break;
        const unsigned long long mine = meta | (unsigned long long)(VAL_PENDING | i);
        const uint64_t target = tomb_pos != ~0ull ? tomb_pos : pos;
        unsigned long long expected = tomb_pos != ~0ull ? tomb_word : 0ull;
        if (__hip_atomic_compare_exchange_strong(&t.ktab[target].w, &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            st = ST_CLAIMANT;
            ax = (uint32_t)target;
            recycled = tomb_pos != ~0ull;
        }
        // else: somebody claimed that entry in the meantime -- walk again (it now shows as a pending claim)
    }
    if (INSERT && st == ST_MISSING) atomicExch(t.error_flag, 1u); // no room on this key's chain: TC_E_TABLE_FULL
    return st;
}

// tombstones recycled by this BLOCK's claims come off the table's count with one atomic on the block's shard
// (reached by every thread of the block; one barrier)
template <int NT>
__device__ __forceinline__ void tombs_sub(const Table& t, bool recycled) {
    __shared__ uint32_t s_t[NT / 64];
    const unsigned long long m = __ballot(recycled);
    if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 0; w < NT / 64; ++w) c += s_t[w];
        if (c) atomicSub(&t.tombs[blockIdx.x % TOMB_SHARDS], c);
    }
}

// Claimants of keys longer than INLINE_KEY reserve their overflow bytes before binding, so that binding knows who
// takes a slot (offset / 16 rides in `slot`).  ONE atomic per block: the block's claimants line their (16-byte
// rounded) sizes up with a scan and share the block's reservation.  (One returning atomicAdd per claimant on the
// arena's cursor -- a single word -- was the whole cost of k_probe on long keys: ~100 k of them per 1 Mi-key batch
// of the configs[4] mix serialise at ~6 ns each, 0.6 of the kernel's 0.73 ms.)  Reached by every thread of the block.
template <int NT>
__device__ __forceinline__ void reserve_overflow(const Table& t, bool live, uint32_t len, uint32_t& st, uint32_t& slot) {
    __shared__ uint32_t s_w[NT / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool need = live && st == ST_CLAIMANT && len > INLINE_KEY;
    const uint32_t bytes = need ? (len + 15u) & ~15u : 0u;
    uint32_t v = bytes;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    if (lane == 63) s_w[wave] = v;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) before += s_w[w];
        total += s_w[w];
    }
    if (total == 0) return; // (block-uniform)
    if (threadIdx.x == 0) {
        const unsigned long long base = atomicAdd(t.overflow_used, (unsigned long long)total);
        s_base = base;
        // what reaches past the end of the arena is handed back at once: a cursor left beyond the end would refuse every
        // later long key, shorter ones included, until the next sweep compacts the arena
        if (base + total > t.overflow_bytes) {
            const unsigned long long keep = base < t.overflow_bytes ? t.overflow_bytes - base : 0ull;
            atomicAdd(t.overflow_used, ~((unsigned long long)total - keep) + 1ull); // -= the part beyond the end
        }
    }
    __syncthreads();
    if (need) {
        const unsigned long long ovf = s_base + before + (v - bytes);
        if (ovf + len > t.overflow_bytes) st = ST_NOSPACE;
        else slot = (uint32_t)(ovf >> 4);
    }
}

// outputs: slot_out[i], state[i], aux[i] as probe_request leaves them; claim_cnt[block]
template <bool INSERT>
__global__ __launch_bounds__(THREADS) void k_probe(Table t, const uint8_t* __restrict__ key_bytes,
                                                   const uint32_t* __restrict__ key_off, uint32_t n,
                                                   uint32_t* __restrict__ slot_out, uint8_t* __restrict__ state,
                                                   uint32_t* __restrict__ aux, uint32_t* __restrict__ claim_cnt) {
    const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    uint32_t st = ST_MISSING;
    uint32_t slot = NO_SLOT, ax = 0;
    uint64_t h = 0;
    bool recycled = false;
    if (i < n) {
        st = probe_request<INSERT>(t, key_bytes, key_off, n, i, slot, ax, h, recycled);
    }
    if (INSERT) tombs_sub<THREADS>(t, recycled);
    if (INSERT) reserve_overflow<THREADS>(t, i < n, i < n ? key_off[i + 1] - key_off[i] : 0u, st, slot);
    if (i < n) { // (round 4: the hash is not handed on -- a claimant hashes its key again in k_bind, from words it loads anyway -- and the
                 // state is a byte: 20 -> 9 MB of scratch columns per 1 Mi keys)
        slot_out[i] = slot;
        state[i] = (uint8_t)st;
        aux[i] = ax;
    }
    if (INSERT) {
        // claimants of this block, for k_bind's slot assignment (no atomics on the free stack)
        __shared__ uint32_t s_claims[THREADS / 64];
        const unsigned long long m = __ballot(i < n && st == ST_CLAIMANT);
        if ((threadIdx.x & 63) == 0) s_claims[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = 0;
            for (int w = 0; w < THREADS / 64; ++w) c += s_claims[w];
            claim_cnt[blockIdx.x] = c;
        }
    }
}

// rank of each flagged lane inside its block + the block total (one barrier)
template <int NT = THREADS>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t& total) {
    __shared__ uint32_t s_w[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) before += s_w[w];
        tot += s_w[w];
    }
    total = tot;
    return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Claimant i (its entry is ktab[pos], hash h, long keys: overflow offset / 16 in ovf16) takes
// free_slots[stack_idx] (stack_idx < 0: the stack ran dry), stores the key and publishes the binding.
// Returns the slot or NO_SLOT.
__device__ __forceinline__ uint32_t bind_claimant(const Table& t, const uint8_t* __restrict__ key_bytes,
                                                  const uint32_t* __restrict__ key_off, uint32_t n_keys, uint32_t i, uint32_t pos, uint64_t h,
                                                  uint32_t ovf16, int stack_idx) {
    const uint32_t off = key_off[i], len = key_off[i + 1] - off;
    const uint8_t* key = key_bytes + off;
    const uint32_t slot = stack_idx >= 0 ? t.free_slots[stack_idx] : NO_SLOT;
    if (slot != NO_SLOT) {
        KeyRec& kr = t.rec[slot];
        uint8_t* dst = kr.bytes;
        if (len > INLINE_KEY) {
            const uint64_t o64 = (uint64_t)ovf16 << 4; // reserved by the probe
            __builtin_memcpy(kr.bytes, &o64, 8);
            dst = t.overflow + (uint64_t)*t.overflow_half * t.overflow_bytes + o64; // reservations are 16-byte multiples: dst is 16-byte aligned
        }
        if (len <= 8u * KEY_WORDS) {
            // every load of the key first, then the stores (whole words: the destination -- 48 inline bytes, or an arena
            // reservation rounded up to 16 -- has room for the zero padding of the last one)
            uint64_t w[KEY_WORDS];
            load_words(key, len, (uint64_t)key_off[n_keys] - off, w);
#pragma unroll
            for (int j = 0; j < KEY_WORDS; ++j)
                if ((uint32_t)j * 8u < len) __builtin_memcpy(dst + j * 8, &w[j], 8);
        } else {
            uint32_t b = 0;
            for (; b + 8 <= len; b += 8) {
                uint64_t w;
                __builtin_memcpy(&w, key + b, 8);
                __builtin_memcpy(dst + b, &w, 8);
            }
            for (; b < len; ++b) dst[b] = key[b];
        }
        kr.hash = h;
        kr.len = len;
        kr.pos = pos;
        t.pos_col[slot] = pos;
        t.bound[slot] = 1;
        Entry* en = &t.ktab[pos];
        uint64_t k0 = 0, k1 = 0;
        if (len <= ENTRY_KEY) short_key_words(key, len, k0, k1);
        en->hash = h;
        en->key[0] = k0;
        en->key[1] = k1;
        en->w = entry_meta(h, len) | (unsigned long long)(slot + 2u);
        resurrect_denials(t, h, key, len, slot);
    } else { // the free stack ran dry
        t.ktab[pos].w = entry_meta(h, len) | VAL_TOMB;
        atomicAdd(&t.tombs[pos % TOMB_SHARDS], 1u);
        atomicExch(t.error_flag, 1u);
    }
    return slot;
}
// a claimant without room for its long key gives the claimed entry back as a tombstone
__device__ __forceinline__ void release_claim(const Table& t, const uint32_t* __restrict__ key_off, uint32_t i, uint32_t pos, uint64_t h) {
    const uint32_t len = key_off[i + 1] - key_off[i];
    t.ktab[pos].w = entry_meta(h, len) | VAL_TOMB;
    atomicAdd(&t.tombs[pos % TOMB_SHARDS], 1u);
    atomicExch(t.error_flag, 1u);
}

// claimants: take a slot, store the key, publish the binding.  Claimant number R of the batch (in request
// order) takes free_slots[top - 1 - R]: k_probe left the number of claimants per block in claim_cnt[], and a block here sums
// the counts of the blocks before it itself -- 256 lanes over at most a few thousand words that sit in L2 (round 4: the
// one-block k_claim_scan between the two kernels was 7-9 us and a launch on the key stage's chain).  The last block leaves
// the batch's total behind the counts; the stack pointer itself moves once, in k_follow.  No atomics: one on a single
// address costs ~12 ns and serialises (a per-wave pop was 197 us of a 225 us kernel, a per-1024-block pop still 12 us).
static __global__ __launch_bounds__(THREADS) void k_bind(Table t, const uint8_t* __restrict__ key_bytes,
                                                  const uint32_t* __restrict__ key_off, uint32_t n,
                                                  uint32_t* __restrict__ slot_out, const uint8_t* __restrict__ state,
                                                  const uint32_t* __restrict__ aux, uint32_t* __restrict__ claim_cnt) {
    __shared__ uint32_t s_part[THREADS / 64];
    const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    const uint32_t st = i < n ? state[i] : ST_FOUND;
    const bool want = st == ST_CLAIMANT;
    uint32_t total = 0;
    const uint32_t rank = block_rank<THREADS>(want, total);
    const bool last = blockIdx.x == gridDim.x - 1;
    uint32_t before = 0;
    if (total != 0u || last) { // (block-uniform)
        uint32_t part = 0;
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += THREADS) part += claim_cnt[j];
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
        __syncthreads();
        for (int w = 0; w < THREADS / 64; ++w) before += s_part[w];
        if (last && threadIdx.x == 0) claim_cnt[gridDim.x] = before + total; // (k_follow)
    }
    if (want) {
        const int top = *t.free_top; // moves in k_follow, not here
        slot_out[i] = bind_claimant(t, key_bytes, key_off, n, i, aux[i], request_hash(key_bytes, key_off, n, i), slot_out[i],
                                    top - 1 - (int)(before + rank));
    } else if (st == ST_NOSPACE) {
        release_claim(t, key_off, i, aux[i], request_hash(key_bytes, key_off, n, i));
        slot_out[i] = NO_SLOT;
    }
}

// duplicates of a key first seen in this batch take the claimant's slot; block 0 also moves the free
// stack's pointer past the slots k_bind handed out and counts the insertions
static __global__ __launch_bounds__(THREADS) void k_follow(uint32_t n, uint32_t* __restrict__ slot_out,
                                                    const uint8_t* __restrict__ state, const uint32_t* __restrict__ aux, Table t,
                                                    const uint32_t* __restrict__ claim_cnt, uint32_t n_blocks,
                                                    unsigned long long* inserted_counter) {
    const uint32_t i = blockIdx.x * THREADS + threadIdx.x;
    if (i < n && state[i] == ST_FOLLOWER) slot_out[i] = slot_out[aux[i]];
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            const uint32_t total = claim_cnt[n_blocks]; // (k_bind's last block)
            const int top = *t.free_top;
            const int got = top < 0 ? 0 : (top < (int)total ? top : (int)total);
            *t.free_top = top - got;
            if (got) atomicAdd(inserted_counter, (unsigned long long)got);
        }
    }
}

// Where the p-th of m items goes when the items (given in slot order) are to LEAVE in an order that visits the whole slot range
// again and again: the items are cut into rows of `cols` consecutive ones and handed out column by column.  A bijection of
// [0, m).  The free stack is popped from the top down, so that consecutive pops -- the new keys of one batch -- get slots from
// all over the key space and the batch's slots stay spread evenly enough for the range path (radix_sort.hpp: two grouping
// launches instead of a histogram and three LSD passes).  Round 4 pushed freed slots in slot order: neighbouring records for
// k_bind, but a fifth of every configs[4] batch in one narrow slot range, and with it the LSD passes for every key batch.
// (in RUNS of SPREAD_RUN consecutive items: a batch's claimants -- consecutive pops -- still bind neighbouring slots run by run,
// so that their `bound` bytes, position words and denial counters share memory lines; one slot at a time k_bind took 65 instead of
// 45 us and the sweep's push 77 instead of 40)
constexpr uint32_t SPREAD_RUN = 64;
__host__ __device__ inline uint32_t spread_position(uint32_t g, uint32_t m, uint32_t cols) {
    const uint32_t runs = m / SPREAD_RUN;                      // whole runs; the items behind them stay where they are
    const uint32_t run = g / SPREAD_RUN, in_run = g - run * SPREAD_RUN;
    if (run >= runs) return g;
    const uint32_t rows_full = runs / cols, rem = runs - rows_full * cols; // the last row holds `rem` runs
    const uint32_t row = run / cols, c = run - row * cols;
    return (c * rows_full + (c < rem ? c : rem) + row) * SPREAD_RUN + in_run;
}
__host__ __device__ inline uint32_t spread_cols(uint32_t m) { return m / SPREAD_RUN / 256u + 1u; } // ~256 rows: one per key range of the range path

// spread: slot order in, the order above out (slot 0 is still the first one handed out: it sits on top)
static __global__ __launch_bounds__(THREADS) void k_init_free(uint32_t* free_slots, uint8_t* bound, uint32_t capacity, uint32_t spread) {
    const uint32_t cols = spread_cols(capacity);
    for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < capacity; i += gridDim.x * THREADS) {
        // item i (= slot i) is popped spread_position-th: the stack is popped from its top, entry capacity - 1, down
        const uint32_t pop = spread ? spread_position(i, capacity, cols) : i;
        free_slots[capacity - 1u - pop] = i;
        bound[i] = 0;
    }
}

// One key, one thread, no other key stage running: find the key's slot or (insert) bind a fresh one.
// The single-request path (tc_rate_limit) uses this instead of the three-kernel batch protocol.
// Returns NO_SLOT if the key is absent (lookup) or cannot be bound (*full = true).
__device__ inline uint32_t find_or_bind_one(Table& t, const uint8_t* key, uint32_t len, bool insert, bool* full,
                                            unsigned long long* inserted_counter) {
    const uint64_t h = hash_key(key, len);
    const unsigned long long meta = entry_meta(h, len);
    uint64_t k0 = 0, k1 = 0;
    if (len <= ENTRY_KEY) short_key_words(key, len, k0, k1);
    uint64_t pos = h & t.nb_mask, tomb_pos = ~0ull, empty_pos = ~0ull;
    for (uint64_t probes = 0; probes <= t.nb_mask; ++probes, pos = (pos + 1) & t.nb_mask) {
        const Entry* en = &t.ktab[pos];
        const unsigned long long e = en->w;
        const uint32_t val = (uint32_t)e;
        if (e == 0ull) {
            empty_pos = pos;
            break;
        }
        if (val == VAL_TOMB) {
            if (tomb_pos == ~0ull) tomb_pos = pos;
            continue;
        }
        if (val & VAL_PENDING) continue; // (no batch is in flight: no pending claims)
        if ((e & 0xFFFFFFFF00000000ull) == meta && en->hash == h) {
            const uint32_t s = val - 2u;
            const bool same = len <= ENTRY_KEY ? (en->key[0] == k0 && en->key[1] == k1)
                                               : (t.rec[s].len == len && bytes_equal(stored_key(t, s, len), key, len));
            if (same) return s;
        }
    }
    if (!insert) return NO_SLOT;
    const uint64_t target = tomb_pos != ~0ull ? tomb_pos : empty_pos; // the chain's first tombstone is recycled
    if (target != ~0ull) {
        do {
            unsigned long long ovf = 0;
            if (len > INLINE_KEY) {
                const unsigned long long want = (len + 15u) & ~15u;
                ovf = atomicAdd(t.overflow_used, want);
                if (ovf + len > t.overflow_bytes) {
                    atomicAdd(t.overflow_used, ~want + 1ull); // give it back: shorter keys may still fit
                    break;
                }
            }
            const int old = atomicSub(t.free_top, 1);
            if (old <= 0) {
                atomicAdd(t.free_top, 1);
                break;
            }
            const uint32_t slot = t.free_slots[old - 1];
            KeyRec& kr = t.rec[slot];
            uint8_t* dst = kr.bytes;
            if (len > INLINE_KEY) {
                const uint64_t o64 = ovf;
                __builtin_memcpy(kr.bytes, &o64, 8);
                dst = t.overflow + (uint64_t)*t.overflow_half * t.overflow_bytes + ovf;
            }
            for (uint32_t b = 0; b < len; ++b) dst[b] = key[b];
            kr.hash = h;
            kr.len = len;
            kr.pos = (uint32_t)target;
            t.pos_col[slot] = (uint32_t)target;
            t.bound[slot] = 1;
            Entry* en = &t.ktab[target];
            en->hash = h;
            en->key[0] = k0;
            en->key[1] = k1;
            en->w = meta | (unsigned long long)(slot + 2u);
            resurrect_denials(t, h, key, len, slot);
            if (target == tomb_pos) atomicSub(&t.tombs[0], 1u);
            atomicAdd(inserted_counter, 1ull);
            return slot;
        } while (false);
    }
    if (insert) {
        *full = true;
        atomicExch(t.error_flag, 1u);
    }
    return NO_SLOT;
}

// Overflow arena (keys longer than 48 bytes).  Space is handed out by a bump pointer and a swept key's bytes
// are not given back one by one; instead the sweep compacts: once more than half of the current half is
// handed out, every bound long key is copied into the other half (fresh bump pointer) and the halves swap.
// Decided and done ON THE DEVICE (the asynchronous sweep never waits for the host).
// flag[0] = compact now, flag[1] = bytes handed out in the new half.
// (the decision itself is taken by mk::k_sweep_decide, together with the rebuild's)
__device__ __forceinline__ void overflow_decide(const Table& t, unsigned long long* __restrict__ flag) {
    flag[0] = *t.overflow_used > t.overflow_bytes / 2 ? 1ull : 0ull;
    flag[1] = 0ull;
}
// (one reservation per block and 2048 slots: a returning atomicAdd per copied key on the new half's cursor -- one
// word -- serialised the kernel: 1.7 ms for the ~3 M long keys of the configs[4] stream)
constexpr int COMPACT_ITEMS = 8;
__device__ __forceinline__ void overflow_compact(const Table& t, unsigned long long* __restrict__ flag) {
    if (flag[0] == 0ull) return; // (uniform over the grid)
    __shared__ uint32_t s_w[THREADS / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t from = (uint64_t)*t.overflow_half * t.overflow_bytes, to = (uint64_t)(*t.overflow_half ^ 1u) * t.overflow_bytes;
    const uint32_t chunk = THREADS * COMPACT_ITEMS;
    for (uint32_t c0 = blockIdx.x * chunk; c0 < t.capacity; c0 += gridDim.x * chunk) { // (block-uniform trip count)
        uint32_t len[COMPACT_ITEMS];
        uint64_t off[COMPACT_ITEMS];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < COMPACT_ITEMS; ++j) {
            const uint32_t s = c0 + j * THREADS + threadIdx.x;
            len[j] = 0;
            off[j] = 0;
            if (s < t.capacity && t.bound[s]) {
                const uint32_t l = t.rec[s].len;
                if (l > INLINE_KEY && l != NO_SLOT) {
                    len[j] = l;
                    __builtin_memcpy(&off[j], t.rec[s].bytes, 8);
                    mine += (l + 15u) & ~15u;
                }
            }
        }
        uint32_t v = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t x = __shfl_up(v, o, 64);
            if (lane >= o) v += x;
        }
        __syncthreads(); // (s_w / s_base of the previous chunk have been read)
        if (lane == 63) s_w[wave] = v;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < THREADS / 64; ++w) {
            if (w < wave) before += s_w[w];
            total += s_w[w];
        }
        if (total == 0) continue; // (block-uniform)
        if (threadIdx.x == 0) s_base = atomicAdd(&flag[1], (unsigned long long)total);
        __syncthreads();
        uint64_t at = s_base + before + (v - mine);
#pragma unroll
        for (int j = 0; j < COMPACT_ITEMS; ++j) {
            if (len[j] == 0) continue;
            const uint32_t s = c0 + j * THREADS + threadIdx.x;
            const uint8_t* src = t.overflow + from + off[j];
            uint8_t* dst = t.overflow + to + at;
            uint32_t b = 0;
            for (; b + 8 <= len[j]; b += 8) {
                uint64_t w8;
                __builtin_memcpy(&w8, src + b, 8);
                __builtin_memcpy(dst + b, &w8, 8);
            }
            for (; b < len[j]; ++b) dst[b] = src[b];
            __builtin_memcpy(t.rec[s].bytes, &at, 8);
            at += (len[j] + 15u) & ~15u;
        }
    }
}
__device__ __forceinline__ void overflow_swap(const Table& t, const unsigned long long* __restrict__ flag) { // (one thread, the kernel after)
    if (flag[0] != 0ull) {
        *t.overflow_used = flag[1];
        *t.overflow_half ^= 1u;
    }
}

// Rebuild (tombstones lengthen probe chains; inserts recycle the ones on their own chain, the rest stays): decided ON THE DEVICE so that
// a sweep never waits for the host -- mk::k_sweep_decide latches "tombstones (with the keys just unbound) > 1/4 of the
// table" into a flag word, the two kernels below do nothing unless it is set.
// Two launches do the rare work behind a sweep, near-empty when nothing is due (round 4: they were five): k_table_clear_compact
// clears the table for a rebuild and / or copies the long keys into the arena's other half (independent of each other);
// k_table_reinsert flips the arena's halves and re-enters every bound slot into the cleared table.
static __global__ __launch_bounds__(THREADS) void k_table_clear_compact(Table t, const uint32_t* __restrict__ flag, unsigned long long* __restrict__ oflag) {
    if (*flag != 0u) {
        // 32-byte entries as two 16-byte stores per thread
        typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
        v2u64* raw = reinterpret_cast<v2u64*>(t.ktab);
        const uint64_t n16 = (t.nb_mask + 1) * 2;
        const v2u64 z = {0ull, 0ull};
        for (uint64_t i = (uint64_t)blockIdx.x * THREADS + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * THREADS)
            __builtin_nontemporal_store(z, &raw[i]);
    }
    overflow_compact(t, oflag);
}

// re-enter every bound slot into the cleared table
// Round 5: the bound slots of a tile of 4 096 are first gathered into LDS, then entered with every lane busy.  A rebuild follows a sweep
// that unbound most keys (configs[4]'s first: 1.5 M of 10.5 M stay), and the kernel used to walk the slots lane by lane: one lane in
// eight had a key to enter, and so that no load sat under a branch the others fetched a record line anyway -- 2.9 M of those against
// 1.5 M useful ones, 349 us.  Now a thread reads 16 `bound` bytes at once, the block ranks its bound slots (one LDS atomic per wave),
// and the list is worked off four keys per thread: four records requested, four claims issued, then the fills.
constexpr int RE_PER = 16;                        // slots per thread and tile
constexpr uint32_t RE_TILE = THREADS * RE_PER;    // 4 096 slots: 16 KB of LDS for the list
static __global__ __launch_bounds__(THREADS) void k_table_reinsert(Table t, const uint32_t* __restrict__ flag, const unsigned long long* __restrict__ oflag) {
    if (blockIdx.x == 0 && threadIdx.x == 0) overflow_swap(t, oflag); // (nothing below looks at the arena)
    if (*flag == 0u) return;
    if (blockIdx.x == 0 && threadIdx.x < TOMB_SHARDS) t.tombs[threadIdx.x] = 0u;
    __shared__ uint32_t s_list[RE_TILE];
    __shared__ uint32_t s_fill;
    const int lane = threadIdx.x & 63;
    for (uint64_t tile0 = (uint64_t)blockIdx.x * RE_TILE; tile0 < t.capacity; tile0 += (uint64_t)gridDim.x * RE_TILE) {
        if (threadIdx.x == 0) s_fill = 0u;
        __syncthreads();
        const uint64_t mine0 = tile0 + (uint64_t)threadIdx.x * RE_PER;
        uint32_t w[RE_PER / 4];
        if (mine0 + RE_PER <= t.capacity) { // (hipMalloc aligns `bound`, tiles and threads start at multiples of 16)
            const uint4 v = *reinterpret_cast<const uint4*>(t.bound + mine0);
            w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < RE_PER / 4; ++q) {
                w[q] = 0u;
                for (int b = 0; b < 4; ++b) {
                    const uint64_t s = mine0 + (uint64_t)q * 4 + b;
                    if (s < t.capacity && t.bound[s]) w[q] |= 1u << (8 * b);
                }
            }
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < RE_PER / 4; ++q)
            for (int b = 0; b < 4; ++b) cnt += ((w[q] >> (8 * b)) & 0xFFu) != 0u;
        uint32_t incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        uint32_t base = 0;
        if (lane == 63) base = atomicAdd(&s_fill, incl); // (any order will do)
        base = __shfl(base, 63, 64);
        uint32_t at = base + incl - cnt;
#pragma unroll
        for (int q = 0; q < RE_PER / 4; ++q)
            for (int b = 0; b < 4; ++b)
                if ((w[q] >> (8 * b)) & 0xFFu) s_list[at++] = (uint32_t)(mine0 + (uint64_t)q * 4 + b);
        __syncthreads();
        const uint32_t total = s_fill;
        constexpr int RI = 4;
        for (uint32_t k0 = threadIdx.x; k0 < total; k0 += THREADS * RI) {
            uint32_t s[RI], len[RI];
            uint64_t h[RI], a[RI], b[RI], pos[RI];
            unsigned long long meta[RI];
            bool have[RI], taken[RI];
#pragma unroll
            for (int j = 0; j < RI; ++j) {
                have[j] = k0 + j * THREADS < total;
                s[j] = s_list[have[j] ? k0 + j * THREADS : k0];
            }
#pragma unroll
            for (int j = 0; j < RI; ++j) {
                const KeyRec& kr = t.rec[s[j]];
                h[j] = kr.hash;
                len[j] = kr.len;
                __builtin_memcpy(&a[j], kr.bytes, 8);
                __builtin_memcpy(&b[j], kr.bytes + 8, 8);
            }
#pragma unroll
            for (int j = 0; j < RI; ++j) { // claim with the pending pattern (nobody probes during a rebuild): four claims in flight
                meta[j] = entry_meta(h[j], len[j]);
                pos[j] = h[j] & t.nb_mask;
                taken[j] = false;
                if (have[j]) {
                    unsigned long long expected = 0ull;
                    taken[j] = !__hip_atomic_compare_exchange_strong(&t.ktab[pos[j]].w, &expected, meta[j] | (unsigned long long)VAL_PENDING, __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int j = 0; j < RI; ++j) {
                if (!have[j]) continue;
                while (taken[j]) { // the next entry of the chain
                    pos[j] = (pos[j] + 1) & t.nb_mask;
                    unsigned long long expected = 0ull;
                    taken[j] = !__hip_atomic_compare_exchange_strong(&t.ktab[pos[j]].w, &expected, meta[j] | (unsigned long long)VAL_PENDING, __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                uint64_t k0b = 0, k1b = 0;
                if (len[j] <= ENTRY_KEY) {
                    k0b = keep_bytes(a[j], len[j] < 8u ? len[j] : 8u);
                    k1b = len[j] > 8u ? keep_bytes(b[j], len[j] - 8u) : 0ull;
                }
                Entry* en = &t.ktab[pos[j]];
                en->hash = h[j];
                en->key[0] = k0b;
                en->key[1] = k1b;
                en->w = meta[j] | (unsigned long long)(s[j] + 2u);
                t.pos_col[s[j]] = (uint32_t)pos[j]; // (KeyRec::pos stays what it was at binding: writing it dirtied every bound key's
                                                    // record line, a quarter of the rebuild's 0.77 GB)
            }
        }
        __syncthreads(); // the list is reused by the next tile
    }
}

} // namespace kt
