// tcgpu.hip -- MI355X (gfx950) batched GCRA engine: HIP kernels + the C ABI of
// include/tcgpu.h.  One engine = one GPU-resident key store:
//
//   cells[capacity]   {tat i64, expiry u64}   16 B / slot   (mutable state)
//   rates[capacity]   {ei i64, dvt i64}       16 B / slot   (registered params)
//   bursts[capacity]  i64                      8 B / slot   (limit, result only)
//
// A batch is applied with the reference's sequential semantics
// (throttlecrab/src/core/rate_limiter.rs:102-250 applied in index order):
//   unique slots  : k_eval_unique        one lane per request
//   duplicates    : k_prep -> radix sort (slot, index) -> k_eval_sorted
//                   (closed form for uniform runs, serial walk otherwise)
//                   -> k_commit
// HBM-bound integer work; MFMA is not used (no dense contraction).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tcgpu.h"
#include "gcra_math.hpp"

using tc::Cell;
using tc::Decision;
using tc::Rate;

namespace {

constexpr int BLOCK = 256;
constexpr uint32_t F_REGISTERED = 1u; // Params.flags

// ---------------------------------------------------------------------------
// kernel argument block
// ---------------------------------------------------------------------------
struct Params {
    uint32_t n;
    uint32_t flags;
    const uint32_t* slot;
    const int64_t* burst;
    const int64_t* count;
    const int64_t* period;
    const int64_t* q;
    const int64_t* now;
    int64_t burst_s, count_s, period_s, q_s, now_s;
    uint8_t* allowed;
    int64_t* limit;
    int64_t* remaining;
    int64_t* reset;
    int64_t* retry;
    uint8_t* status;
    Cell* cells;
    const Rate* rates;
    const int64_t* bursts;
    uint64_t capacity;
    unsigned long long* counters;
};

struct Req {
    int64_t ei, dvt, q, now, limit;
    int status;
};

__device__ __forceinline__ Req load_req(const Params& p, uint32_t i, uint32_t slot) {
    Req r;
    r.q = p.q ? p.q[i] : p.q_s;
    r.now = p.now ? p.now[i] : p.now_s;
    r.ei = r.dvt = r.limit = 0;
    if (slot >= p.capacity) {
        r.status = tc::ST_INTERNAL;
        return r;
    }
    if (p.flags & F_REGISTERED) {
        const Rate rt = p.rates[slot];
        r.limit = p.bursts[slot];
        r.ei = rt.ei;
        r.dvt = rt.dvt;
        if (r.q < 0) r.status = tc::ST_NEGATIVE_QUANTITY;          // rate_limiter.rs:111
        else if (r.limit <= 0) r.status = tc::ST_INVALID_RATE_LIMIT; // never registered
        else r.status = tc::check_request(r.q, r.now, r.dvt);
    } else {
        const int64_t burst = p.burst ? p.burst[i] : p.burst_s;
        const int64_t count = p.count ? p.count[i] : p.count_s;
        const int64_t period = p.period ? p.period[i] : p.period_s;
        r.limit = burst;
        r.status = tc::derive_request(burst, count, period, r.q, r.now, r.ei, r.dvt);
    }
    return r;
}

__device__ __forceinline__ void write_out(const Params& p, uint32_t i, const Req& r, const Decision& d) {
    const bool ok = r.status == tc::ST_OK;
    if (p.allowed) p.allowed[i] = (ok && d.allowed) ? 1 : 0;
    if (p.status) p.status[i] = (uint8_t)r.status;
    if (p.limit) p.limit[i] = ok ? r.limit : 0;
    if (p.remaining) p.remaining[i] = ok ? d.remaining : 0;
    if (p.reset) p.reset[i] = ok ? d.reset_after : 0;
    if (p.retry) p.retry[i] = ok ? d.retry_after : 0;
}

// sum three small per-thread counts over the block, one atomic each per block
__device__ __forceinline__ void block_count3(uint32_t a, uint32_t b, uint32_t c, unsigned long long* counters) {
    __shared__ uint32_t s_cnt[3][BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_cnt[0][wave] = a;
        s_cnt[1][wave] = b;
        s_cnt[2][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t t = 0;
        for (int w = 0; w < BLOCK / 64; ++w) t += s_cnt[threadIdx.x][w];
        if (t) {
            const int which = threadIdx.x == 0 ? TC_CNT_ALLOWED : (threadIdx.x == 1 ? TC_CNT_DENIED : TC_CNT_ERRORS);
            atomicAdd(&counters[which], (unsigned long long)t);
            atomicAdd(&counters[TC_CNT_TOTAL], (unsigned long long)t);
        }
    }
}

// ---------------------------------------------------------------------------
// K1: one lane per request, slots unique within the batch
// ---------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(BLOCK) void k_eval_unique(Params p) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t na = 0, nd = 0, ne = 0;
    if (i < p.n) {
        const uint32_t slot = p.slot[i];
        const Req r = load_req(p, i, slot);
        Decision d;
        d.allowed = false;
        d.remaining = d.reset_after = d.retry_after = 0;
        if (r.status == tc::ST_OK) {
            Cell c = p.cells[slot];
            d = tc::gcra_step<FULL>(c, r.ei, r.dvt, r.q, r.now);
            if (d.allowed) p.cells[slot] = c;
            na = d.allowed;
            nd = !d.allowed;
        } else {
            ne = 1;
        }
        write_out(p, i, r, d);
    }
    block_count3(na, nd, ne, p.counters);
}

// ---------------------------------------------------------------------------
// K2a: sort keys = min(slot, capacity) (one sentinel segment for bad slots)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_prep(const uint32_t* __restrict__ slot, uint32_t n, uint32_t cap,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const uint32_t s = slot[i];
        keys[i] = s < cap ? s : cap;
        vals[i] = i;
    }
}

// block-wide inclusive max-scan (values are position+1, 0 = none)
__device__ __forceinline__ uint32_t block_scan_max(uint32_t v) {
    __shared__ uint32_t s_wmax[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, 64);
        if (lane >= off) v = max(v, o);
    }
    if (lane == 63) s_wmax[wave] = v;
    __syncthreads();
    uint32_t carry = 0;
    for (int w = 0; w < wave; ++w) carry = max(carry, s_wmax[w]);
    return max(v, carry);
}

// ---------------------------------------------------------------------------
// K2b: evaluate over the (slot, index)-sorted batch.
//   UNIFORM (one `now`, one `quantity`, per-slot or scalar params): every
//     request of a slot's segment is identical, so lane r of the segment
//     derives the state left by its r predecessors in closed form
//     (tc::run_form) and applies the ordinary step to it.  The single lane
//     that performs the segment's last allowed step parks the new cell in
//     pend[] (k_commit stores it) so that no lane of the segment can read a
//     half-updated cell.
//   otherwise: the segment head walks its segment in index order.
// ---------------------------------------------------------------------------
template <bool FULL, bool UNIFORM>
__global__ __launch_bounds__(BLOCK) void k_eval_sorted(Params p, const uint32_t* __restrict__ ss,
                                                       const uint32_t* __restrict__ si, Cell* __restrict__ pend,
                                                       uint8_t* __restrict__ pend_flag) {
    const uint32_t n = p.n;
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = k < n;
    const uint32_t slot = valid ? ss[k] : 0xFFFFFFFFu;
    const bool head = valid && (k == 0 || ss[k - 1] != slot);
    uint32_t na = 0, nd = 0, ne = 0;

    if (!UNIFORM) {
        if (head) {
            Cell c;
            c.tat = 0;
            c.expiry = 0;
            const bool in_range = slot < p.capacity;
            if (in_range) c = p.cells[slot];
            bool dirty = false;
            for (uint32_t j = k; j < n && ss[j] == slot; ++j) {
                const uint32_t i = si[j];
                const Req r = load_req(p, i, slot);
                Decision d;
                d.allowed = false;
                d.remaining = d.reset_after = d.retry_after = 0;
                if (r.status == tc::ST_OK) {
                    d = tc::gcra_step<FULL>(c, r.ei, r.dvt, r.q, r.now);
                    dirty |= d.allowed;
                    na += d.allowed;
                    nd += !d.allowed;
                } else {
                    ne += 1;
                }
                write_out(p, i, r, d);
            }
            if (dirty) p.cells[slot] = c;
        }
        block_count3(na, nd, ne, p.counters);
        return;
    }

    // ---- UNIFORM: rank of this lane inside its segment ----
    __shared__ uint32_t s_start;
    const uint32_t block_start = blockIdx.x * BLOCK;
    if (threadIdx.x == 0 && valid && !head) {
        // segment of ss[block_start] began in an earlier block: lower_bound
        uint32_t lo = 0, hi = block_start;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (ss[mid] < slot) lo = mid + 1;
            else hi = mid;
        }
        s_start = lo;
    }
    const uint32_t hp = block_scan_max(head ? k + 1 : 0u); // has a __syncthreads()
    const uint32_t seg_start = hp ? hp - 1 : s_start;
    bool writer = false;
    Cell wcell;
    wcell.tat = 0;
    wcell.expiry = 0;
    if (valid) {
        const uint32_t r = k - seg_start;
        const bool is_last = (k + 1 == n) || (ss[k + 1] != slot);
        const uint32_t i = si[k];
        const Req rq = load_req(p, i, slot);
        Decision d;
        d.allowed = false;
        d.remaining = d.reset_after = d.retry_after = 0;
        if (rq.status != tc::ST_OK) {
            ne = 1;
            write_out(p, i, rq, d);
        } else {
            Cell c = p.cells[slot];
            const Decision d0 = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now); // c = cell after request 0
            if (!d0.allowed) {
                // request 0 denied => state untouched => every request of the run is request 0
                nd = 1;
                write_out(p, i, rq, d0);
            } else {
                const tc::RunForm f = tc::run_form(c, rq.ei, rq.dvt, rq.q, rq.now);
                if (r == 0) {
                    na = 1;
                    write_out(p, i, rq, d0);
                    if (is_last || (f.regular && f.n_tot == 1)) {
                        writer = true;
                        wcell = c;
                    } else if (!f.regular) {
                        // irregular run (saturation, zero increment, immediate expiry):
                        // walk the rest of the segment one request at a time
                        for (uint32_t j = k + 1; j < n && ss[j] == slot; ++j) {
                            const Decision dj = tc::gcra_step<FULL>(c, rq.ei, rq.dvt, rq.q, rq.now);
                            na += dj.allowed;
                            nd += !dj.allowed;
                            write_out(p, si[j], rq, dj);
                        }
                        writer = true;
                        wcell = c;
                    }
                } else if (f.regular) {
                    const int64_t j = (int64_t)r < f.n_tot ? (int64_t)r : f.n_tot;
                    Cell v;
                    v.tat = f.new0 + (j - 1) * f.inc;
                    v.expiry = UINT64_MAX;
                    d = tc::gcra_step<FULL>(v, rq.ei, rq.dvt, rq.q, rq.now);
                    na = d.allowed;
                    nd = !d.allowed;
                    write_out(p, i, rq, d);
                    if (d.allowed && (is_last || (int64_t)r + 1 == f.n_tot)) {
                        writer = true;
                        wcell = v;
                    }
                }
                // irregular && r > 0: the head lane produced this request's outputs
            }
        }
        pend_flag[k] = writer ? 1 : 0;
        if (writer) pend[k] = wcell;
    }
    block_count3(na, nd, ne, p.counters);
}

__global__ __launch_bounds__(BLOCK) void k_commit(const uint32_t* __restrict__ ss, uint32_t n,
                                                  const Cell* __restrict__ pend, const uint8_t* __restrict__ pend_flag,
                                                  Cell* __restrict__ cells) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n && pend_flag[k]) cells[ss[k]] = pend[k];
}

// allowed[] bytes -> bitmask (wavefront ballot, one u64 per wave)
__global__ __launch_bounds__(BLOCK) void k_pack_bits(const uint8_t* __restrict__ allowed, uint32_t n,
                                                     uint64_t* __restrict__ bits) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool a = i < n && allowed[i];
    const unsigned long long m = __ballot(a);
    if ((threadIdx.x & 63) == 0 && i < n) bits[i >> 6] = m;
}

// ---------------------------------------------------------------------------
// K4: expiry sweep == AdaptiveStore::cleanup (adaptive_cleanup.rs:173-203)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_sweep(Cell* __restrict__ cells, uint64_t capacity, int64_t now,
                                                 unsigned long long* counters, unsigned long long* removed_out) {
    uint32_t removed = 0, live = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) {
        Cell c = cells[i];
        if (c.expiry != 0) {
            if (!(c.expiry > (uint64_t)now)) { // retain(|exp| *exp > now)
                c.tat = 0;
                c.expiry = 0;
                cells[i] = c;
                removed++;
            } else {
                live++;
            }
        }
    }
    __shared__ uint32_t s_r[BLOCK / 64], s_l[BLOCK / 64];
    for (int off = 32; off > 0; off >>= 1) {
        removed += __shfl_down(removed, off, 64);
        live += __shfl_down(live, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_r[threadIdx.x >> 6] = removed;
        s_l[threadIdx.x >> 6] = live;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0, l = 0;
        for (int w = 0; w < BLOCK / 64; ++w) {
            r += s_r[w];
            l += s_l[w];
        }
        if (r) {
            atomicAdd(&counters[TC_CNT_SWEPT], (unsigned long long)r);
            atomicAdd(removed_out, (unsigned long long)r);
        }
        if (l) atomicAdd(&counters[TC_CNT_LIVE_SLOTS], (unsigned long long)l);
    }
}

__global__ __launch_bounds__(BLOCK) void k_fill_rates(Rate* __restrict__ rates, int64_t* __restrict__ bursts,
                                                      uint64_t capacity, Rate r, int64_t burst) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < capacity; i += (uint64_t)gridDim.x * BLOCK) {
        rates[i] = r;
        bursts[i] = burst;
    }
}

__global__ __launch_bounds__(BLOCK) void k_scatter_rates(Rate* __restrict__ rates, int64_t* __restrict__ bursts,
                                                         const uint32_t* __restrict__ slots, const Rate* __restrict__ src_r,
                                                         const int64_t* __restrict__ src_b, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        const uint64_t s = slots ? slots[i] : i;
        rates[s] = src_r[i];
        bursts[s] = src_b[i];
    }
}

// `trait Store` shims on one resolved slot (store/mod.rs:85-133,
// adaptive_cleanup.rs:221-279).  op: 0 get, 1 cas, 2 set_nx
struct StoreOpResult {
    int64_t value;
    int32_t flag;
    int32_t pad;
};
__global__ void k_store_op(Cell* cells, uint64_t slot, int op, int64_t a, int64_t b, uint64_t ttl, int64_t now,
                           StoreOpResult* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Cell c = cells[slot];
    const bool live = c.expiry > (uint64_t)now;
    StoreOpResult r;
    r.value = 0;
    r.flag = 0;
    r.pad = 0;
    uint64_t e = (uint64_t)now + ttl;
    if (e < ttl) e = UINT64_MAX;
    if (op == 0) {
        if (live) {
            r.value = c.tat;
            r.flag = 1;
        }
    } else if (op == 1) {
        if (live && c.tat == a) {
            c.tat = b;
            c.expiry = e;
            cells[slot] = c;
            r.flag = 1;
        }
    } else {
        if (!live) {
            c.tat = a;
            c.expiry = e;
            cells[slot] = c;
            r.flag = 1;
        }
    }
    *out = r;
}

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
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    uint64_t capacity = 0, max_batch = 0;
    uint32_t cfg_flags = 0;

    Cell* cells = nullptr;
    Rate* rates = nullptr;
    int64_t* bursts = nullptr;
    unsigned long long* counters = nullptr; // TC_CNT_COUNT + 1 (scratch word)

    // grouping scratch (max_batch each)
    uint32_t *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    Cell* pend = nullptr;
    uint8_t* pend_flag = nullptr;
    uint8_t* allowed_tmp = nullptr;
    StoreOpResult* op_result = nullptr;

    // staging for host-pointer batches (lazy)
    struct Stage {
        uint32_t* slot = nullptr;
        int64_t* in[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        uint8_t* allowed = nullptr;
        uint64_t* bits = nullptr;
        int64_t* out[4] = {nullptr, nullptr, nullptr, nullptr};
        uint8_t* status = nullptr;
        bool ready = false;
    } stage;

    uint64_t batches = 0; // TC_CNT_BATCHES is kept on the host

    // optional per-stage HIP-event timing (tc_profile_*)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_stage;
    size_t prof_used = 0;
    double prof_ms[TC_STAGE_COUNT] = {0};
    uint64_t prof_calls[TC_STAGE_COUNT] = {0};

    std::string err;
};

// record "stage begins here" (stage < 0: end of the batch) on the engine's stream
static void prof_mark(tc_engine* e, int stage) {
    if (!e->prof_on) return;
    if (e->prof_used == e->prof_ev.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        e->prof_ev.push_back(ev);
        e->prof_stage.push_back(-1);
    }
    e->prof_stage[e->prof_used] = stage;
    (void)hipEventRecord(e->prof_ev[e->prof_used], e->stream);
    e->prof_used++;
}

#define TC_HIP(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _rc = (call);                                                                     \
        if (_rc != hipSuccess) {                                                                     \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_rc);                           \
            return TC_E_HIP;                                                                         \
        }                                                                                            \
    } while (0)

static int fail(tc_engine* e, int code, const char* msg) {
    if (e) e->err = msg;
    return code;
}

extern "C" uint32_t tc_abi_version(void) { return TCGPU_ABI_VERSION; }

extern "C" const char* tc_last_error(const tc_engine* e) { return e ? e->err.c_str() : "null engine"; }

static int engine_alloc(tc_engine* e) {
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    const uint64_t cap = e->capacity, mb = e->max_batch;
    TC_HIP(e, hipMalloc(&e->cells, cap * sizeof(Cell)));
    TC_HIP(e, hipMalloc(&e->rates, cap * sizeof(Rate)));
    TC_HIP(e, hipMalloc(&e->bursts, cap * sizeof(int64_t)));
    TC_HIP(e, hipMalloc(&e->counters, (TC_CNT_COUNT + 1) * sizeof(unsigned long long)));
    TC_HIP(e, hipMemsetAsync(e->cells, 0, cap * sizeof(Cell), e->stream));
    TC_HIP(e, hipMemsetAsync(e->rates, 0, cap * sizeof(Rate), e->stream));
    TC_HIP(e, hipMemsetAsync(e->bursts, 0, cap * sizeof(int64_t), e->stream));
    TC_HIP(e, hipMemsetAsync(e->counters, 0, (TC_CNT_COUNT + 1) * sizeof(unsigned long long), e->stream));
    TC_HIP(e, hipMalloc(&e->keys_a, mb * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&e->keys_b, mb * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&e->vals_a, mb * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&e->vals_b, mb * sizeof(uint32_t)));
    TC_HIP(e, hipMalloc(&e->pend, mb * sizeof(Cell)));
    TC_HIP(e, hipMalloc(&e->pend_flag, mb));
    TC_HIP(e, hipMalloc(&e->allowed_tmp, mb));
    TC_HIP(e, hipMalloc(&e->op_result, sizeof(StoreOpResult)));
    size_t tmp = 0;
    TC_HIP(e, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, e->keys_a, e->keys_b, e->vals_a, e->vals_b,
                                                 (int)mb, 0, 32, e->stream));
    e->sort_tmp_bytes = tmp;
    TC_HIP(e, hipMalloc(&e->sort_tmp, tmp ? tmp : 16));
    TC_HIP(e, hipStreamSynchronize(e->stream));
    return TC_E_OK;
}

extern "C" tc_engine* tc_engine_create(const tc_config* cfg, int* err) {
    int dummy;
    if (!err) err = &dummy;
    *err = TC_E_OK;
    if (!cfg || cfg->struct_size < sizeof(tc_config) || cfg->capacity == 0 || cfg->max_batch == 0 ||
        cfg->capacity >= 0x7FFFFFFFull || cfg->max_batch >= 0x7FFFFFFFull) {
        *err = TC_E_INVALID_ARG;
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device_id < 0 || cfg->device_id >= ndev) {
        *err = TC_E_NO_DEVICE;
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
    if (cfg->flags & TC_CFG_KEY_MODE) {
        *err = TC_E_UNSUPPORTED;
        delete e;
        return nullptr;
    }
    int rc = engine_alloc(e);
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
    if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    void* ptrs[] = {e->cells, e->rates, e->bursts, e->counters, e->keys_a, e->keys_b, e->vals_a, e->vals_b,
                    e->sort_tmp, e->pend, e->pend_flag, e->allowed_tmp, e->op_result, e->stage.slot,
                    e->stage.in[0], e->stage.in[1], e->stage.in[2], e->stage.in[3], e->stage.in[4],
                    e->stage.allowed, e->stage.bits, e->stage.out[0], e->stage.out[1], e->stage.out[2],
                    e->stage.out[3], e->stage.status};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

extern "C" int tc_engine_set_stream(tc_engine* e, void* hip_stream) {
    if (!e) return TC_E_INVALID_ARG;
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return TC_E_OK;
}

extern "C" int tc_synchronize(tc_engine* e) {
    if (!e) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipStreamSynchronize(e->stream));
    return TC_E_OK;
}

extern "C" int tc_register_params_uniform(tc_engine* e, int64_t max_burst, int64_t count_per_period, int64_t period) {
    if (!e) return TC_E_INVALID_ARG;
    Rate r;
    if (tc::derive_rate(max_burst, count_per_period, period, r.ei, r.dvt) != tc::ST_OK)
        return fail(e, TC_E_INVALID_ARG, "tc_register_params_uniform: invalid (burst,count,period)");
    TC_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_fill_rates, dim3(std::min<uint64_t>(nblocks(e->capacity), 4096)), dim3(BLOCK), 0, e->stream,
                       e->rates, e->bursts, e->capacity, r, max_burst);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipStreamSynchronize(e->stream));
    return TC_E_OK;
}

extern "C" int tc_register_params(tc_engine* e, uint64_t n, const uint32_t* slots, const int64_t* max_burst,
                                  const int64_t* count_per_period, const int64_t* period) {
    if (!e || !max_burst || !count_per_period || !period) return TC_E_INVALID_ARG;
    if (n == 0) return TC_E_OK;
    if (!slots && n > e->capacity) return fail(e, TC_E_INVALID_ARG, "tc_register_params: n > capacity");
    std::vector<Rate> hr(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (slots && slots[i] >= e->capacity) return fail(e, TC_E_INVALID_ARG, "tc_register_params: slot out of range");
        if (tc::derive_rate(max_burst[i], count_per_period[i], period[i], hr[i].ei, hr[i].dvt) != tc::ST_OK)
            return fail(e, TC_E_INVALID_ARG, "tc_register_params: invalid (burst,count,period)");
    }
    TC_HIP(e, hipSetDevice(e->device));
    Rate* d_r = nullptr;
    int64_t* d_b = nullptr;
    uint32_t* d_s = nullptr;
    TC_HIP(e, hipMalloc(&d_r, n * sizeof(Rate)));
    TC_HIP(e, hipMalloc(&d_b, n * sizeof(int64_t)));
    if (slots) TC_HIP(e, hipMalloc(&d_s, n * sizeof(uint32_t)));
    TC_HIP(e, hipMemcpyAsync(d_r, hr.data(), n * sizeof(Rate), hipMemcpyHostToDevice, e->stream));
    TC_HIP(e, hipMemcpyAsync(d_b, max_burst, n * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
    if (slots) TC_HIP(e, hipMemcpyAsync(d_s, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_scatter_rates, dim3(nblocks(n)), dim3(BLOCK), 0, e->stream, e->rates, e->bursts, d_s, d_r, d_b, n);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipStreamSynchronize(e->stream));
    (void)hipFree(d_r);
    (void)hipFree(d_b);
    if (d_s) (void)hipFree(d_s);
    return TC_E_OK;
}

// ---- batch over slots -------------------------------------------------------
static int stage_ensure(tc_engine* e) {
    if (e->stage.ready) return TC_E_OK;
    const uint64_t mb = e->max_batch;
    TC_HIP(e, hipMalloc(&e->stage.slot, mb * sizeof(uint32_t)));
    for (int j = 0; j < 5; ++j) TC_HIP(e, hipMalloc(&e->stage.in[j], mb * sizeof(int64_t)));
    TC_HIP(e, hipMalloc(&e->stage.allowed, mb));
    TC_HIP(e, hipMalloc(&e->stage.bits, ((mb + 63) / 64) * sizeof(uint64_t)));
    for (int j = 0; j < 4; ++j) TC_HIP(e, hipMalloc(&e->stage.out[j], mb * sizeof(int64_t)));
    TC_HIP(e, hipMalloc(&e->stage.status, mb));
    e->stage.ready = true;
    return TC_E_OK;
}

// all pointers in `b` are device pointers here
static int run_slots_device(tc_engine* e, const tc_batch& b) {
    const uint32_t n = (uint32_t)b.n;
    Params p;
    p.n = n;
    p.flags = (b.flags & TC_B_REGISTERED_PARAMS) ? F_REGISTERED : 0u;
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
    p.cells = e->cells;
    p.rates = e->rates;
    p.bursts = e->bursts;
    p.capacity = e->capacity;
    p.counters = e->counters;
    const bool full = p.remaining || p.reset || p.retry;
    const dim3 grid(nblocks(n)), block(BLOCK);
    hipStream_t s = e->stream;

    if (b.flags & TC_B_UNIQUE_SLOTS) {
        prof_mark(e, TC_STAGE_EVAL);
        if (full) hipLaunchKernelGGL(k_eval_unique<true>, grid, block, 0, s, p);
        else hipLaunchKernelGGL(k_eval_unique<false>, grid, block, 0, s, p);
    } else {
        prof_mark(e, TC_STAGE_PREP);
        hipLaunchKernelGGL(k_prep, grid, block, 0, s, b.slot, n, (uint32_t)e->capacity, e->keys_a, e->vals_a);
        prof_mark(e, TC_STAGE_SORT);
        size_t tmp = e->sort_tmp_bytes;
        const int end_bit = std::max(1, bit_width_u64(e->capacity));
        TC_HIP(e, hipcub::DeviceRadixSort::SortPairs(e->sort_tmp, tmp, e->keys_a, e->keys_b, e->vals_a, e->vals_b,
                                                     (int)n, 0, end_bit, s));
        const bool params_by_slot = (b.flags & TC_B_REGISTERED_PARAMS) || (!p.burst && !p.count && !p.period);
        const bool uniform = !p.q && !p.now && params_by_slot;
        prof_mark(e, TC_STAGE_EVAL);
        if (uniform) {
            if (full) hipLaunchKernelGGL((k_eval_sorted<true, true>), grid, block, 0, s, p, e->keys_b, e->vals_b, e->pend, e->pend_flag);
            else hipLaunchKernelGGL((k_eval_sorted<false, true>), grid, block, 0, s, p, e->keys_b, e->vals_b, e->pend, e->pend_flag);
            prof_mark(e, TC_STAGE_COMMIT);
            hipLaunchKernelGGL(k_commit, grid, block, 0, s, e->keys_b, n, e->pend, e->pend_flag, e->cells);
        } else {
            if (full) hipLaunchKernelGGL((k_eval_sorted<true, false>), grid, block, 0, s, p, e->keys_b, e->vals_b, e->pend, e->pend_flag);
            else hipLaunchKernelGGL((k_eval_sorted<false, false>), grid, block, 0, s, p, e->keys_b, e->vals_b, e->pend, e->pend_flag);
        }
    }
    if (b.allowed_bits) {
        prof_mark(e, TC_STAGE_PACK);
        hipLaunchKernelGGL(k_pack_bits, grid, block, 0, s, p.allowed, n, b.allowed_bits);
    }
    prof_mark(e, -1);
    TC_HIP(e, hipGetLastError());
    return TC_E_OK;
}


extern "C" int tc_rate_limit_batch_slots(tc_engine* e, const tc_batch* bp) {
    if (!e || !bp || bp->struct_size < sizeof(tc_batch)) return TC_E_INVALID_ARG;
    const tc_batch& b = *bp;
    if (b.n == 0) return TC_E_OK;
    if (b.n > e->max_batch) return fail(e, TC_E_BATCH_TOO_LARGE, "batch larger than max_batch");
    if (!b.slot) return fail(e, TC_E_INVALID_ARG, "slot column is NULL");
    TC_HIP(e, hipSetDevice(e->device));
    if (b.flags & TC_B_DEVICE_PTRS) {
        e->batches++;
        return run_slots_device(e, b);
    }
    // host pointers: stage in, run, stage out, synchronise
    int rc = stage_ensure(e);
    if (rc != TC_E_OK) return rc;
    const uint64_t n = b.n;
    hipStream_t s = e->stream;
    tc_batch d = b;
    d.flags |= TC_B_DEVICE_PTRS;
    TC_HIP(e, hipMemcpyAsync(e->stage.slot, b.slot, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    d.slot = e->stage.slot;
    const int64_t* hin[5] = {b.max_burst, b.count_per_period, b.period, b.quantity, b.now_ns};
    const int64_t** din[5] = {&d.max_burst, &d.count_per_period, &d.period, &d.quantity, &d.now_ns};
    for (int j = 0; j < 5; ++j) {
        if (hin[j]) {
            TC_HIP(e, hipMemcpyAsync(e->stage.in[j], hin[j], n * sizeof(int64_t), hipMemcpyHostToDevice, s));
            *din[j] = e->stage.in[j];
        }
    }
    d.allowed = b.allowed ? e->stage.allowed : nullptr;
    d.allowed_bits = b.allowed_bits ? e->stage.bits : nullptr;
    d.limit = b.limit ? e->stage.out[0] : nullptr;
    d.remaining = b.remaining ? e->stage.out[1] : nullptr;
    d.reset_after_ns = b.reset_after_ns ? e->stage.out[2] : nullptr;
    d.retry_after_ns = b.retry_after_ns ? e->stage.out[3] : nullptr;
    d.status = b.status ? e->stage.status : nullptr;
    e->batches++;
    rc = run_slots_device(e, d);
    if (rc != TC_E_OK) return rc;
    if (b.allowed) TC_HIP(e, hipMemcpyAsync(b.allowed, e->stage.allowed, n, hipMemcpyDeviceToHost, s));
    if (b.allowed_bits)
        TC_HIP(e, hipMemcpyAsync(b.allowed_bits, e->stage.bits, ((n + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    int64_t* hout[4] = {b.limit, b.remaining, b.reset_after_ns, b.retry_after_ns};
    for (int j = 0; j < 4; ++j)
        if (hout[j]) TC_HIP(e, hipMemcpyAsync(hout[j], e->stage.out[j], n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (b.status) TC_HIP(e, hipMemcpyAsync(b.status, e->stage.status, n, hipMemcpyDeviceToHost, s));
    TC_HIP(e, hipStreamSynchronize(s));
    return TC_E_OK;
}

extern "C" int tc_rate_limit_batch_keys(tc_engine* e, const tc_batch* b) {
    (void)b;
    return fail(e, TC_E_UNSUPPORTED, "key mode not built yet");
}

extern "C" int tc_rate_limit(tc_engine* e, const uint8_t* key, size_t key_len, int64_t max_burst,
                             int64_t count_per_period, int64_t period, int64_t quantity, int64_t now_ns,
                             tc_result* out) {
    if (!e || !out || (!key && key_len)) return TC_E_INVALID_ARG;
    if (e->cfg_flags & TC_CFG_KEY_MODE) return fail(e, TC_E_UNSUPPORTED, "key mode not built yet");
    // slot-mode engines: the key is the 4-byte little-endian slot id
    if (key_len != 4) return fail(e, TC_E_INVALID_ARG, "slot-mode engine: key must be a 4-byte slot id");
    uint32_t slot;
    memcpy(&slot, key, 4);
    uint8_t allowed = 0, status = 0;
    int64_t limit = 0, remaining = 0, reset = 0, retry = 0;
    tc_batch b;
    memset(&b, 0, sizeof b);
    b.struct_size = sizeof b;
    b.n = 1;
    b.slot = &slot;
    b.max_burst_scalar = max_burst;
    b.count_per_period_scalar = count_per_period;
    b.period_scalar = period;
    b.quantity_scalar = quantity;
    b.now_ns_scalar = now_ns;
    b.allowed = &allowed;
    b.status = &status;
    b.limit = &limit;
    b.remaining = &remaining;
    b.reset_after_ns = &reset;
    b.retry_after_ns = &retry;
    int rc = tc_rate_limit_batch_slots(e, &b);
    if (rc != TC_E_OK) return rc;
    out->allowed = allowed;
    out->status = status;
    out->limit = limit;
    out->remaining = remaining;
    out->reset_after_ns = reset;
    out->retry_after_ns = retry;
    return TC_E_OK;
}

extern "C" int tc_sweep_expired(tc_engine* e, int64_t now_ns, uint64_t* removed) {
    if (!e) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    unsigned long long* scratch = e->counters + TC_CNT_COUNT;
    TC_HIP(e, hipMemsetAsync(scratch, 0, sizeof(unsigned long long), e->stream));
    TC_HIP(e, hipMemsetAsync(e->counters + TC_CNT_LIVE_SLOTS, 0, sizeof(unsigned long long), e->stream));
    hipLaunchKernelGGL(k_sweep, dim3(std::min<uint64_t>(nblocks(e->capacity), 2048)), dim3(BLOCK), 0, e->stream,
                       e->cells, e->capacity, now_ns, e->counters, scratch);
    TC_HIP(e, hipGetLastError());
    unsigned long long r = 0;
    TC_HIP(e, hipMemcpyAsync(&r, scratch, sizeof r, hipMemcpyDeviceToHost, e->stream));
    TC_HIP(e, hipStreamSynchronize(e->stream));
    if (removed) *removed = r;
    return TC_E_OK;
}

extern "C" int tc_counters(tc_engine* e, uint64_t out[TC_CNT_COUNT]) {
    if (!e || !out) return TC_E_INVALID_ARG;
    TC_HIP(e, hipSetDevice(e->device));
    TC_HIP(e, hipMemcpyAsync(out, e->counters, TC_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
    TC_HIP(e, hipStreamSynchronize(e->stream));
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
    TC_HIP(e, hipStreamSynchronize(e->stream));
    for (size_t i = 0; i + 1 < e->prof_used; ++i) {
        const int st = e->prof_stage[i];
        if (st < 0) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]) == hipSuccess) {
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

static int store_slot_of(tc_engine* e, const uint8_t* key, size_t key_len, uint64_t* slot) {
    if (e->cfg_flags & TC_CFG_KEY_MODE) return fail(e, TC_E_UNSUPPORTED, "key mode not built yet");
    if (!key || key_len != 4) return fail(e, TC_E_INVALID_ARG, "slot-mode engine: key must be a 4-byte slot id");
    uint32_t s;
    memcpy(&s, key, 4);
    if (s >= e->capacity) return fail(e, TC_E_INVALID_ARG, "slot out of range");
    *slot = s;
    return TC_E_OK;
}

static int store_op(tc_engine* e, uint64_t slot, int op, int64_t a, int64_t b, uint64_t ttl, int64_t now,
                    StoreOpResult* r) {
    if (now < 0) return fail(e, TC_E_INVALID_ARG, "now_ns < 0");
    TC_HIP(e, hipSetDevice(e->device));
    hipLaunchKernelGGL(k_store_op, dim3(1), dim3(64), 0, e->stream, e->cells, slot, op, a, b, ttl, now, e->op_result);
    TC_HIP(e, hipGetLastError());
    TC_HIP(e, hipMemcpyAsync(r, e->op_result, sizeof *r, hipMemcpyDeviceToHost, e->stream));
    TC_HIP(e, hipStreamSynchronize(e->stream));
    return TC_E_OK;
}

extern "C" int tc_store_get(tc_engine* e, const uint8_t* key, size_t key_len, int64_t now_ns, int64_t* value, int* found) {
    if (!e || !value || !found) return TC_E_INVALID_ARG;
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, &slot);
    if (rc != TC_E_OK) return rc;
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
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, &slot);
    if (rc != TC_E_OK) return rc;
    StoreOpResult r;
    rc = store_op(e, slot, 1, old_value, new_value, ttl_ns, now_ns, &r);
    if (rc != TC_E_OK) return rc;
    *swapped = r.flag;
    return TC_E_OK;
}

extern "C" int tc_store_set_if_not_exists_with_ttl(tc_engine* e, const uint8_t* key, size_t key_len, int64_t value,
                                                   uint64_t ttl_ns, int64_t now_ns, int* was_set) {
    if (!e || !was_set) return TC_E_INVALID_ARG;
    uint64_t slot;
    int rc = store_slot_of(e, key, key_len, &slot);
    if (rc != TC_E_OK) return rc;
    StoreOpResult r;
    rc = store_op(e, slot, 2, value, 0, ttl_ns, now_ns, &r);
    if (rc != TC_E_OK) return rc;
    *was_set = r.flag;
    return TC_E_OK;
}

extern "C" int tc_read_state(tc_engine* e, uint64_t first, uint64_t n, int64_t* tat, uint64_t* expiry) {
    if (!e || first + n > e->capacity) return TC_E_INVALID_ARG;
    if (n == 0) return TC_E_OK;
    TC_HIP(e, hipSetDevice(e->device));
    std::vector<Cell> h(n);
    TC_HIP(e, hipMemcpyAsync(h.data(), e->cells + first, n * sizeof(Cell), hipMemcpyDeviceToHost, e->stream));
    TC_HIP(e, hipStreamSynchronize(e->stream));
    for (uint64_t i = 0; i < n; ++i) {
        if (tat) tat[i] = h[i].tat;
        if (expiry) expiry[i] = h[i].expiry;
    }
    return TC_E_OK;
}

extern "C" int tc_lookup_slot(tc_engine* e, const uint8_t* key, size_t key_len, int64_t* slot) {
    (void)key;
    (void)key_len;
    (void)slot;
    return fail(e, TC_E_UNSUPPORTED, "key mode not built yet");
}
