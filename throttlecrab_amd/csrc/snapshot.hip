// snapshot.hip -- tc_snapshot_save / tc_snapshot_load
#include "engine.hpp"

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
constexpr uint32_t SNAP_VERSION = 6; // 3: sharded tombstone count; 4: 128-byte key records, table of retired keys (denial counts); 5: 65 536 retired keys + statistics record; 6: entry positions as a column
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
        v.push_back({t.pos_col, (size_t)t.capacity * 4});
        if (e->retired) v.push_back({e->retired, ((size_t)kt::RETIRED_CAP + 1) * sizeof(kt::RetiredRec)});
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
    aside_reset(e);
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
