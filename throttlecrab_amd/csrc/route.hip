// route.hip -- multi-GPU: routing a global stream to the GPUs that own its keys
#include "engine.hpp"

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
    // One scratch lane: [ look-back words of k_route_one: ONE_PASS_TILES x u64 | tile_cnt: u32 [world][tiles] of the count / scan /
    // scatter routers and of k_route_one | ... FIRST HALF ends | the split router's tagged u64 words [world][tiles] | ticket: 2 ].
    // The split router's words are validated by the call's sequence number in their high half and never cleared, so no router
    // that writes plain u32 counts may ever touch them (they used to share tile_cnt's place, and a stale
    // odd-indexed count equal to a later call's small sequence number read as "this tile has published"): they have the
    // second half of the lane to themselves, at a place that does not move while the allocation stands.
    const size_t half_need = 2 * (size_t)rt::ONE_PASS_TILES + 2 * (size_t)tiles * r.world + 2;
    if (2 * half_need > e->route_ws_words) {
        if (e->route_ws) {
            TC_HIP(e, hipDeviceSynchronize()); // (whatever streams earlier routers ran on)
            (void)hipFree(e->route_ws);
            e->route_ws = nullptr;
            e->route_ws_words = 0;
        }
        // one scratch per grouping stream + one (lane 0) for every other stream a router may run on: routers on different
        // grouping streams never wait for one another, lane 0's users are ordered by an event
        const size_t each = 4 * half_need; // (room to grow: a reallocation drains the device)
        TC_HIP(e, hipMalloc(&e->route_ws, each * (1 + AUX_MAX) * sizeof(uint32_t)));
        // (hipMemset on device memory returns before the fill has run, and the fill runs on the NULL stream, which the
        // engine's non-blocking streams are not ordered behind: without the wait the fill could land on top of the
        // first router's tile counts -- seen as a 1-in-50 wrong partition on a fresh engine)
        TC_HIP(e, hipMemset(e->route_ws, 0, each * (1 + AUX_MAX) * sizeof(uint32_t)));
        TC_HIP(e, hipDeviceSynchronize());
        e->route_ws_words = each;
    }
    if (lane == 0) {
        // every router off the grouping streams shares scratch lane 0: two of them on DIFFERENT streams (a caller's stream and
        // the engine's, or two callers' streams) are ordered by an event instead of racing on the tile counts
        if (!e->route_l0_done) TC_HIP(e, hipEventCreateWithFlags(&e->route_l0_done, hipEventDisableTiming));
        if (e->route_l0_used && e->route_l0_stream != s) TC_HIP(e, hipStreamWaitEvent(s, e->route_l0_done, 0));
    }
    rt::Work w;
    uint32_t* scratch = e->route_ws + (size_t)lane * e->route_ws_words;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(scratch); // [ONE_PASS_TILES] look-back words of the one-pass router
    w.tile_cnt = scratch + 2 * (size_t)rt::ONE_PASS_TILES;
    unsigned long long* split_status = reinterpret_cast<unsigned long long*>(scratch + e->route_ws_words / 2); // (route_ws_words is a multiple of 4)
    w.totals = r.out_count;
    w.tiles = tiles;
    w.host_totals = r.out_count_host;
    w.tag = r.tag;
    // one destination: count, offset and scatter in ONE pass over the ids for grids of at most ONE_PASS_TILES (tiles of 4096
    // or 8192 ids); every destination, or a bigger batch: count | scan | scatter
    const uint32_t tiles32 = (n + 2 * rt::TILE - 1) / (2 * rt::TILE);
    bool published = false;
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
        // (the totals of every destination, and the host's copy + tag, in one launch)
        hipLaunchKernelGGL(rt::k_route_totals, dim3(r.world), dim3(rt::THREADS), 0, s, w, r.world, scratch + e->route_ws_words - 2);
        published = true;
    } else if (r.out_dst && r.only < 0 && tiles <= rt::ONE_PASS_TILES && !getenv("TCGPU_ROUTE_3PASS")) {
        // every destination into its own buffer (the exchange): one pass as well
        if (++e->route_seq == 0u) e->route_seq = 1u;
        hipLaunchKernelGGL(rt::k_route_split_one, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, split_status, e->route_seq, split,
                           e->counters + (TC_CNT_COUNT + 1) + 3);
    } else {
        hipLaunchKernelGGL(rt::k_route_count, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w);
        hipLaunchKernelGGL(rt::k_route_scan, dim3(r.world), dim3(rt::THREADS), 0, s, w);
        if (r.out_dst) hipLaunchKernelGGL(rt::k_route_scatter<true>, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (int)r.only, r.out_slot, r.out_pos, split);
        else hipLaunchKernelGGL(rt::k_route_scatter<false>, dim3(tiles), dim3(rt::THREADS), 0, s, r.global_id, n, m, w, (int)r.only, r.out_slot, r.out_pos, split);
    }
    if (r.out_count_host && !published) hipLaunchKernelGGL(rt::k_route_publish, dim3(1), dim3(rt::MAX_WORLD), 0, s, w, r.world);
    TC_HIP(e, hipGetLastError());
    if (lane == 0) {
        TC_HIP(e, hipEventRecord(e->route_l0_done, s));
        e->route_l0_stream = s;
        e->route_l0_used = true;
    }
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

// String keys (BASELINE configs[4] across GPUs; README.md:247-249 shards by KEY): owner(key) = mix64(hash(key) ^ salt) mod world,
// SURVEY.md section 8(e).  A front end (the actor, a RESP connection handler) splits a key batch by owner with this and hands
// every GPU's key-mode engine the keys it owns; the engines never see one another's keys.  The salt decorrelates the owner
// from the placement inside an engine's table (which uses the low bits of the same hash).  Host only: no device needed.
extern "C" int tc_route_keys_host(uint32_t world, uint64_t n, const uint8_t* key_bytes, const uint32_t* key_off, uint32_t* owner) {
    if (world == 0 || world > 64 || (n && (!key_bytes || !key_off || !owner))) return TC_E_INVALID_ARG;
    for (uint64_t i = 0; i < n; ++i) {
        if (key_off[i + 1] < key_off[i]) return TC_E_INVALID_ARG;
        const uint64_t h = kt::hash_key(key_bytes + key_off[i], key_off[i + 1] - key_off[i]);
        owner[i] = (uint32_t)(kt::mix64(h ^ 0xa5a5a5a5a5a5a5a5ull) % world);
    }
    return TC_E_OK;
}
