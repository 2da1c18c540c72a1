#!/usr/bin/env python3
"""bench.py -- GCRA decisions/s of the MI355X engine on BASELINE.json's workload.

    python bench.py [--gpus N --steps K --warmup W] [--workload uniform|zipf]

One "step" = one batch of 1 Mi requests (u32 slots already resident in HBM)
decided and applied against a 10 M-key resident store through the C ABI
(tc_rate_limit_batch_slots, device pointers, per-slot registered rate
parameters (100, 1000/3600 s), quantity 1, one timestamp per batch,
decisions-only output).  Duplicate keys inside a batch are honoured exactly.

N > 1 (torchrun, one rank per GPU; BASELINE configs[3]): ONE global request stream
over N x 10 M global key ids, N x 1 Mi requests per step (weak scaling).  Every rank
is handed the global batch, keeps the requests whose keys it owns with the device
partition kernel (tc_route_batch: owner and shard-local slot by a bijection of the
global id space, stable) and decides them on its own engine -- no collective on the
decision path, routing inside the timed region.  The per-GPU counter blocks are
all-gathered over RCCL every METRICS_EVERY steps and after the last one, the
per-GPU top-denied blocks once at the end, both inside the timed region (the only
exchange the path has: aggregate metrics).  The line carries `per_gpu` (requests
owned, allowed share, decisions/s): under Zipf the owner of the hottest key gets
11.6 % of ALL traffic on top of its share.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes streams onto 4 hardware queues by default.  The engine uses the main stream + 3
# grouping streams (+ 1 key stream); RCCL and torch add their own.  A main stream that shares a
# hardware queue with a grouping stream serialises the pipeline (measured 12 -> 5.7 G/s); the engine
# probes its side streams against the main stream and keeps only those that run concurrently, and
# asking the runtime for 8 queues before HIP initialises gives it more to choose from.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_KEYS = 10_000_000
BATCH = 1 << 20
ALG_BYTES_PER_DECISION = 36.125  # SURVEY.md section 8(d), slot mode, decisions only
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
METRICS_EVERY = 32               # N > 1: RCCL all-gather of the counter blocks every this many steps (+ the last)


KERNEL_OF_STAGE = {"prep": "rs::k_hist", "sort": "rs::k_onesweep", "eval": "ev::k_eval_sorted", "commit": "ev::k_commit_list",
                   "pack": "ev::k_pack_bits", "hash": "kt::k_probe + k_bind + k_follow", "bucket_hist": "bp::k_tile_hist",
                   "bucket_scan": "bp::k_bucket_scan", "bucket_scatter": "bp::k_scatter", "bucket_eval": "bp::k_bucket_eval"}
# kernel name fragments in the rocprofv3 summaries under profiles/
PROFILE_NAME_OF_STAGE = {"prep": "rs::k_hist", "sort": "k_onesweep", "eval": "k_eval_sorted", "commit": "k_commit_list",
                         "bucket_hist": "k_tile_hist", "bucket_scan": "k_bucket_scan", "bucket_scatter": "k_scatter",
                         "bucket_eval": "k_bucket_eval", "hash": "kt::k_probe"}
COPY_CEILING_GBS = 6290.0        # MI355X_MICROARCH.md: measured copy ceiling (what line-granular traffic can reach)


def pmc_traffic(stage, stream, layout, launches_per_batch):
    """HBM bytes per BATCH of the stage's kernels from the newest committed rocprofv3 PMC summary of this
    command (profiles/r<round>_v<visit>_<stream>_<layout>*_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE
    passes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as is).
    -> (bytes or None, source dict or None).  The numbers are NOT measured by this run: `source` says which
    file (and which commit it was taken at) so that a stale profile cannot pass for a fresh one."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", f"r*_{stream}_{layout}*_pmc.json")) + \
        glob.glob(os.path.join(ROOT, "profiles", f"r*_{stream}_pmc.json"))
    if layout == "wide":  # (round-1 summaries carry no layout tag: they are the 16-byte layout)
        files += [f for f in glob.glob(os.path.join(ROOT, "profiles", f"r*_{stream}_1M*_pmc.json"))]
    want = PROFILE_NAME_OF_STAGE.get(stage)
    if not files or not want:
        return None, None

    def visit(path):  # newest visit of the newest round ("v10" sorts after "v9")
        m = re.match(r"r(\d+)_v(\d+)_", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    for path in sorted(files, key=visit, reverse=True):
        doc = json.load(open(path))
        total, names = 0.0, []
        for name, v in doc["kernels"].items():
            # (launches that left at once -- the gated-off path of a batch enqueued on both -- carry a few KB)
            if want in name and v.get("launches", 0) >= 3 and 2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"] > 512.0:
                # several variants of one stage (first / later radix pass): weigh by launches
                total += (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0 * v["launches"]
                names.append((name, v["launches"]))
        if names:
            n_launch = sum(n for _, n in names)
            return total / n_launch * launches_per_batch, {"file": os.path.relpath(path, ROOT), "git_sha": doc.get("git_sha"),
                                                           "command": doc.get("command"), "kernels": [n for n, _ in names]}
    return None, None


def log(msg):
    """progress to stderr (stdout carries the one JSON line)"""
    if os.environ.get("TC_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--keys", type=int, default=N_KEYS)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--cpu-sample-batches", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--in-order", action="store_true", help="with --profile-run: batches without TC_B_INPUTS_READY (one stream)")
    ap.add_argument("--profile-run", action="store_true",
                    help="warmup + timed region only (what rocprofv3 is pointed at: no per-kernel events, no secondary runs)")
    ap.add_argument("--layout", default="fixed", choices=["wide", "fixed"],
                    help="resident state: 16-byte {tat, expiry} cells, or TC_CFG_FIXED_PARAMS (8-byte TAT column)")
    return ap.parse_args()


def make_batches(kind, n_keys, batch, count, seed_shift=0):
    from throttlecrab_amd import workload as W
    if kind == "zipf":
        z = W.Zipf(n_keys)
        return [z.slots(batch, seed=3 + seed_shift, start=i * batch) for i in range(count)]
    return [W.uniform_slots(n_keys, batch, seed=2 + seed_shift, start=i * batch) for i in range(count)]


def run_gpu(eng, d_batches, out, now0, steps, warmup, dist, cnt_view, gathered, want=("allowed",), piped=True):
    """warmup + timed region; returns seconds for `steps` batches (max over ranks)."""
    import torch
    it = 0

    def one(i, last=False):
        eng.rate_limit_batch_slots(d_batches[i % len(d_batches)], registered=True, quantity=1,
                                   now_ns=now0 + i * 1_000_000, want=want, out=out, inputs_ready=piped)
        if dist is not None and (i % METRICS_EVERY == METRICS_EVERY - 1 or last):
            eng.counters_refresh()
            dist.all_gather_into_tensor(gathered, cnt_view)

    for _ in range(warmup):
        one(it)
        it += 1
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        one(it, last=(k == steps - 1))
        it += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, it


def stage_profile(eng, d_batches, out, now0, steps, it0, piped=True):
    """`steps` more batches with a HIP event pair around each of the engine's kernels (recorded on the stream the
    kernel runs on).  piped=True: issued exactly like the timed region (the grouping of later batches overlaps
    the evaluation of earlier ones), so the durations include that contention.
    -> {stage: {"kernel", "launches_per_batch", "avg_ms", "per_batch_ms"}}"""
    import torch
    eng.profile_enable(True)
    for i in range(steps):
        eng.rate_limit_batch_slots(d_batches[(it0 + i) % len(d_batches)], registered=True, quantity=1,
                                   now_ns=now0 + (it0 + i) * 1_000_000, want=("allowed",), out=out, inputs_ready=piped)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    return {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps, "avg_ms": ms / calls, "per_batch_ms": ms / steps}
            for k, (ms, calls) in prof.items() if calls}


def roofline_block(a, stream, dt, piped, inorder):
    """achieved = algorithmic bytes (36.125 B per decision x the batch) / duration, against the
    8 TB/s HBM peak.  The headline entry is the stage with the largest PER-BATCH total in the configuration that
    was timed (pipelined: `value` comes from there); `stages` has every stage, in that configuration and with
    the batches strictly in order on one stream (what rocprofv3 --kernel-trace shows: the tracer serialises)."""
    alg = ALG_BYTES_PER_DECISION * a.batch

    def fill(st, k):
        s = dict(st)
        s["achieved_GBs"] = alg / (s["per_batch_ms"] * 1e-3) / 1e9
        s["frac"] = s["achieved_GBs"] / HBM_PEAK_GBS
        tr, src = pmc_traffic(k, stream, a.layout, s["launches_per_batch"])
        s["traffic_bytes_per_batch"] = tr
        if tr:
            s["traffic_floor_ms"] = tr / (COPY_CEILING_GBS * 1e9) * 1e3  # what this traffic costs at the copy ceiling
            s["traffic_source"] = src
        return s
    p_st = {k: fill(v, k) for k, v in piped.items()}
    i_st = {k: fill(v, k) for k, v in inorder.items()}
    if "bucket_eval" in i_st:
        # in order the batch took the bucket path: the sort path's kernels were launched behind the gate and left at
        # once (what is timed is the launch), they move no data
        for k in ("prep", "sort", "eval"):
            if k in i_st:
                for f in ("traffic_floor_ms", "traffic_source", "achieved_GBs", "frac"):
                    i_st[k].pop(f, None)
                i_st[k]["traffic_bytes_per_batch"] = None
                i_st[k]["note"] = "gated off: launched behind the bucket path's gate and left at once"
    dom = max(p_st, key=lambda k: p_st[k]["per_batch_ms"])
    d = p_st[dom]
    # the kernel that touches the resident state (what the algorithmic bytes describe), in both configurations
    ev_p = p_st.get("eval") or p_st.get("bucket_eval")
    ev_i = max((i_st[k] for k in ("eval", "bucket_eval") if k in i_st), key=lambda s: s["per_batch_ms"])
    block = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_batch": alg,
             "stage": dom, "kernel": d["kernel"], "launches_per_batch": d["launches_per_batch"], "avg_ms": d["avg_ms"],
             "per_batch_ms": d["per_batch_ms"], "achieved": d["achieved_GBs"], "frac": d["frac"],
             "traffic": d["traffic_bytes_per_batch"], "traffic_source": d.get("traffic_source"),
             "measured_in": "the configuration of the timed region (batches pipelined with TC_B_INPUTS_READY; HIP events on the "
                            "stream each kernel runs on, same process, right after the timed region)",
             "whole_step_frac": alg / (dt / a.steps) / 1e9 / HBM_PEAK_GBS,
             "whole_step_GBs": alg / (dt / a.steps) / 1e9,
             "evaluation_kernel": {"pipelined": ev_p, "in_order": ev_i},
             "stages": {"pipelined": p_st, "in_order": i_st},
             "in_order_sum_ms": sum(s["per_batch_ms"] for s in i_st.values())}
    return block


def measure_stream(a, t, W, stream, dev, local, rank, seed_shift, dist, world):
    """One engine, one request stream: warmup + timed region (pipelined), then the per-kernel profile."""
    import torch
    eng = t.Engine(a.keys, a.batch, device=local, fixed_params=(a.layout == "fixed"))
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    nb = a.steps + a.warmup
    host_batches = make_batches(stream, a.keys, a.batch, min(nb, 64), seed_shift=seed_shift)
    d_batches = [torch.from_numpy(b.astype(np.int32)).to(dev) for b in host_batches]
    out = t.BatchResult()
    cnt_view = gathered = None
    if dist is not None:
        from throttlecrab_amd.sharded import device_counter_view
        cnt_view = device_counter_view(eng)
        gathered = torch.zeros(world * cnt_view.numel(), dtype=torch.int64, device=dev)
    dt, it = run_gpu(eng, d_batches, out, W.T0_NS, a.steps, a.warmup, dist, cnt_view, gathered, piped=not a.in_order)
    c = eng.counters()
    res = {"value": a.steps * a.batch * world / dt, "unit": "decisions/s", "ms_per_step": 1e3 * dt / a.steps,
           "allowed_fraction": c["allowed"] / max(1, c["total"])}
    if rank == 0 and not a.profile_run:
        piped = stage_profile(eng, d_batches, out, W.T0_NS, a.steps, it, piped=True)
        inorder = stage_profile(eng, d_batches, out, W.T0_NS, a.steps, it + a.steps, piped=False)
        res["roofline"] = roofline_block(a, stream, dt, piped, inorder)
    return res, eng, d_batches, dt


def keys_bench(a, dev):
    """BASELINE configs[4] at its stated configuration (SURVEY.md section 8(d) cfg 5; workload.Config4Stream, the
    stream tests/test_gpu_keys_spec.py checks against the oracle): 10 batches of new keys (10.5 M keys, untimed),
    then mixed batches of 1 Mi requests -- 70 % hits of live keys (a fifth of them on 1000 hot keys), 20 % new keys,
    10 % re-hits of expired keys --, rate (10, 100 / 60 s), one second per batch, an expiry sweep every 4 batches
    INSIDE the timed region (the first one unbinds ~9 M keys: table rebuild, overflow compaction).  Both key
    sets: `key_%d` (5-12 B, confirmed inside the 32-byte table entry) and 32..64 B random ASCII (64-byte key
    records; over 48 bytes: the overflow arena).  Key arenas resident in HBM, batches pipelined."""
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W
    B, n_prefill = a.batch, 10
    steps = max(4, min(a.steps, 16) // 4 * 4)
    out = {}
    for label, long in (("key_%d", False), ("ascii_32_64", True)):
        log(f"  key set {label}")
        st = W.Config4Stream(B, n_prefill=n_prefill, long=long, sweep_every=4)
        cap = B * n_prefill + B
        eng = t.Engine(cap, B, device=dev.index or 0, key_mode=True, key_arena_bytes=(448 << 20) if long else 0)
        eng.use_torch_stream()
        res = t.BatchResult()
        b, c, p = W.Config4Stream.PARAMS

        def one(arena, now, piped=os.environ.get("TC_BENCH_KEYS_IN_ORDER") != "1"):
            eng.rate_limit_batch_keys(arena[0], arena[1], max_burst=b, count_per_period=c, period=p, quantity=1, now_ns=now,
                                      want=("allowed",), out=res, inputs_ready=piped)
        for k in range(n_prefill):
            ids, now = st.prefill(k)
            arena = st.keys(ids, device=dev)
            one(arena, now)
            torch.cuda.synchronize()
            del arena
        mixed = []
        key_bytes = 0
        for s_ in range(2 * steps):  # timed steps + the same number for the per-kernel profile
            ids, now = st.mixed(s_)
            arena = st.keys(ids, device=dev)
            key_bytes += int(arena[0].numel())
            mixed.append((arena, now))
        torch.cuda.synchronize()
        c0 = eng.counters()
        t0 = time.perf_counter()
        for s_ in range(steps):
            one(*mixed[s_])
            if st.sweep_due(s_):
                eng.sweep_expired_async(mixed[s_][1])  # enqueued behind the batch, no host wait
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = eng.counters()
        mean_len = key_bytes / (2 * steps * B)
        alg = (ALG_BYTES_PER_DECISION + 12 + 2 * mean_len) * B  # + offset 4, table probe 8, key bytes read twice
        r = {"value": steps * B / dt, "unit": "decisions/s", "steps": steps, "ms_per_step": 1e3 * dt / steps,
             "mean_key_bytes": mean_len, "keys_inserted_before": c0["keys_inserted"], "keys_inserted": c1["keys_inserted"] - c0["keys_inserted"],
             "swept": c1["swept"] - c0["swept"], "sweeps": steps // 4,
             "allowed_fraction": (c1["allowed"] - c0["allowed"]) / max(1, c1["total"] - c0["total"])}
        # per-kernel times of the same stream, pipelined as in the timed region (sweeps left out: not a per-batch stage)
        eng.profile_enable(True)
        for s_ in range(steps, 2 * steps):
            one(*mixed[s_])
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile_enable(False)
        stages = {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps, "avg_ms": ms / calls, "per_batch_ms": ms / steps}
                  for k, (ms, calls) in prof.items() if calls}
        dom = max(stages, key=lambda k: stages[k]["per_batch_ms"])
        d = stages[dom]
        tr, src = pmc_traffic(dom, "string_keys" + ("_long" if long else ""), "wide", d["launches_per_batch"])
        ach = alg / (d["per_batch_ms"] * 1e-3) / 1e9
        r["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_batch": alg,
                         "algorithmic_bytes_per_decision": alg / B, "stage": dom, "kernel": d["kernel"],
                         "launches_per_batch": d["launches_per_batch"], "avg_ms": d["avg_ms"], "per_batch_ms": d["per_batch_ms"],
                         "achieved": ach, "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src,
                         "whole_step_frac": alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, "stages": {"pipelined": stages},
                         "note": "the timed region also holds the sweeps (" + str(steps // 4) + "), the profile steps do not"}
        out[label] = r
        eng.close()
        del mixed
        torch.cuda.empty_cache()
    out["workload"] = ("10 x 1 Mi new keys, then 1 Mi-request batches: 70 % hits (1/5 of them on 1000 hot keys), 20 % new keys, "
                       "10 % re-hits of expired keys; rate (10,100/60 s); 1 s per batch; sweep every 4 batches")
    return out


def cpu_baseline(kind, n_keys, batch, n_batches):
    """The oracle (a port of RateLimiter<AdaptiveStore>, string keys "key_<slot>") timed on this box's host cores:
    (i) one thread over the first n_batches of the same stream == the reference's one actor task
    (throttlecrab-server/src/actor.rs:217-236); (ii) every core, one store per thread, keys routed by hash
    (what README.md:247-249 recommends); (iii) the reference's OWN benchmark loop (store_comparison.rs: 2 000 keys,
    400 000 calls, key formatting and clock read inside), next to the number the reference publishes for it."""
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    slots = np.concatenate(make_batches(kind, n_keys, batch, n_batches))
    kb, ko = O.format_keys(slots)
    now = (W.T0_NS + (np.arange(slots.size, dtype=np.int64) // batch) * 1_000_000)
    b, c, p = W.REF_PARAMS
    # store sized like the server would for this key count, server-default max_operations (config.rs:301)
    st = O.AdaptiveOracle(capacity=n_keys, created_ns=W.T0_NS, max_operations=1_000_000)
    t0 = time.perf_counter()
    st.batch_keys(kb, ko, b, c, p, 1, now)
    t1 = time.perf_counter() - t0
    del st
    # the cores this process may actually use: its CPU set, and the container's CPU-time quota if it has one
    ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    if quota:
        ncores = max(1, min(ncores, int(quota + 0.5)))
    tm, _ = O.batch_keys_mt(ncores, max(1000, n_keys // ncores), W.T0_NS, kb, ko, b, c, p, 1, now)
    O.reference_shape(2000, 50_000)  # warm
    ts, allowed, blocked = min(O.reference_shape(2000, 400_000) for _ in range(3))
    return {"value": slots.size / t1, "unit": "decisions/s", "cores": 1, "kind": "port",
            "sample": f"first {n_batches} batches ({slots.size} requests) of the same {kind} stream, "
                      f"string keys key_<slot>, AdaptiveStore port, 1 thread",
            "all_cores": {"value": slots.size / tm, "cores": ncores, "machine_cores": os.cpu_count(), "cgroup_cpu_quota": quota, "speedup_over_one_thread": t1 / tm,
                          "note": "one AdaptiveStore port per thread, keys routed by hash (routing inside the timing)"},
            "reference_shape": {"value": 400_000 / ts, "unit": "decisions/s", "cores": 1, "allowed": allowed, "blocked": blocked,
                                "sample": "throttlecrab-server/examples/store_comparison.rs: 400 000 rate_limit calls over 2 000 keys key_<i>, "
                                          "(100, 1000 / 3600 s), key formatting and clock read inside the loop, best of 3",
                                "published_by_reference": {"value": 12_500_000, "unit": "req/s", "hardware": "Apple M3 Max",
                                                           "source": "docs/benchmark-results.md:26-30 (AdaptiveStore)"}}}


def secondary(a, t, W, eng2, ob, d_batches, dev, local, other):
    """Other output forms and batch shapes on the second engine (its stream `ob` is the `other` one)."""
    import torch
    also = {}
    out = t.BatchResult()
    nb = a.steps + a.warmup
    full = t.BatchResult()
    dt3, _ = run_gpu(eng2, ob, full, W.T0_NS + 10**9, a.steps, 2, None, None, None,
                     want=t.Engine.ALL_FIELDS)
    also[f"{other}_stream_full_result"] = {"value": a.steps * a.batch / dt3, "unit": "decisions/s"}
    log("  records")
    rec = t.BatchResult()
    dt4, _ = run_gpu(eng2, ob, rec, W.T0_NS + 3 * 10**9, a.steps, 2, None, None, None,
                     want=t.Engine.RECORD_FIELDS)
    also[f"{other}_stream_full_result_records"] = {"value": a.steps * a.batch / dt4, "unit": "decisions/s",
                                                  "note": "result4: one 32-byte RateLimitResult record per request"}
    dec = t.BatchResult()
    dt5, _ = run_gpu(eng2, ob, dec, W.T0_NS + 3 * 10**9 + 10**8, a.steps, 2, None, None, None,
                     want=t.Engine.DECISION_FIELDS)
    also[f"{other}_stream_full_result_decision_records"] = {
        "value": a.steps * a.batch / dt5, "unit": "decisions/s",
        "note": "tc_decision: remaining, reset_after, retry_after, allowed, status in one 32-byte record"}
    # grouped output (TC_B_GROUPED_OUTPUT): rows in evaluation order + the request index of each row
    log("  grouped")
    grp = t.BatchResult()
    d_main = d_batches
    for label, streams, want in ((f"{a.workload}_stream_grouped_output", d_main, ("allowed",)),
                                 (f"{a.workload}_stream_grouped_bits", d_main, ("allowed_bits",)),
                                 (f"{a.workload}_stream_grouped_decision_records", d_main, t.Engine.DECISION_FIELDS)):
        for i in range(a.warmup):
            eng2.rate_limit_batch_slots(streams[i % len(streams)], registered=True, quantity=1, now_ns=W.T0_NS + 5 * 10**9 + i,
                                        want=want, out=grp, inputs_ready=True, grouped=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            eng2.rate_limit_batch_slots(streams[i % len(streams)], registered=True, quantity=1,
                                        now_ns=W.T0_NS + 5 * 10**9 + 10**6 * (i + 1), want=want, out=grp, inputs_ready=True,
                                        grouped=True)
        torch.cuda.synchronize()
        also[label] = {"value": a.steps * a.batch / (time.perf_counter() - t0), "unit": "decisions/s",
                       "note": "output rows in the engine's evaluation order + order[] (request index of each row)" +
                               ("; the decisions as one bit per row, packed by the evaluation's own wave ballots (no byte column)"
                                if want == ("allowed_bits",) else "")}
        grp = t.BatchResult()
    # general batches: every request carries its own timestamp (strictly increasing inside
    # the batch), so the closed form does not apply and k_eval_general runs
    log("  per-request timestamps")
    nows = [torch.arange(a.batch, dtype=torch.int64, device=dev) + (W.T0_NS + 4 * 10**9 + b * 10**6)
            for b in range(a.warmup + a.steps)]
    gout = t.BatchResult()
    for label, streams in ((f"{other}_stream_per_request_timestamps", ob), (f"{a.workload}_stream_per_request_timestamps", d_batches)):
        eng3 = t.Engine(a.keys, a.batch, device=local, fixed_params=(a.layout == "fixed"))
        eng3.use_torch_stream()
        eng3.register_params_uniform(*W.REF_PARAMS)
        for i in range(a.warmup):
            eng3.rate_limit_batch_slots(streams[i % len(streams)], registered=True, quantity=1, now_ns=nows[i],
                                        want=("allowed",), out=gout, inputs_ready=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.warmup, a.warmup + a.steps):
            eng3.rate_limit_batch_slots(streams[i % len(streams)], registered=True, quantity=1, now_ns=nows[i],
                                        want=("allowed",), out=gout, inputs_ready=True)
        torch.cuda.synchronize()
        also[label] = {"value": a.steps * a.batch / (time.perf_counter() - t0), "unit": "decisions/s"}
        eng3.close()
    log("  host buffers")
    # PCIe-inclusive rate: the same stream handed over as HOST buffers (never `value`)
    hb = make_batches(other, a.keys, a.batch, 8)
    hout = t.BatchResult()
    for i in range(2):
        eng2.rate_limit_batch_slots(hb[i], registered=True, quantity=1, now_ns=W.T0_NS + 2 * 10**9, want=("allowed",), out=hout)
    t0 = time.perf_counter()
    for i in range(8):
        eng2.rate_limit_batch_slots(hb[i], registered=True, quantity=1, now_ns=W.T0_NS + 2 * 10**9 + i, want=("allowed",), out=hout)
    also[f"{other}_stream_host_buffers_pcie_inclusive"] = {"value": 8 * a.batch / (time.perf_counter() - t0),
                                                          "unit": "decisions/s"}
    # the same, as a server would feed it: TC_B_ASYNC batches from a ring of pinned buffers, so the PCIe
    # transfers of one batch overlap the evaluation of others (never `value` either)
    K, NA = 4, 48
    ring = [(eng2.host_alloc(a.batch, np.uint32), t.BatchResult(allowed=eng2.host_alloc(a.batch, np.uint8))) for _ in range(K)]

    def feed(i):
        if i >= K:
            eng2.wait_batches(K - 1)   # the oldest set's results are in: a server would answer them here
        sl, ob = ring[i % K]
        eng2.rate_limit_batch_slots(sl, registered=True, quantity=1, now_ns=W.T0_NS + 3 * 10**9 + i, want=("allowed",),
                                    out=ob, async_=True)
    for i in range(K):
        ring[i][0][:] = hb[i % len(hb)]  # (producing the requests is the server's work, not the engine's)
    for i in range(K):
        feed(i)
    eng2.wait_batches(0)
    t0 = time.perf_counter()
    for i in range(NA):
        feed(i)
    eng2.wait_batches(0)
    also[f"{other}_stream_host_buffers_pinned_async_pcie_inclusive"] = {
        "value": NA * a.batch / (time.perf_counter() - t0), "unit": "decisions/s",
        "note": f"TC_B_ASYNC, ring of {K} pinned buffer sets (slots in, decisions out over PCIe every batch)"}
    del ring
    return also


def run_sharded(a, t, W, dev, local, rank, world, dist):
    """The N > 1 path (also taken with TC_BENCH_FORCE_DIST=1 on one GPU)."""
    import torch
    from throttlecrab_amd import sharded
    B = a.batch
    G = world * B                       # requests per global batch
    cap_batch = min(G, 4 * B)           # an owner's share of a global batch, evaluated in chunks of at most this
    eng = t.Engine(a.keys, cap_batch, device=local, fixed_params=(a.layout == "fixed"), track_denied=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    nb = a.steps + a.warmup
    n_distinct = min(nb, 8)
    if a.workload == "zipf":
        z = W.Zipf(world * a.keys)
        host = [z.slots(G, seed=3, start=i * G) for i in range(n_distinct)]   # the same stream on every rank
    else:
        host = [W.uniform_slots(world * a.keys, G, seed=2, start=i * G) for i in range(n_distinct)]
    d_global = [torch.from_numpy(h.astype(np.int32)).to(dev) for h in host]
    RING = 8  # routed slot columns stay untouched while up to pipeline-depth batches are in flight
    ring = [(torch.empty(G, dtype=torch.int32, device=dev), None, torch.zeros(world, dtype=torch.int32, device=dev)) for _ in range(RING)]
    counts_host = [eng.host_alloc(world + 1, np.uint32) for _ in range(RING)]   # pinned: counts, then the batch's tag
    for c in counts_host:
        c[:] = 0
    out = t.BatchResult()
    cnt_view = sharded.device_counter_view(eng)
    gathered = torch.zeros(world * cnt_view.numel(), dtype=torch.int64, device=dev)
    top_gathered = torch.zeros(world * sharded.TOPK, 2, dtype=torch.int64, device=dev)
    decided = 0

    # TC_ROUTE_AHEAD: the router runs on the engine's grouping streams, beside the evaluations of earlier batches (it
    # only reads the global batch and writes a ring entry whose last reader is already on the engine's stream), and
    # its last block writes the counts + the batch's tag into pinned host memory: the host learns how many requests
    # it owns by polling that word -- no event, no stream synchronisation on the way.
    def route(i):
        r = i % RING
        eng.route_batch(d_global[i % n_distinct], world, only=rank, out=ring[r], ahead=True, host_counts=counts_host[r], tag=i + 1)

    def evaluate(i, last=False, metrics=True):
        nonlocal decided
        r = i % RING
        while int(counts_host[r][world]) != i + 1:   # routed LOOKAHEAD steps ago: no wait in steady state
            pass
        mine = int(counts_host[r][rank])
        for lo in range(0, mine, cap_batch):
            hi = min(mine, lo + cap_batch)
            eng.rate_limit_batch_slots(ring[r][0][lo:hi], registered=True, quantity=1, now_ns=W.T0_NS + i * 1_000_000,
                                       want=("allowed",), out=out, inputs_ready=True)
        decided += mine
        if not metrics:
            return
        if i % METRICS_EVERY == METRICS_EVERY - 1 or last:
            eng.counters_refresh()
            dist.all_gather_into_tensor(gathered, cnt_view)
        if last:  # the optional part of the metrics payload: every shard's most denied keys as (global id, count)
            block = torch.from_numpy(sharded.pack_top_denied(eng.top_denied(sharded.TOPK), rank, world, a.keys)).to(dev)
            dist.all_gather_into_tensor(top_gathered, block)

    # The router runs LOOKAHEAD global batches ahead of the evaluation: the host needs a batch's count (how many of
    # its requests this rank owns) before it can enqueue the evaluation, and that read must not drain the stream.
    LOOKAHEAD = 4
    it = 0
    for j in range(LOOKAHEAD):
        route(j)
    for _ in range(a.warmup):
        route(it + LOOKAHEAD)
        evaluate(it)
        it += 1
    dist.barrier()
    torch.cuda.synchronize()
    decided = 0
    t0 = time.perf_counter()
    for k in range(a.steps):
        route(it + LOOKAHEAD)   # (the last LOOKAHEAD of them are routed for nothing: inside the timing, against us)
        evaluate(it, last=(k == a.steps - 1))
        it += 1
    torch.cuda.synchronize()
    dist.barrier()
    dt_mine = time.perf_counter() - t0
    tm = torch.tensor([dt_mine], dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dt = float(tm.item())
    c = eng.counters()
    mine = torch.tensor([decided, c["allowed"], c["denied"]], dtype=torch.int64, device=dev)
    everyone = torch.zeros(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(everyone, mine)
    ev = everyone.view(world, 3).cpu().numpy()
    total_decided = int(ev[:, 0].sum())
    assert total_decided == a.steps * G, (total_decided, a.steps * G)   # every request of the global stream has one owner
    per_gpu = [{"rank": r, "decisions": int(ev[r, 0]), "share_of_traffic": float(ev[r, 0]) / max(1, total_decided),
                "decisions_per_s": float(ev[r, 0]) / dt,
                "allowed_fraction_since_start": float(ev[r, 1]) / max(1, int(ev[r, 1] + ev[r, 2]))} for r in range(world)]
    res = {"value": total_decided / dt, "ms_per_step": 1e3 * dt / a.steps, "per_gpu": per_gpu,
           "imbalance_max_over_mean": float(ev[:, 0].max()) / max(1.0, float(ev[:, 0].mean())),
           "allowed_fraction": float(ev[:, 1].sum()) / max(1, int(ev[:, 1].sum() + ev[:, 2].sum())),
           "metrics_exchange": {"counter_block_bytes_per_gpu": 8 * int(cnt_view.numel()), "every_steps": METRICS_EVERY,
                                "top_denied_block_bytes_per_gpu": 16 * sharded.TOPK, "top_denied_exchanges": 1,
                                "top_denied_global": sharded.merge_top_denied(top_gathered.cpu().numpy(), 5)}}
    # roofline of this rank's evaluation (same kernels as the N = 1 run, fed by the router): HIP events per kernel
    steps_p = min(a.steps, 20)
    decided = 0
    eng.profile_enable(True)
    for _ in range(steps_p):
        route(it + LOOKAHEAD)
        evaluate(it, metrics=False)
        it += 1
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    stages = {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps_p, "avg_ms": ms / calls, "per_batch_ms": ms / steps_p}
              for k, (ms, calls) in prof.items() if calls}
    if stages:
        alg = ALG_BYTES_PER_DECISION * decided / steps_p
        dom = max(stages, key=lambda k: stages[k]["per_batch_ms"])
        ach = alg / (stages[dom]["per_batch_ms"] * 1e-3) / 1e9
        res["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "rank": rank, "algorithmic_bytes_per_batch": alg,
                           "stage": dom, "kernel": stages[dom]["kernel"], "launches_per_batch": stages[dom]["launches_per_batch"],
                           "avg_ms": stages[dom]["avg_ms"], "per_batch_ms": stages[dom]["per_batch_ms"], "achieved": ach,
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "stages": {"pipelined": stages},
                           "whole_step_frac": ALG_BYTES_PER_DECISION * G / (dt / a.steps) / 1e9 / (HBM_PEAK_GBS * world)}
    eng.close()
    return res


def main():
    a = parse()
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # TC_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, RCCL all-gather of the counter
    # blocks, max-over-ranks timing) even with one rank -- the only way to exercise it on a 1-GPU box
    if world > 1 or os.environ.get("TC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device(f"cuda:{local}")

    if dist is not None:
        # a real stream for the engine AND torch / RCCL (torch's default stream has handle 0, which the engine reads as
        # "use your own stream": the metrics all-gather would then not be ordered behind the counters' refresh)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            sh = run_sharded(a, t, W, dev, local, rank, world, dist)
            torch.cuda.synchronize()
        if rank == 0:
            print(json.dumps({
                "metric": "GCRA decisions/sec, 10M keys per GPU", "value": sh["value"], "unit": "decisions/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": sh["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                "config": {"workload": f"configs[3]: {world} x {a.keys} keys hash-sharded across {world} GPU(s), ONE global {a.workload} "
                                       f"request stream of {world} x {a.batch} requests per step routed on the device (tc_route_batch), "
                                       f"params (100,1000/3600s), q=1; RCCL all-gather of the metrics only",
                           "keys_per_gpu": a.keys, "global_batch": world * a.batch, "stream": a.workload,
                           "parallelism": f"hash-shard x{world}", "outputs": "allowed u8 (decisions only)",
                           "resident_state": a.layout, "metrics_allgather_every": METRICS_EVERY},
                "allowed_fraction": sh["allowed_fraction"], "per_gpu": sh["per_gpu"],
                "imbalance_max_over_mean": sh["imbalance_max_over_mean"], "metrics_exchange": sh["metrics_exchange"],
                "roofline": sh.get("roofline"), "cpu_baseline": None,
                "note": "cpu_baseline is reported by the N = 1 run; roofline here is rank 0's evaluation of the requests it owns "
                        "(the router's three small kernels over the global batch are not in it)"}))
        dist.barrier()
        dist.destroy_process_group()
        return

    log(f"headline: {a.workload} / {a.layout}")
    main_res, eng, d_batches, dt = measure_stream(a, t, W, a.workload, dev, local, rank, 0, None, 1)
    result = {
        "metric": "GCRA decisions/sec, 10M keys", "value": main_res["value"], "unit": "decisions/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": f"configs[{1 if a.workload == 'uniform' else 2}]: {a.keys} pre-hashed keys SoA per GPU, "
                               f"{a.workload} request stream, batch={a.batch}, params (100,1000/3600s), q=1",
                   "keys_per_gpu": a.keys, "batch": a.batch, "stream": a.workload,
                   "resident_state": ("TC_CFG_FIXED_PARAMS: TAT column, 8 B per key + the plan dictionary (emission interval, "
                                      "tolerance, burst capacity per plan)" if a.layout == "fixed" else
                                      "{tat, expiry} cell, 16 B per key + plan id column + the plan dictionary"),
                   "parallelism": f"hash-shard x{world}", "outputs": "allowed u8 (decisions only)",
                   "pipelining": "TC_B_INPUTS_READY: batch k+1.. grouped on auxiliary streams while batch k is evaluated",
                   "metrics_allgather_every": METRICS_EVERY if world > 1 else None},
        "allowed_fraction": main_res["allowed_fraction"],
    }
    if "roofline" in main_res:
        result["roofline"] = main_res["roofline"]

    if rank == 0:
        if not a.no_also and world == 1 and not a.profile_run:
            other = "zipf" if a.workload == "uniform" else "uniform"
            # the other BASELINE stream (configs[2]: Zipf s = 1.1, north_star's target stream), measured the same way
            log(f"other stream: {other}")
            o_res, eng2, ob, _ = measure_stream(a, t, W, other, dev, local, 0, 0, None, 1)
            o_res["config"] = f"configs[{2 if other == 'zipf' else 1}]: same engine shape, {other} request stream"
            result[f"{other}_stream"] = o_res
            # the headline stream on the other resident-state layout
            a2 = argparse.Namespace(**vars(a))
            a2.layout = "fixed" if a.layout == "wide" else "wide"
            log(f"other layout: {a2.layout}")
            l_res, eng_l, _, _ = measure_stream(a2, t, W, a.workload, dev, local, 0, 0, None, 1)
            eng_l.close()
            l_res.pop("roofline", None)
            result[f"{a.workload}_stream_{a2.layout}_layout"] = l_res
            log("secondary output forms")
            result["also"] = secondary(a, t, W, eng2, ob, d_batches, dev, local, other)
            eng2.close()
            log("string keys")
            result["also"]["string_keys_config4"] = keys_bench(a, dev)
        if not a.no_cpu and not a.profile_run:
            log("cpu baseline")
            result["cpu_baseline"] = cpu_baseline(a.workload, a.keys, a.batch, a.cpu_sample_batches)
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
