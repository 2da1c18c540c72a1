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

Output (rank 0): the LAST stdout line is ONE compact JSON object (< 4 KB: headline, roofline of the critical-path
kernel, cpu_baseline, one number per secondary workload); everything else (per-stage tables, PMC provenance, notes)
goes to gpurun_out/bench_detail.json.
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
ALG_BYTES_GENERAL = 44.125       # ... + the request's own 8-byte timestamp (general batches: per-request `now`)
OUT_RING = 8                     # result sets the pipelined runs cycle through (more than the engine keeps in flight)
COMPACT_LIMIT = 4096             # bytes: the driver keeps an 8 KB stdout tail; the compact line stays well inside it
DETAIL_PATH = os.path.join("gpurun_out", "bench_detail.json")
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
METRICS_EVERY = 32               # N > 1: RCCL all-gather of the counter blocks every this many steps (+ the last)


KERNEL_OF_STAGE = {"prep": "rs::k_hist", "sort": "rp::k_tile_part + rs::k_finish (range path) / rs::k_onesweep (LSD passes)", "eval": "ev::k_eval_sorted", "commit": "ev::k_commit_list",
                   "pack": "ev::k_pack_bits", "hash": "kt::k_probe + k_bind + k_follow", "bucket_hist": "bp::k_tile_hist",
                   "bucket_scan": "bp::k_bucket_scan", "bucket_scatter": "bp::k_scatter", "bucket_eval": "bp::k_bucket_eval"}
# kernel name fragments in the rocprofv3 summaries under profiles/
PROFILE_NAME_OF_STAGE = {"prep": "rs::k_hist", "sort": ("k_onesweep", "k_tile_part", "rs::k_finish"), "eval": ("k_eval_sorted", "k_eval_lean_hot"), "commit": "k_commit_list",
                         "bucket_hist": "k_tile_hist", "bucket_scan": "k_bucket_scan", "bucket_scatter": "k_scatter",
                         "bucket_eval": "k_bucket_eval", "hash": "kt::k_probe", "eval_general": "k_eval_general"}
COPY_CEILING_GBS = 6290.0        # MI355X_MICROARCH.md: measured copy ceiling (what line-granular traffic can reach)


def pmc_traffic(stage, stream, layout, launches_per_batch):
    """HBM bytes per BATCH of the stage's kernels from the newest committed rocprofv3 PMC summary of this
    command (profiles/r<round>_v<visit>_<stream>_<layout>*_pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE
    passes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as is).
    -> (bytes or None, source dict or None).  The numbers are NOT measured by this run: `source` says which
    file (and which commit it was taken at) so that a stale profile cannot pass for a fresh one."""
    import glob
    import re
    pat = re.compile(rf"r\d+_v\d+_{re.escape(stream)}(_{re.escape(layout)})?(_1M)?(_piped)?_pmc\.json$")
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if pat.match(os.path.basename(f))]
    if layout != "wide":  # (summaries without a layout tag are round 1's: the 16-byte layout)
        files = [f for f in files if f"_{layout}" in os.path.basename(f)]
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
            if any(w in name for w in ((want,) if isinstance(want, str) else want)) and v.get("launches", 0) >= 3 and 2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"] > 512.0:
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
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU).  Started plainly -- `python bench.py --gpus N`, no torchrun around it -- bench.py "
                         "launches the N ranks itself (torch.distributed.run on 127.0.0.1) and fails if the node has fewer GPUs; under "
                         "torchrun it must agree with WORLD_SIZE.  Default: WORLD_SIZE, else 1")
    ap.add_argument("--plan-only", action="store_true", help="print what --gpus N would launch, as JSON, and exit (no GPU touched)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="uniform", choices=["uniform", "zipf", "general", "general_zipf"],
                    help="uniform / zipf: one timestamp per batch (BASELINE configs[1] / [2]); general / general_zipf: the same "
                         "slot streams with a timestamp PER REQUEST, as every transport stamps them "
                         "(throttlecrab-server/src/transport/http.rs:128) -- k_eval_general")
    ap.add_argument("--route", default="replicate", choices=["exchange", "replicate"],
                    help="N > 1: replicate = every rank is handed the whole global batch and keeps what it owns (default: faster "
                         "on every configuration that could be measured, DESIGN.md section 9); exchange = every rank routes only its "
                         "1/N slice straight into the owners' inboxes (peer memory), per-GPU routing work independent of N")
    ap.add_argument("--keys", type=int, default=N_KEYS)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--cpu-sample-batches", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--in-order", action="store_true", help="with --profile-run: batches without TC_B_INPUTS_READY (one stream)")
    ap.add_argument("--profile-run", action="store_true",
                    help="warmup + timed region only (what rocprofv3 is pointed at: no per-kernel events, no secondary runs)")
    ap.add_argument("--plans", default="one", choices=["one", "tiers4", "tiers1000"],
                    help="rate plans of the headline run: one plan for the whole table (tc_register_params_uniform: BASELINE configs "
                         "1-2 as SURVEY.md states them) or a plan PER KEY, 4 / 1000 distinct (burst, count, period) tiers assigned by a "
                         "hash of the slot (tc_register_params: the evaluation reads rate_id[] and the plan dictionary)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle replay of the timed batches")
    ap.add_argument("--layout", default="fixed", choices=["wide", "fixed"],
                    help="resident state: 16-byte {tat, expiry} cells, or TC_CFG_FIXED_PARAMS (8-byte TAT column)")
    return ap.parse_args()


def make_batches(kind, n_keys, batch, count, seed_shift=0):
    from throttlecrab_amd import workload as W
    if kind == "zipf":
        z = W.Zipf(n_keys)
        return [z.slots(batch, seed=3 + seed_shift, start=i * batch) for i in range(count)]
    return [W.uniform_slots(n_keys, batch, seed=2 + seed_shift, start=i * batch) for i in range(count)]


def now_of(now0, i, nows):
    """the batch's timestamp: one scalar (1 ms per batch), or -- general batches -- a resident column of per-request
    timestamps (request j of batch i is stamped i ms + j ns: strictly increasing inside the batch)"""
    return now0 + i * 1_000_000 if nows is None else nows[i % len(nows)]


def make_nows(dev, batch, count, now0):
    import torch
    return [torch.arange(batch, dtype=torch.int64, device=dev) + (now0 + b * 1_000_000) for b in range(count)]


def plan_tiers(kind, n_keys):
    """Per-key rate plans: -> (tiers int64[T, 3] of (burst, count, period), tier_of uint16[n_keys]).  Tier 0 is the reference
    benchmark's plan (store_comparison.rs:17); the others vary burst 20..210, rate and period, all valid on the 8-byte
    layout (burst >= 2, emission interval and tolerance far below 2^60 ns).  A key's tier is a hash of its slot, so
    neighbouring slots carry unrelated plans (the worst case for the rate_id[] gather's locality)."""
    from throttlecrab_amd import workload as W
    T = {"tiers4": 4, "tiers1000": 1000}[kind]
    k = np.arange(T, dtype=np.int64)
    tiers = np.stack([20 + (k * 37) % 191, 100 + 13 * k, np.where(k % 3 == 0, 3600, np.where(k % 3 == 1, 60, 600))], axis=1).astype(np.int64)
    tiers[0] = W.REF_PARAMS
    tier_of = (W.splitmix64(np.arange(n_keys, dtype=np.uint64) ^ np.uint64(0x7157)) % np.uint64(T)).astype(np.uint16)
    return tiers, tier_of


def register_plans(eng, plans, n_keys):
    """one plan for the table, or a plan per key; -> per-slot [n_keys, 3] triples (None: one plan)"""
    from throttlecrab_amd import workload as W
    if plans in (None, "one"):
        eng.register_params_uniform(*W.REF_PARAMS)
        return None
    tiers, tier_of = plan_tiers(plans, n_keys)
    per_slot = tiers[tier_of]
    eng.register_params(per_slot[:, 0], per_slot[:, 1], per_slot[:, 2])
    return per_slot


def verify_run(snap, host_batches, n_nows, per_slot, n_keys, batch, general):
    """The checker of the timed run (the oracle is never timed here): replays EVERY batch the engine was given since its
    creation -- warmup and timed region, same slots, same timestamps -- through the oracle's dense store one request at a
    time (rate_limiter.rs:147-205 in queue order) and compares (i) the allowed / denied / error counters of the whole run,
    (ii) the decision bytes of the last batches (the ring of result arrays still holds them) and (iii) the whole resident
    state (tc_read_state over every key).  -> dict with "ok"."""
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    t0 = time.perf_counter()
    orc = O.DenseOracle(n_keys)
    th = O.host_threads()
    nb = snap["batches"]
    allowed = 0
    bytes_ok, bytes_checked = True, 0
    for i in range(nb):
        sl = host_batches[i % len(host_batches)]
        now = W.T0_NS + i * 1_000_000
        if general:
            now = W.T0_NS + (i % n_nows) * 1_000_000 + np.arange(batch, dtype=np.int64)
        if per_slot is None:
            ref = orc.batch_slots(sl, *W.REF_PARAMS, 1, now, threads=th)
        else:
            pr = per_slot[sl]
            ref = orc.batch_slots(sl, pr[:, 0], pr[:, 1], pr[:, 2], 1, now, threads=th)
        allowed += int(ref.allowed.sum())
        if i in snap["ring"]:
            bytes_checked += 1
            bytes_ok = bytes_ok and bool(np.array_equal(snap["ring"][i], ref.allowed))
    c = snap["counters"]
    counters_ok = (c["allowed"] == allowed and c["denied"] == nb * batch - allowed and c["errors"] == 0 and c["total"] == nb * batch)
    otat, oexp, occ = orc.dump()
    tat, exp = snap["state"]
    state_ok = bool(not exp[~occ].any() and np.array_equal(tat[occ], otat[occ]) and np.array_equal(exp[occ], oexp[occ]))
    return {"ok": bool(counters_ok and bytes_ok and state_ok and snap["selfcheck"] == 0 and bytes_checked > 0),
            "batches_replayed": nb, "counters": bool(counters_ok), "decision_bytes_of_last_batches": bool(bytes_ok),
            "batches_compared_bytewise": bytes_checked, "resident_state_all_keys": state_ok, "selfcheck": snap["selfcheck"],
            "oracle_allowed": allowed, "engine_allowed": c["allowed"], "seconds": time.perf_counter() - t0}


def run_gpu(eng, d_batches, out, now0, steps, warmup, dist, cnt_view, gathered, want=("allowed",), piped=True, nows=None, it0=0):
    """warmup + timed region; returns seconds for `steps` batches (max over ranks).  (With `nows` the columns repeat
    after len(nows) batches: timestamps then go back by that many ms once per cycle, which the general path takes
    like any other non-monotone clock.)"""
    import torch
    it = it0   # (batches already given to the engine: the stream and its clock go on from there)

    # every batch in flight writes its results to arrays of its own (a ring of OUT_RING result sets, as a consumer that
    # reads them later needs anyway); decisions-only pipelined batches say so with TC_B_OUTPUTS_IDLE
    ring = out if isinstance(out, list) else [out]
    idle = piped and len(ring) > 1

    def one(i, last=False):
        eng.rate_limit_batch_slots(d_batches[i % len(d_batches)], registered=True, quantity=1,
                                   now_ns=now_of(now0, i, nows), want=want, out=ring[i % len(ring)], inputs_ready=piped,
                                   outputs_idle=idle)
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


def stage_profile(eng, d_batches, out, now0, steps, it0, piped=True, nows=None):
    """`steps` more batches with a HIP event pair around each of the engine's kernels (recorded on the stream the
    kernel runs on).  piped=True: issued exactly like the timed region (the grouping of later batches overlaps
    the evaluation of earlier ones), so the durations include that contention.
    -> {stage: {"kernel", "launches_per_batch", "avg_ms", "per_batch_ms"}}"""
    import torch
    eng.profile_enable(True)
    ring = out if isinstance(out, list) else [out]
    for i in range(steps):
        eng.rate_limit_batch_slots(d_batches[(it0 + i) % len(d_batches)], registered=True, quantity=1,
                                   now_ns=now_of(now0, it0 + i, nows), want=("allowed",), out=ring[(it0 + i) % len(ring)],
                                   inputs_ready=piped, outputs_idle=piped and len(ring) > 1)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    return {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps, "avg_ms": ms / calls, "per_batch_ms": ms / steps}
            for k, (ms, calls) in prof.items() if calls}


def stage_table(stages, alg, stream, layout, gated=()):
    """Detail only (bench_detail.json): every stage with its algorithmic rate and the PMC traffic of the newest committed
    profile.  A stage that did not do a whole batch's work per batch (gated-off launches, < 1 launch per batch) gets
    no rate: algorithmic bytes divided by the time of a kernel that left at once is not a roofline figure."""
    out = {}
    for k, st in stages.items():
        s = dict(st)
        if k in gated or s["launches_per_batch"] < 0.999:
            s["note"] = "gated off or not launched for every batch: no rate"
        else:
            s["achieved_GBs"] = alg / (s["per_batch_ms"] * 1e-3) / 1e9
            tr, src = pmc_traffic(k, stream, layout, s["launches_per_batch"])
            s["traffic_bytes_per_batch"] = tr
            if tr:
                s["traffic_floor_ms"] = tr / (COPY_CEILING_GBS * 1e9) * 1e3  # what this traffic costs at the copy ceiling
                s["traffic_source"] = src
        out[k] = s
    return out


def roofline_entry(kernel, avg_ms, alg, ms_per_step, traffic, source):
    """The roofline object of the JSON line for ONE launch of the critical-path kernel.  achieved = algorithmic bytes
    of the launch / its average duration (HIP events on the stream it runs on, inside this process); frac must lie
    in (0, 1] and the kernel cannot take longer than the step it is part of -- otherwise the entry is withheld."""
    ach = alg / (avg_ms * 1e-3) / 1e9
    frac = ach / HBM_PEAK_GBS
    r = {"bound": "hbm", "kernel": kernel, "avg_ms": avg_ms, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
         "traffic": traffic, "traffic_over_algorithmic": (traffic / alg) if traffic else None,
         "whole_step_frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "source": source}
    if not (0.0 < frac <= 1.0):
        r.update({"achieved": None, "frac": None, "invalid": "frac outside (0, 1]"})
    elif avg_ms > ms_per_step * 1.02:
        # events on a stream that other streams contend with can stretch; a kernel "longer than its step" is not evidence
        r["invalid"] = "avg_ms > ms_per_step: per-kernel events stretched by concurrent streams"
    return r


def measure_stream(a, t, W, stream, dev, local, rank, seed_shift, dist, world, general=False, profile=True, plans=None):
    """One engine, one request stream: warmup + timed region (pipelined), the oracle replay of exactly those batches, then
    the per-kernel profile.  general: every request carries its own timestamp (k_eval_general instead of k_eval_sorted).
    plans: None / "one" = one registered plan for the table; "tiers4" / "tiers1000" = a plan per key (plan_tiers)."""
    import torch
    plans = a.plans if plans is None else plans
    eng = t.Engine(a.keys, a.batch, device=local, fixed_params=(a.layout == "fixed"))
    eng.use_torch_stream()
    per_slot = register_plans(eng, plans, a.keys)
    nb = a.steps + a.warmup
    host_batches = make_batches(stream, a.keys, a.batch, min(nb, 64), seed_shift=seed_shift)
    d_batches = [torch.from_numpy(b.astype(np.int32)).to(dev) for b in host_batches]
    nows = make_nows(dev, a.batch, min(nb + 2 * a.steps, 32), W.T0_NS) if general else None
    # result arrays allocated up front: with a warmup shorter than the ring, the first use of a ring entry (a device
    # allocation) would otherwise fall into the timed region (20 timed batches: 53 instead of 48 us per batch)
    out = [t.BatchResult(allowed=torch.empty(a.batch, dtype=torch.uint8, device=dev)) for _ in range(OUT_RING)]
    cnt_view = gathered = None
    if dist is not None:
        from throttlecrab_amd.sharded import device_counter_view
        cnt_view = device_counter_view(eng)
        gathered = torch.zeros(world * cnt_view.numel(), dtype=torch.int64, device=dev)
    dt, it = run_gpu(eng, d_batches, out, W.T0_NS, a.steps, a.warmup, dist, cnt_view, gathered, piped=not a.in_order, nows=nows)
    c = eng.counters()
    engine_info = eng.info()   # (did the pipelined batches really overlap -- side streams kept, probe verdicts, grouping path)
    ms = 1e3 * dt / a.steps
    alg = (ALG_BYTES_GENERAL if general else ALG_BYTES_PER_DECISION) * a.batch
    res = {"value": a.steps * a.batch * world / dt, "unit": "decisions/s", "ms_per_step": ms,
           "allowed_fraction": c["allowed"] / max(1, c["total"]), "whole_step_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "pipelining_degraded": bool(engine_info["pipelining_degraded"]), "engine_info": engine_info,
           "grouping_path": engine_info.get("grouping_path")}
    if plans not in (None, "one"):
        res["plans"] = plans
    if rank == 0 and dist is None and seed_shift == 0 and not a.no_verify and not a.profile_run and not a.in_order and nb <= 400:
        # what the timed run left behind, taken before anything else touches the engine: counters, the result arrays of the
        # last batches, the state of every key -- compared with the oracle's replay of the same nb batches
        snap = {"batches": nb, "counters": c, "selfcheck": eng.selfcheck(), "state": eng.read_state(0, a.keys),
                "ring": {i: out[i % OUT_RING].allowed.cpu().numpy()[:a.batch].copy() for i in range(max(0, nb - OUT_RING), nb)}}
        try:
            res["verified"] = verify_run(snap, host_batches, len(nows) if nows else 0, per_slot, a.keys, a.batch, general)
        except Exception as ex:  # noqa: BLE001 (the checker must not take the measurement down with it)
            res["verified"] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"[:160]}
        del snap
        log(f"  verified: {res['verified']}")
    if rank == 0 and dist is None and seed_shift == 0 and not a.profile_run and not a.in_order and os.environ.get("TC_BENCH_REPEATS", "1") != "0":
        # the timed region is 20 x 44 us = 0.9 ms -- four more regions of the same length right behind it (the
        # stream goes on), so that a 3 % difference between two rounds can be told from noise.  `value` is the FIRST region's.
        reps = [ms]
        for _ in range(4):
            dt_r, it = run_gpu(eng, d_batches, out, W.T0_NS, a.steps, 0, None, None, None, piped=True, nows=nows, it0=it)
            reps.append(1e3 * dt_r / a.steps)
        res["repeats"] = {"ms_per_step": reps, "median": float(np.median(reps)), "min": min(reps), "max": max(reps)}
        log("  five regions, ms per step: " + " ".join(f"{x:.4f}" for x in reps))
    if rank == 0 and profile and not a.profile_run:
        piped = stage_profile(eng, d_batches, out, W.T0_NS, a.steps, it, piped=True, nows=nows)
        inorder = stage_profile(eng, d_batches, out, W.T0_NS, a.steps, it + a.steps, piped=False, nows=nows)
        tag = ("general_" if general else "") + stream
        lay = a.layout + ("" if plans in (None, "one") else "_" + plans)   # (PMC summaries of per-key-plan runs carry the tier count)
        # The critical-path kernel of the pipelined run is the evaluation on the engine's stream (the grouping of later
        # batches runs beside it on the auxiliary streams): its average launch duration as timed in the pipelined
        # configuration is the roofline's avg_ms.
        ev = piped.get("eval") or piped.get("bucket_eval")
        tr, src = pmc_traffic("eval_general" if general else "eval", tag, lay, 1.0)
        if ev:  # (no record: the launch went untimed -- the line then carries no roofline rather than a made-up one)
            # (pipelined, decisions only, every run regular: the evaluation the engine launches is the lean variant)
            kname = "ev::k_eval_general" if general else ("ev::k_eval_sorted_lean" if ev["kernel"] == "ev::k_eval_sorted" else ev["kernel"])
            if not general and engine_info.get("grouping_path", "").endswith("hot slots peeled"):
                kname = "ev::k_eval_lean_hot"   # (round 6: the skewed stream's lean batches -- hot role + the sorted part in one kernel)
            source = {"avg_ms": "start/stop HIP events on the kernel's own dispatch packet (hipExtLaunchKernelGGL), engine's "
                                "stream, pipelined run of this process",
                      "traffic": (src or {}).get("file")}
            res["roofline"] = roofline_entry(kname, ev["avg_ms"], alg, ms, tr, source)
        gated = ("prep", "sort", "eval") if "bucket_eval" in inorder else ()
        res["detail"] = {"stages": {"pipelined": stage_table(piped, alg, tag, lay),
                                    "in_order": stage_table(inorder, alg, tag, lay, gated=gated)},
                         "in_order_sum_ms": sum(s["per_batch_ms"] for s in inorder.values()),
                         "traffic_source": src}
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
        # The checker (VERDICT r4 #6d: the string-key legs were the only timed legs not replayed): every batch the engine was given
        # -- prefill and timed region, the sweeps where they fell -- through the oracle's dense store keyed by the id a key is a
        # function of (one to one; the Store semantics are the same), on all host cores.  Compared: the allowed / denied / error /
        # swept counters of the whole run, the store's size, the decision bytes of the last timed batch.
        if not a.no_verify:
            tv = time.perf_counter()
            try:
                from oracle import oracle as O
                last_bytes = res.allowed.cpu().numpy()[:B].copy()
                orc = O.DenseOracle(cap + steps * B)   # (ids are dense from 0 in order of first appearance)
                th = O.host_threads()
                st2 = W.Config4Stream(B, n_prefill=n_prefill, long=long, sweep_every=4)   # the same stream again (deterministic)
                tot_allowed = tot = swept = 0
                for k in range(n_prefill):
                    ids, now = st2.prefill(k)
                    r_ = orc.batch_slots(ids.astype(np.uint32), b, c, p, 1, now, threads=th)
                    tot_allowed += int(r_.allowed.sum())
                    tot += len(ids)
                ok_bytes = False
                for s_ in range(steps):
                    ids, now = st2.mixed(s_)
                    r_ = orc.batch_slots(ids.astype(np.uint32), b, c, p, 1, now, threads=th)
                    tot_allowed += int(r_.allowed.sum())
                    tot += len(ids)
                    if s_ == steps - 1:
                        ok_bytes = bool(np.array_equal(last_bytes, r_.allowed))
                    if st2.sweep_due(s_):
                        swept += orc.sweep(now)
                ok_cnt = (c1["allowed"] == tot_allowed and c1["denied"] == tot - tot_allowed and c1["errors"] == 0 and c1["total"] == tot)
                ok_sweep = c1["swept"] == swept and c1["live_slots"] == orc.live()
                verified_keys = {"ok": bool(ok_cnt and ok_bytes and ok_sweep and eng.selfcheck() == 0), "counters": bool(ok_cnt),
                                 "decision_bytes_of_last_batch": ok_bytes, "swept_and_store_size": bool(ok_sweep), "batches_replayed": n_prefill + steps,
                                 "seconds": time.perf_counter() - tv}
                del orc
            except Exception as ex:  # noqa: BLE001 (the checker must not take the measurement down with it)
                verified_keys = {"ok": False, "error": f"{type(ex).__name__}: {ex}"[:160]}
            log(f"  verified: {verified_keys}")
        else:
            verified_keys = None
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
        r["whole_step_frac"] = alg / (dt / steps) / 1e9 / HBM_PEAK_GBS
        r["algorithmic_bytes_per_decision"] = alg / B
        # kernels per batch: the key stage is ONE profiled stage of three launches (k_probe, k_bind, k_follow)
        r["launches_per_batch"] = sum(st_["launches_per_batch"] * (3 if k_ == "hash" else 1) for k_, st_ in stages.items())
        # the dominant stage's kernels as ONE roofline entry (per-batch total of the stage: the key stage is several launches)
        r["roofline"] = roofline_entry(d["kernel"], d["per_batch_ms"], alg, 1e3 * dt / steps, tr,
                                       {"avg_ms": "HIP events, per-batch total of the stage's launches, pipelined profile steps "
                                                  "(sweeps left out)", "traffic": (src or {}).get("file")})
        r["grouping_path"] = eng.info()["grouping_path"]
        if verified_keys is not None:
            r["verified"] = verified_keys
        r["detail"] = {"stages": {"pipelined": stages}, "traffic_source": src,
                       "note": "the timed region also holds the sweeps (" + str(steps // 4) + "), the profile steps do not"}
        out[label] = r
        eng.close()
        del mixed
        torch.cuda.empty_cache()
    out["workload"] = ("10 x 1 Mi new keys, then 1 Mi-request batches: 70 % hits (1/5 of them on 1000 hot keys), 20 % new keys, "
                       "10 % re-hits of expired keys; rate (10,100/60 s); 1 s per batch; sweep every 4 batches")
    return out


def abi_shape_stream(n_keys, n, seed, now0):
    """One batch of the call a reference-side caller makes (rate_limiter.rs:102-110 per request, stamped by its transport,
    throttlecrab-server/src/transport/http.rs:128): string keys `key_<id>` (id uniform over the table), the key's own
    (burst, count, period) -- one of four plans by a hash of the id, handed over with EVERY request as the reference's API
    does --, quantity 1 or 2, a timestamp per request that is NOT monotone inside the batch (arrival order != stamp order).
    -> dict of host arrays."""
    from throttlecrab_amd import workload as W
    ids = W.uniform_slots(n_keys, n, seed=seed)
    h = W.splitmix64(ids.astype(np.uint64) ^ np.uint64(0xAB1))
    plans = np.array([(100, 1000, 3600), (20, 100, 60), (50, 600, 600), (10, 100, 60)], dtype=np.int64)
    pl = plans[(h % np.uint64(4)).astype(np.int64)]
    r = W.splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9))
    kb, ko = W.string_keys(ids)
    return {"ids": ids, "key_bytes": kb, "key_off": ko, "max_burst": pl[:, 0].copy(), "count_per_period": pl[:, 1].copy(),
            "period": pl[:, 2].copy(), "quantity": (1 + (r >> np.uint64(60)) % np.uint64(2)).astype(np.int64),
            "now_ns": (now0 + (np.arange(n, dtype=np.int64) * 900) // max(1, n // 1000) + ((r >> np.uint64(20)) % np.uint64(200_000)).astype(np.int64))}


ABI_COLS = ("max_burst", "count_per_period", "period", "quantity", "now_ns")


def abi_shape_bench(a, dev):
    """the reference-shaped call, timed.  `tc_rate_limit_batch_keys` exactly as rust/throttlecrab-gpu's
    `GpuRateLimiter::rate_limit_batch` issues it -- HOST pointers, string keys, per-request (burst, count, period), quantity
    and timestamp, `tc_decision` records out, flags 0, the store's own cleanup policy on (GpuStore::new: adaptive, the server's
    defaults) -- at 4 Ki, 64 Ki and 1 Mi requests over a table that holds `--keys` keys; the same from pinned buffers
    (tc_host_alloc), and from a ring of 4 pinned sets with TC_B_ASYNC.  PCIe-inclusive by construction (never `value`).
    Every call's decisions are compared with the oracle's replay of the same requests in the same order (the dense store keyed
    by the id inside `key_<id>`: keys and ids correspond one to one, the Store semantics are the same)."""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    S = 10**9
    n_keys, B = a.keys, a.batch
    eng = t.Engine(n_keys + B, B, device=dev.index or 0, key_mode=True)
    eng.set_sweep_policy("adaptive", created_ns=W.T0_NS, min_interval_ns=5 * S, max_interval_ns=300 * S, max_operations=1_000_000)
    orc = O.DenseOracle(n_keys)
    th = O.host_threads()
    # the table: every key seen once (device-pointer batches: not what is measured), with its own plan
    t_fill = time.perf_counter()
    for at in range(0, n_keys, B):
        ids = np.arange(at, min(at + B, n_keys), dtype=np.uint32)
        kb, ko = W.string_keys(ids)
        hh = W.splitmix64(ids.astype(np.uint64) ^ np.uint64(0xAB1))
        plans = np.array([(100, 1000, 3600), (20, 100, 60), (50, 600, 600), (10, 100, 60)], dtype=np.int64)
        pl = plans[(hh % np.uint64(4)).astype(np.int64)]
        cols = [torch.from_numpy(pl[:, j].copy()).to(dev) for j in range(3)]
        eng.rate_limit_batch_keys(torch.from_numpy(kb).to(dev), torch.from_numpy(ko.astype(np.int32)).to(dev), max_burst=cols[0],
                                  count_per_period=cols[1], period=cols[2], quantity=1, now_ns=W.T0_NS, want=("allowed",))
        eng.synchronize()
        orc.batch_slots(ids, pl[:, 0], pl[:, 1], pl[:, 2], 1, W.T0_NS, threads=th)
    fill_s = time.perf_counter() - t_fill
    out = {"table_keys": n_keys, "prefill_seconds": fill_s}
    ok_all, calls_checked, step_no = True, 0, 0

    def check(batch, dec_np, ctx):
        nonlocal ok_all, calls_checked
        ref = orc.batch_slots(batch["ids"], batch["max_burst"], batch["count_per_period"], batch["period"], batch["quantity"], batch["now_ns"], threads=th)
        d = t.Engine.unpack_decisions(dec_np)
        good = all(np.array_equal(d[f].astype(np.int64), getattr(ref, f).astype(np.int64)) for f in ("allowed", "status", "remaining", "reset_after_ns", "retry_after_ns"))
        ok_all = ok_all and good
        calls_checked += 1
        if not good:
            log(f"  abi_shape: {ctx} differs from the oracle")

    # The first ~20 TC_B_ASYNC calls of a process each stall the submitting thread for 5-7 ms, once (calls 1, 5, 11, 19 on
    # every box looked at, whatever the batch size: the runtime growing its pools; tools/abi_stall.py) -- a server pays that
    # at start-up, a measurement of 8 calls must not: 32 small calls first.
    wb = abi_shape_stream(n_keys, 4096, 999, W.T0_NS + S // 2)
    wp = {c: eng.host_alloc(wb[c].size, wb[c].dtype) for c in ("key_bytes", "key_off") + ABI_COLS}
    for c in wp:
        wp[c][:] = wb[c]
    wres = [t.BatchResult(decisions=eng.host_alloc(4 * 4096, np.int64)) for _ in range(4)]
    for i in range(32):
        eng.wait_batches(3)
        eng.rate_limit_batch_keys(wp["key_bytes"], wp["key_off"], **{c: wp[c] for c in ABI_COLS}, want=("decisions",), out=wres[i % 4], async_=True)
        orc.batch_slots(wb["ids"], wb["max_burst"], wb["count_per_period"], wb["period"], wb["quantity"], wb["now_ns"], threads=th)
    eng.wait_batches(0)
    for n, calls in ((4096, 48), (65536, 24), (B, 8)):
        if n > B:
            continue
        distinct = 4
        batches = []
        for k in range(distinct):
            batches.append(abi_shape_stream(n_keys, n, 1000 + step_no, W.T0_NS + S + step_no * 1_000_000))
            step_no += 1
        bytes_in = sum(int(batches[0][c].nbytes) for c in ("key_bytes", "key_off") + ABI_COLS) / n
        leg = {"requests": n, "bytes_in_per_request": bytes_in, "bytes_out_per_request": 32}
        # (i) pageable host arrays, synchronous: GpuRateLimiter::rate_limit_batch verbatim
        res = [t.BatchResult(decisions=np.zeros(4 * n, np.int64)) for _ in range(distinct)]
        for k in range(2):  # warm: staging allocations, the first sweep
            eng.rate_limit_batch_keys(batches[k]["key_bytes"], batches[k]["key_off"], **{c: batches[k][c] for c in ABI_COLS}, want=("decisions",), out=res[k])
            check(batches[k], res[k].decisions, f"{n} warm")
        t0 = time.perf_counter()
        for i in range(calls):
            k = i % distinct
            eng.rate_limit_batch_keys(batches[k]["key_bytes"], batches[k]["key_off"], **{c: batches[k][c] for c in ABI_COLS}, want=("decisions",), out=res[k])
            if i >= calls - distinct or i < distinct:
                pass
        dt = time.perf_counter() - t0
        leg["sync_pageable"] = {"value": calls * n / dt, "unit": "decisions/s", "us_per_call": 1e6 * dt / calls, "calls": calls}
        # the oracle follows in call order (the engine's results of the last `distinct` calls are still in `res`)
        for i in range(calls):
            k = i % distinct
            if i >= calls - distinct:
                check(batches[k], res[k].decisions, f"{n} sync call {i}")
            else:
                orc.batch_slots(batches[k]["ids"], batches[k]["max_burst"], batches[k]["count_per_period"], batches[k]["period"], batches[k]["quantity"], batches[k]["now_ns"], threads=th)
        # (ii) the same arrays in pinned memory (tc_host_alloc), synchronous; (iii) TC_B_ASYNC over a ring of 4 pinned sets
        pinned = []
        for k in range(distinct):
            pb = {c: eng.host_alloc(batches[k][c].size, batches[k][c].dtype) for c in ("key_bytes", "key_off") + ABI_COLS}
            for c in pb:
                pb[c][:] = batches[k][c]
            # ... and the same requests as the shims send them since round 6 (TC_B_PLAN_DICT: the distinct triples once, a 16-bit
            # index per request, the quantities as u32 -- rust/throttlecrab-gpu batch_chunk, throttlecrab_gpu.hpp submit_batch)
            dic, pid = t.Engine.encode_plans(batches[k]["max_burst"], batches[k]["count_per_period"], batches[k]["period"])
            for name, arr in (("plan_dict", dic.reshape(-1)), ("plan_id", pid), ("quantity32", batches[k]["quantity"].astype(np.uint32))):
                pb[name] = eng.host_alloc(arr.size, arr.dtype)
                pb[name][:] = arr
            pinned.append((pb, t.BatchResult(decisions=eng.host_alloc(4 * n, np.int64))))
        leg["bytes_in_per_request_dict"] = sum(int(pinned[0][0][c].nbytes) for c in ("key_bytes", "key_off", "now_ns", "plan_id", "quantity32")) / n

        def call(pb, r_, asy, dict_form):
            if dict_form:
                return eng.rate_limit_batch_keys(pb["key_bytes"], pb["key_off"], now_ns=pb["now_ns"], plan_dict=(pb["plan_dict"], pb["plan_id"], pb["quantity32"]),
                                                 want=("decisions",), out=r_, async_=asy)
            return eng.rate_limit_batch_keys(pb["key_bytes"], pb["key_off"], **{c: pb[c] for c in ABI_COLS}, want=("decisions",), out=r_, async_=asy)

        for mode in ("sync_pinned", "async_pinned_ring4", "sync_pinned_dict", "async_pinned_ring4_dict"):
            asy = mode.startswith("async")
            dict_form = mode.endswith("_dict")
            # warm: a freshly pinned array stalls the submitting thread for 5-7 ms the first few times a transfer touches it
            # (tools/abi_stall.py: calls 1, 5, 11, 19 of a ring of 4 sets, then never again) -- a server's staging buffers live
            # as long as the server; here every leg pins new ones, so each set is used six times before the clock starts
            n_warm = (6 if n >= (1 << 18) else 16) * distinct if mode == "sync_pinned" or asy else (2 * distinct if dict_form else 0)
            for i in range(n_warm):
                pb, r_ = pinned[i % distinct]
                if asy:
                    eng.wait_batches(distinct - 1)
                call(pb, r_, asy, dict_form)
            eng.wait_batches(0)
            for i in range(n_warm):
                k = i % distinct
                orc.batch_slots(batches[k]["ids"], batches[k]["max_burst"], batches[k]["count_per_period"], batches[k]["period"], batches[k]["quantity"], batches[k]["now_ns"], threads=th)
            t0 = time.perf_counter()
            for i in range(calls):
                pb, r_ = pinned[i % distinct]
                if asy:
                    eng.wait_batches(distinct - 1)  # the oldest set of the ring is free again
                call(pb, r_, asy, dict_form)
            if asy:
                eng.wait_batches(0)
            dt = time.perf_counter() - t0
            leg[mode] = {"value": calls * n / dt, "unit": "decisions/s", "us_per_call": 1e6 * dt / calls, "calls": calls}
            for i in range(calls):
                k = i % distinct
                if i >= calls - distinct:
                    check(batches[k], pinned[k][1].decisions, f"{n} {mode} call {i}")
                else:
                    orc.batch_slots(batches[k]["ids"], batches[k]["max_burst"], batches[k]["count_per_period"], batches[k]["period"], batches[k]["quantity"], batches[k]["now_ns"], threads=th)
        out[str(n)] = leg
        del pinned, res, batches
    c = eng.counters()
    st = eng.sweep_stats()
    out["verified"] = {"ok": bool(ok_all and c["errors"] == 0 and eng.selfcheck() == 0), "calls_compared_with_the_oracle": calls_checked,
                       "note": "the last 4 calls of every leg field by field; the oracle replays every call in between to stay in step"}
    out["store_cleanups"] = {k: st[k] for k in ("sweeps", "sweeps_by_time", "sweeps_by_operations", "sweeps_by_size", "sweeps_for_room", "retries", "feed_waits")}
    eng.close()
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


def run_exchange(a, t, W, dev, local, rank, world, dist, fab):
    """--route exchange: rank r is handed slice r (a.batch requests) of every global batch, routes it into one segment
    per destination (tc_route_batch, only = -1), puts the segments into the destinations' inboxes (tc_forward_segments:
    peer copies, no collective) and evaluates the `world` inboxes of a step as ONE batch (segmented slot column,
    sources in rank order).  Per-GPU routing work is a.batch ids per step whatever the number of GPUs."""
    import torch
    from throttlecrab_amd import sharded
    B = a.batch
    G = world * B
    eng = t.Engine(a.keys, 2 * B, device=local, fixed_params=(a.layout == "fixed"), track_denied=True)
    eng.use_torch_stream()
    eng.register_params_uniform(*W.REF_PARAMS)
    n_distinct = 8
    if a.workload == "zipf":
        z = W.Zipf(world * a.keys)
        host = [z.slots(B, seed=3, start=i * G + rank * B) for i in range(n_distinct)]   # slice `rank` of global batch i
    else:
        host = [W.uniform_slots(world * a.keys, B, seed=2, start=i * G + rank * B) for i in range(n_distinct)]
    d_slice = [torch.from_numpy(h.astype(np.int32)).to(dev) for h in host]
    xr = sharded.ExchangeRank(eng, fab, rank, world, B)
    # a step's decisions: at most world x B requests may land here (every source's whole slice)
    outs = [t.BatchResult(allowed=torch.empty(world * B, dtype=torch.uint8, device=dev)) for _ in range(OUT_RING)]
    cnt_view = sharded.device_counter_view(eng)
    gathered = torch.zeros(world * cnt_view.numel(), dtype=torch.int64, device=dev)
    top_gathered = torch.zeros(world * sharded.TOPK, 2, dtype=torch.int64, device=dev)
    decided = 0
    # the host never waits in steady state: a slice is routed (straight into the destinations' inboxes) LA_ROUTE steps and
    # announced LA_POST steps before it is evaluated; the three phases of a step are ONE library call (tc_exchange_step)
    LA_ROUTE, LA_POST = 4, 1

    def step(i, last=False, metrics=True):
        nonlocal decided
        decided += xr.timed("step", i, d_slice[(i + LA_ROUTE) % n_distinct], LA_ROUTE, LA_POST, W.T0_NS + i * 1_000_000, outs)
        if not metrics:
            return
        if i % METRICS_EVERY == METRICS_EVERY - 1 or last:
            eng.counters_refresh()
            dist.all_gather_into_tensor(gathered, cnt_view)

    def top_denied_exchange():
        # The optional part of the metrics payload -- every shard's most denied keys as (global id, count) -- is an ON-DEMAND query
        # (the reference builds it when /metrics is scraped: throttlecrab-server/src/metrics.rs), not a part of a step: once, right
        # BEHIND the timed region.  (Until late round 6 it ran inside the last timed step: a device synchronisation, a copy to the
        # host, the packing in Python and an all-gather, 1.4-1.5 ms -- 70 us per step of the driver's 20-step region, more than
        # the step itself.)  The counter block's all-gather every METRICS_EVERY steps and at the last step stays inside.
        block = torch.from_numpy(sharded.pack_top_denied(eng.top_denied(sharded.TOPK), rank, world, a.keys)).to(dev)
        dist.all_gather_into_tensor(top_gathered, block)

    for j in range(LA_ROUTE):
        xr.route(j, d_slice[j % n_distinct])
    for j in range(LA_POST):
        xr.post(j)
    it = 0
    for _ in range(a.warmup):
        step(it)
        it += 1
    # (the counter block's all-gather falls on every METRICS_EVERY-th step: a short warmup holds none, and the first call of a
    # collective kind pays RCCL's set-up for it -- one untimed call here)
    eng.counters_refresh()
    dist.all_gather_into_tensor(gathered, cnt_view)
    dist.barrier()
    torch.cuda.synchronize()
    decided = 0
    for k_ in xr.host_s:
        xr.host_s[k_] = 0.0   # (the warmup holds one-time costs: scratch allocation, the stream probe)
    xr.wait_us()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(it, last=(k == a.steps - 1))
        it += 1
    torch.cuda.synchronize()
    dist.barrier()
    dt_mine = time.perf_counter() - t0
    top_denied_exchange()
    host_us = {k_: 1e6 * v / a.steps for k_, v in xr.host_s.items() if v}   # of the timed region only
    w_ = xr.wait_us()   # ... of which the library spent waiting (for inbox slots, the router's tag, the other ranks)
    host_us.update({"waiting_for_inbox_slots": w_[0] / a.steps, "waiting_for_router": w_[1] / a.steps, "waiting_for_sources": w_[2] / a.steps})
    tm = torch.tensor([dt_mine], dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dt = float(tm.item())
    res = sharded_summary(a, t, W, eng, dev, rank, world, dist, dt, decided, G, cnt_view, top_gathered)
    res["route"] = "exchange"
    res["host_us_per_step"] = host_us  # where the host's time goes
    print(f"[bench] rank {rank} host us/step (timed region): {host_us}", file=sys.stderr, flush=True)
    # roofline of this rank's evaluation: HIP events per kernel (needs no peers: the inboxes of the last two steps again, handed
    # to the engine directly as segmented batches)
    steps_p = min(a.steps, 12)
    segs_last = [[(fab.inbox(rank, (it - 1 - q) % fab.ring, s_), int(fab.mail[rank, (it - 1 - q) % fab.ring, s_, 0])) for s_ in range(world)]
                 for q in range(2)]
    eng.profile_enable(True)
    n_prof = 0
    for k in range(steps_p):
        sg = segs_last[k % 2]
        tot = sum(c_ for _, c_ in sg)
        if tot == 0 or tot > eng.max_batch:
            continue
        eng.rate_limit_batch_slots(None, segments=sg, registered=True, quantity=1, now_ns=W.T0_NS + (it + k) * 1_000_000, want=("allowed",),
                                   out=outs[k % OUT_RING], inputs_ready=True, outputs_idle=True)
        n_prof += tot
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    stages = {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps_p, "avg_ms": ms / calls, "per_batch_ms": ms / steps_p}
              for k, (ms, calls) in prof.items() if calls}
    if stages:
        alg = ALG_BYTES_PER_DECISION * n_prof / steps_p
        ev = stages.get("eval") or stages.get("bucket_eval")
        if ev and ev["launches_per_batch"] >= 0.999:
            res["roofline"] = roofline_entry(ev["kernel"], ev["per_batch_ms"], alg, 1e3 * dt / a.steps, None,
                                             {"avg_ms": f"HIP events, rank {rank}, per-step total of the evaluation launches", "traffic": None})
        res["stages"] = stages
        print(f"[bench] rank {rank} evaluation stages: " + ", ".join(f"{k} {v['avg_ms'] * 1e3:.1f} us x {v['launches_per_batch']:.1f}" for k, v in stages.items()), file=sys.stderr, flush=True)
    # the routing front end alone (this rank's slice routed straight into the inboxes, 8 routers drained), for the record;
    # the inboxes are scratch by now -- once every rank has left its profile steps
    dist.barrier()
    torch.cuda.synchronize()
    cnt_scratch = [torch.zeros(world, dtype=torch.int32, device=dev) for _ in range(8)]
    t1 = time.perf_counter()
    for k in range(8):
        eng.route_batch(d_slice[k % n_distinct], world, only=-1, out=(None, None, cnt_scratch[k]), ahead=True, no_readers=True,
                        out_dst=[fab.inbox(d, k % fab.ring, rank) for d in range(world)])
    torch.cuda.synchronize()
    res["router_ms_per_step"] = 1e3 * (time.perf_counter() - t1) / 8
    xr.close()
    dist.barrier()
    eng.close()
    return res


def sharded_summary(a, t, W, eng, dev, rank, world, dist, dt, decided, G, cnt_view, top_gathered):
    """what every --route mode reports: whole-job rate, per-GPU shares, imbalance, the metrics exchange"""
    import torch
    from throttlecrab_amd import sharded
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
    return {"value": total_decided / dt, "ms_per_step": 1e3 * dt / a.steps, "per_gpu": per_gpu,
            "imbalance_max_over_mean": float(ev[:, 0].max()) / max(1.0, float(ev[:, 0].mean())),
            "allowed_fraction": float(ev[:, 1].sum()) / max(1, int(ev[:, 1].sum() + ev[:, 2].sum())),
            "metrics_exchange": {"counter_block_bytes_per_gpu": 8 * int(cnt_view.numel()), "every_steps": METRICS_EVERY,
                                 "top_denied_block_bytes_per_gpu": 16 * sharded.TOPK, "top_denied_exchanges": 1,
                                 "top_denied_exchange_where": "once, right behind the timed region (an on-demand query, like a /metrics scrape); the counter block's all-gather is inside it",
                                 "top_denied_global": sharded.merge_top_denied(top_gathered.cpu().numpy(), 5)}}


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
    # every batch in flight writes its decisions to an array of its own (as the N = 1 run does): TC_B_OUTPUTS_IDLE; an array holds
    # a rank's whole share of a global batch (the owner of a hot key gets more than 1 / world of it)
    outs = [t.BatchResult(allowed=torch.empty(G, dtype=torch.uint8, device=dev)) for _ in range(OUT_RING)]
    cnt_view = sharded.device_counter_view(eng)
    gathered = torch.zeros(world * cnt_view.numel(), dtype=torch.int64, device=dev)
    top_gathered = torch.zeros(world * sharded.TOPK, 2, dtype=torch.int64, device=dev)
    decided = 0
    LOOKAHEAD = 4
    # The rank's side of `replicate` is library code (csrc/shard.hip): tc_shard_step routes global batch i + 4 on the
    # engine's grouping streams (beside the evaluations; the router's last block leaves the counts + the batch's tag in pinned host
    # memory), polls the tag of batch i -- routed four steps ago: no wait in steady state, no event, no stream synchronisation --
    # and decides the rank's share in chunks of at most max_batch: ONE call per step (round 4: route_batch + a poll + a batch call
    # per chunk from Python, 26-38 us of host time per step).
    xr = sharded.ShardRank(eng, rank, world, G, ring=LOOKAHEAD + 4)
    host_s = {"step": 0.0}   # where the host's time goes (diagnostics, stderr + detail)

    def route(i):
        xr.route(i, d_global[i % n_distinct])

    def evaluate(i, last=False, metrics=True):
        nonlocal decided
        t_ = time.perf_counter()
        decided += xr.step(i, d_global[(i + LOOKAHEAD) % n_distinct], LOOKAHEAD, W.T0_NS + i * 1_000_000, outs)
        host_s["step"] += time.perf_counter() - t_
        if not metrics:
            return
        t_ = time.perf_counter()
        if i % METRICS_EVERY == METRICS_EVERY - 1 or last:
            eng.counters_refresh()
            dist.all_gather_into_tensor(gathered, cnt_view)
        host_s["counter_exchange"] = host_s.get("counter_exchange", 0.0) + time.perf_counter() - t_

    def top_denied_exchange():
        # (see run_exchange: the on-demand part of the metrics payload, once, right behind the timed region)
        block = torch.from_numpy(sharded.pack_top_denied(eng.top_denied(sharded.TOPK), rank, world, a.keys)).to(dev)
        dist.all_gather_into_tensor(top_gathered, block)

    # The router runs LOOKAHEAD global batches ahead of the evaluation: the host needs a batch's count (how many of
    # its requests this rank owns) before it can enqueue the evaluation, and that read must not drain the stream.
    it = 0
    for j in range(LOOKAHEAD):
        route(j)
    for _ in range(a.warmup):
        evaluate(it)
        it += 1
    # (one untimed all-gather of the counter block: a short warmup holds none -- see run_exchange)
    eng.counters_refresh()
    dist.all_gather_into_tensor(gathered, cnt_view)
    dist.barrier()
    torch.cuda.synchronize()
    decided = 0
    for k_ in host_s:
        host_s[k_] = 0.0   # (the warmup holds one-time costs: scratch allocation, the stream probe)
    xr.wait_us()
    t0 = time.perf_counter()
    for k in range(a.steps):
        evaluate(it, last=(k == a.steps - 1))   # (routes batch it + LOOKAHEAD too: the last LOOKAHEAD of them for nothing, inside the timing, against us)
        it += 1
    t_loop = time.perf_counter() - t0   # (the enqueue loop alone: if the drain behind it is short, the host is what the step waits for)
    torch.cuda.synchronize()
    t_drain = time.perf_counter() - t0 - t_loop
    dist.barrier()
    dt_mine = time.perf_counter() - t0
    t_ = time.perf_counter()
    top_denied_exchange()
    t_top = time.perf_counter() - t_
    host_us = {k_: 1e6 * v / a.steps for k_, v in host_s.items()}   # of the timed region only
    host_us["top_denied_exchange_behind_the_region_total_us"] = 1e6 * t_top
    host_us["enqueue_loop"] = 1e6 * t_loop / a.steps
    host_us["drain_behind_the_loop_total_us"] = 1e6 * t_drain
    host_us["of_it_waiting_for_the_router"] = xr.wait_us() / a.steps
    print(f"[bench] rank {rank} host us/step (timed region): {host_us}", file=sys.stderr, flush=True)
    tm = torch.tensor([dt_mine], dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dt = float(tm.item())
    res = sharded_summary(a, t, W, eng, dev, rank, world, dist, dt, decided, G, cnt_view, top_gathered)
    res["route"] = "replicate"
    res["host_us_per_step"] = host_us
    # roofline of this rank's evaluation (same kernels as the N = 1 run, fed by the router): HIP events per kernel
    steps_p = min(a.steps, 20)
    decided = 0
    eng.profile_enable(True)
    for _ in range(steps_p):
        evaluate(it, metrics=False)
        it += 1
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    stages = {k: {"kernel": KERNEL_OF_STAGE[k], "launches_per_batch": calls / steps_p, "avg_ms": ms / calls, "per_batch_ms": ms / steps_p}
              for k, (ms, calls) in prof.items() if calls}
    if stages:
        alg = ALG_BYTES_PER_DECISION * decided / steps_p
        ev = stages.get("eval") or stages.get("bucket_eval")
        if ev and ev["launches_per_batch"] >= 0.999:
            # (an owner's share may be evaluated in several chunks: per-step total of the evaluation launches)
            res["roofline"] = roofline_entry(ev["kernel"], ev["per_batch_ms"], alg, 1e3 * dt / a.steps, None,
                                             {"avg_ms": f"HIP events, rank {rank}, per-step total of the evaluation launches", "traffic": None})
        res["stages"] = stages
    eng.close()
    return res


def launch_plan(a, argv, env):
    """What `--gpus N` means for THIS process: ("run", world) -- go on as rank RANK of `world` -- or ("launch", cmd) -- this is a
    plain `python bench.py --gpus N` with N > 1: start the N ranks (one per GPU, RCCL over 127.0.0.1), which is what the driver's
    torchrun line does (README.md:247-249: the reference's scale-out is sharding by key; BASELINE configs[3]) -- or
    ("error", message).  Pure: no GPU, no torch."""
    under_torchrun = "WORLD_SIZE" in env
    if under_torchrun:
        world = int(env["WORLD_SIZE"])
        if a.gpus is not None and a.gpus != world:
            return "error", f"--gpus {a.gpus} but torchrun started {world} rank(s) (WORLD_SIZE): the line would claim GPUs that did no work"
        return "run", world
    gpus = 1 if a.gpus is None else a.gpus
    if gpus < 1:
        return "error", f"--gpus {gpus}"
    if gpus == 1:
        return "run", 1
    port = env.get("MASTER_PORT") or str(29400 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + [x for x in argv if x != "--plan-only"]
    return "launch", cmd


def main():
    a = parse()
    kind, plan = launch_plan(a, sys.argv[1:], os.environ)
    if a.plan_only:
        print(json.dumps({"action": kind, "world" if kind == "run" else ("cmd" if kind == "launch" else "error"): plan}))
        return
    if kind == "error":
        sys.exit(f"bench.py: {plan}")
    if kind == "launch":
        if os.environ.get("TC_BENCH_ONE_DEVICE") != "1":  # (that switch puts every rank on GPU 0, over gloo: what a 1-GPU box can test)
            import torch
            have = torch.cuda.device_count()
            if have < a.gpus:
                sys.exit(f"bench.py: --gpus {a.gpus} but this node shows {have} GPU(s); nothing was measured")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(plan[0], plan)  # the ranks' rank 0 prints the line
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W

    world = plan
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # TC_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, RCCL all-gather of the counter
    # blocks, max-over-ranks timing) even with one rank -- the only way to exercise it on a 1-GPU box
    if world > 1 or os.environ.get("TC_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "RANK" not in os.environ:   # TC_BENCH_FORCE_DIST=1 started plainly: a process group of one
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 2000))
        if os.environ.get("TC_BENCH_ONE_DEVICE") == "1":  # several ranks on ONE GPU (what a 1-GPU box can test of the N > 1 path)
            local = 0
        torch.cuda.set_device(local)
        if os.environ.get("TC_BENCH_ONE_DEVICE") == "1":
            dist.init_process_group("gloo")   # (RCCL refuses two ranks on one GPU; gloo carries the metrics through host memory)
            _patch_collectives_for_gloo(dist)
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device(f"cuda:{local}")

    if dist is not None:
        # a real stream for the engine AND torch / RCCL (torch's default stream has handle 0, which the engine reads as
        # "use your own stream": the metrics all-gather would then not be ordered behind the counters' refresh)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            fab = None
            if a.route == "exchange":
                # inboxes shared through IPC handles, mailboxes in shared memory: set up ONCE, outside the timed region.
                # Every rank must succeed, else all of them fall back to --route replicate (which needs no peer memory).
                from throttlecrab_amd import sharded
                ok = 1
                try:
                    fab = sharded.IpcFabric(dist, rank, world, a.batch, 8, dev)  # 8 ring slots of world x batch ids per rank
                except Exception as ex:  # noqa: BLE001 (whatever the IPC layer raises: fall back, but say so)
                    print(f"[bench] rank {rank}: exchange set-up failed ({type(ex).__name__}: {ex}); falling back to --route replicate",
                          file=sys.stderr, flush=True)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int64, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if fab is not None:
                        fab.close()
                    fab = None
            if fab is not None:
                sh = run_exchange(a, t, W, dev, local, rank, world, dist, fab)
                fab.close()
            else:
                sh = run_sharded(a, t, W, dev, local, rank, world, dist)
            torch.cuda.synchronize()
        # n_gpus of the line = the ranks the collective actually reached, not what a flag or an environment variable says
        seen = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen = int(seen.item())
        if rank == 0:
            res = {
                "metric": "GCRA decisions/sec, 10M keys per GPU", "value": sh["value"], "unit": "decisions/s", "n_gpus": ranks_seen,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": sh["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                "config": {"workload": f"configs[3]: {world} x {a.keys} keys hash-sharded across {world} GPU(s), ONE global {a.workload} "
                                       f"stream of {world} x {a.batch} requests per step, route={sh.get('route', a.route)}, "
                                       f"params (100,1000/3600s), q=1; RCCL all-gather of the metrics only",
                           "keys_per_gpu": a.keys, "batch": world * a.batch, "stream": a.workload, "resident_state": a.layout},
                "allowed_fraction": sh["allowed_fraction"], "per_gpu": sh["per_gpu"], "route": sh.get("route", a.route),
                "router_ms_per_step": sh.get("router_ms_per_step"),
                "imbalance_max_over_mean": sh["imbalance_max_over_mean"],
                "roofline": sh.get("roofline"), "cpu_baseline": None}
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            # (RCCL writes its version banner through C stdio, which a pipe only sees when the buffer is flushed -- at exit, BEHIND
            # the line: the process group goes first, C stdio is flushed, and the compact line is the last thing on stdout)
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            emit(res, {"metrics_exchange": sh["metrics_exchange"], "stages": sh.get("stages"), "host_us_per_step": sh.get("host_us_per_step"),
                       "note": "cpu_baseline is reported by the N = 1 run; roofline here is rank 0's evaluation kernel over the "
                               "requests it owns (the router's kernels are not in it)"})
        return

    general = a.workload.startswith("general")
    stream = {"general": "uniform", "general_zipf": "zipf"}.get(a.workload, a.workload)
    log(f"headline: {a.workload} / {a.layout}")
    main_res, eng, d_batches, dt = measure_stream(a, t, W, stream, dev, local, rank, 0, None, 1, general=general)
    cfg_no = {"uniform": 1, "zipf": 2}[stream]
    result = {
        "metric": "GCRA decisions/sec, 10M keys", "value": main_res["value"], "unit": "decisions/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
        "data": "synthetic",
        "config": {"workload": f"configs[{cfg_no}]: {a.keys} pre-hashed keys SoA on 1 GPU, {stream} request stream, batch={a.batch}, "
                               f"params (100,1000/3600s), q=1, " + ("a timestamp per request" if general else "one timestamp per batch") +
                               ", decisions only",
                   "keys_per_gpu": a.keys, "batch": a.batch, "stream": a.workload, "resident_state": a.layout,
                   "pipelined": not a.in_order},
        "allowed_fraction": main_res["allowed_fraction"], "pipelining_degraded": main_res.get("pipelining_degraded"),
    }
    if "repeats" in main_res:
        result["ms_per_step_median5"] = main_res["repeats"]["median"]
        result["ms_per_step_range5"] = [main_res["repeats"]["min"], main_res["repeats"]["max"]]
    if a.plans != "one":
        result["config"]["plans"] = a.plans
        result["config"]["workload"] += f", a plan per key ({a.plans})"
    # every leg that was replayed through the oracle: name -> ok; `verified` = all of them
    verified = {}
    if "verified" in main_res:
        verified["headline"] = main_res["verified"]
    detail = {"headline": main_res.pop("detail", None), "engine_info": main_res.pop("engine_info", None),
              "notes": {"resident_state": {"fixed": "TC_CFG_FIXED_PARAMS: TAT column, 8 B per key + the plan dictionary (emission interval, "
                                                    "tolerance, burst capacity per plan)",
                                           "wide": "{tat, expiry} cell, 16 B per key + plan id column + the plan dictionary"},
                        "pipelining": "TC_B_INPUTS_READY: batch k+1.. grouped on auxiliary streams while batch k is evaluated",
                        "outputs": "allowed u8 (decisions only)"}}
    if "roofline" in main_res:
        result["roofline"] = main_res["roofline"]

    if rank == 0:
        # Every secondary leg is optional: whatever goes wrong in one of them (a HIP error, a full table, no memory) is
        # recorded under "errors" and the headline line is still printed -- the driver's record depends on that line.
        errors = {}

        only = [x for x in os.environ.get("TC_BENCH_LEGS", "").split(",") if x]  # (iteration aid: run only the legs whose name contains one of these)

        def leg(name, fn):
            if only and not any(o in name for o in only):
                return
            log(name)
            try:
                fn()
            except Exception as ex:  # noqa: BLE001
                errors[name] = f"{type(ex).__name__}: {ex}"[:200]
                print(f"[bench] secondary leg '{name}' failed: {errors[name]}", file=sys.stderr, flush=True)

        if not a.no_also and not a.profile_run:
            other = "zipf" if stream == "uniform" else "uniform"
            held = {}

            def other_stream():  # the other BASELINE stream (configs[1] <-> configs[2]), measured the same way
                o_res, held["eng2"], held["ob"], _ = measure_stream(a, t, W, other, dev, local, 0, 0, None, 1)
                detail[f"{other}_stream"] = (o_res.pop("engine_info", None), o_res.pop("detail", None))[1]
                result[f"{other}_stream"] = o_res
                if "verified" in o_res:
                    verified[f"{other}_stream"] = o_res["verified"]

            def other_layout():  # the headline stream on the other resident-state layout
                a2 = argparse.Namespace(**vars(a))
                a2.layout = "fixed" if a.layout == "wide" else "wide"
                l_res, eng_l, _, _ = measure_stream(a2, t, W, stream, dev, local, 0, 0, None, 1, general=general)
                eng_l.close()
                detail[f"{a2.layout}_layout"] = (l_res.pop("engine_info", None), l_res.pop("detail", None))[1]
                result[f"{a2.layout}_layout"] = l_res
                if "verified" in l_res:
                    verified[f"{a2.layout}_layout"] = l_res["verified"]

            def general_of(gs):  # what a server's queue looks like: a timestamp per request (k_eval_general)
                def run():
                    g_res, eng_g, _, _ = measure_stream(a, t, W, gs, dev, local, 0, 0, None, 1, general=True)
                    eng_g.close()
                    detail[f"general_{gs}"] = (g_res.pop("engine_info", None), g_res.pop("detail", None))[1]
                    result[f"general_{gs}"] = g_res
                    if "verified" in g_res:
                        verified[f"general_{gs}"] = g_res["verified"]
                return run

            def per_key_plans():
                # the path north_star describes: (burst, count, period) live PER KEY (rate_limiter.rs:102-123 takes them per
                # call); the evaluation gathers rate_id[slot] and the plan's (emission interval, tolerance) for every request
                pk = {}
                for st_ in ("uniform", "zipf"):
                    for pl in ("tiers4", "tiers1000"):
                        if a.plans == pl and st_ == stream:
                            continue
                        p_res, eng_p, _, _ = measure_stream(a, t, W, st_, dev, local, 0, 0, None, 1, plans=pl)
                        eng_p.close()
                        detail[f"per_key_{st_}_{pl}"] = (p_res.pop("engine_info", None), p_res.pop("detail", None))[1]
                        pk[f"{st_}_{pl}"] = p_res
                        if "verified" in p_res:
                            verified[f"per_key_{st_}_{pl}"] = p_res["verified"]
                result["per_key_plans"] = pk

            def output_forms():
                if "eng2" in held:
                    detail["also"] = secondary(a, t, W, held["eng2"], held["ob"], d_batches, dev, local, other)

            def string_keys():
                sk = keys_bench(a, dev)
                for k, v in sk.items():
                    if isinstance(v, dict) and "verified" in v:
                        verified[f"string_keys_{k}"] = v.pop("verified")
                detail["string_keys"] = {k: (v.pop("detail", None) if isinstance(v, dict) else v) for k, v in sk.items()}
                result["string_keys"] = {k: v for k, v in sk.items() if isinstance(v, dict)}

            leg(f"other stream: {other}", other_stream)
            leg("per-key plans", per_key_plans)
            leg("other layout", other_layout)
            for gs in ("uniform", "zipf"):
                if not (general and gs == stream):
                    leg(f"general batches: {gs}", general_of(gs))
            leg("secondary output forms", output_forms)
            if "eng2" in held:
                leg("close second engine", held["eng2"].close)
            leg("string keys", string_keys)

            def abi_shape():
                ab = abi_shape_bench(a, dev)
                detail["abi_shape"] = ab
                result["abi_shape"] = {k: {m: _pick(v[m], ("value", "us_per_call")) for m in ("sync_pageable", "sync_pinned", "async_pinned_ring4", "sync_pinned_dict", "async_pinned_ring4_dict")}
                                       for k, v in ab.items() if isinstance(v, dict) and "requests" in v}
                result["abi_shape"]["verified"] = ab["verified"]["ok"]
                verified["abi_shape"] = ab["verified"]
            leg("abi shape (host pointers, string keys, per-request parameters)", abi_shape)
        if not a.no_cpu and not a.profile_run:
            def cpu():
                result["cpu_baseline"] = cpu_baseline(stream, a.keys, a.batch, a.cpu_sample_batches)
            leg("cpu baseline", cpu)
        if errors:
            detail["errors"] = errors
            result["errors"] = sorted(errors)
        if verified:
            detail["verified"] = verified
            result["verified"] = all(v.get("ok") for v in verified.values())
            result["verified_legs"] = len(verified)                                        # how many legs were replayed
            result["verify_failed"] = sorted(k for k, v in verified.items() if not v.get("ok"))   # ... and which of them differ
        else:
            result["verified"] = None   # (--no-verify / --profile-run: nothing was replayed)
        emit(result, detail)
    eng.close()


def _patch_collectives_for_gloo(dist):
    """TC_BENCH_ONE_DEVICE=1 only: gloo has no all_gather_into_tensor and reduces CUDA tensors through the host"""
    import torch
    real_ag, real_ar = dist.all_gather, dist.all_reduce

    def all_gather_into_tensor(out, inp):
        parts = [torch.zeros_like(inp, device="cpu") for _ in range(dist.get_world_size())]
        real_ag(parts, inp.cpu())
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).reshape(out.shape))

    def all_reduce(tns, op=dist.ReduceOp.SUM):
        c = tns.cpu()
        real_ar(c, op=op)
        tns.copy_(c)
    dist.all_gather_into_tensor, dist.all_reduce = all_gather_into_tensor, all_reduce


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _r(x, nd=4):
    """round floats (recursively) so that the compact line stays compact"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 2}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, list):
        return [_r(v, nd) for v in x]
    return x


def compact_line(result):
    """The ONE JSON line the driver parses: the contract keys, the roofline of the critical-path kernel, the CPU
    baseline, and one number (+ its whole-step roofline fraction) per secondary workload.  Everything else lives in
    bench_detail.json.  Never longer than COMPACT_LIMIT bytes."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "data", "config", "allowed_fraction", "imbalance_max_over_mean", "router_ms_per_step", "route", "errors",
           "verified", "verified_legs", "verify_failed", "pipelining_degraded", "ms_per_step_median5", "ms_per_step_range5")
    c = _pick(result, top)
    rf = result.get("roofline")
    if rf:
        c["roofline"] = _pick(rf, ("bound", "kernel", "avg_ms", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                   "whole_step_frac", "invalid"))
        c["roofline"]["source"] = (rf.get("source") or {}).get("traffic") if isinstance(rf.get("source"), dict) else rf.get("source")
        if c["roofline"].get("frac") is not None:
            assert 0.0 < c["roofline"]["frac"] <= 1.0, c["roofline"]
    cb = result.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        c["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        if cb.get("all_cores"):
            c["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        if cb.get("reference_shape"):
            c["cpu_baseline"]["reference_shape"] = _pick(cb["reference_shape"], ("value",))
    else:
        c["cpu_baseline"] = None
    # (north_star states its target on the Zipf stream -- its roofline object beside the headline's)
    for k in ("zipf_stream", "uniform_stream"):
        r2 = (result.get(k) or {}).get("roofline") if isinstance(result.get(k), dict) else None
        if r2 and r2.get("frac") is not None and "invalid" not in r2:
            c["roofline_" + k.split("_")[0]] = _pick(r2, ("kernel", "avg_ms", "achieved", "frac", "traffic", "whole_step_frac"))
    one = ("value", "ms_per_step", "whole_step_frac", "grouping_path")
    for k in ("zipf_stream", "uniform_stream", "wide_layout", "fixed_layout", "general_uniform", "general_zipf"):
        if isinstance(result.get(k), dict):
            c[k] = _pick(result[k], one)
            r2 = result[k].get("roofline")
            if r2 and r2.get("frac") is not None and "invalid" not in r2:
                c[k]["kernel_frac"] = r2["frac"]
                c[k]["kernel_ms"] = r2["avg_ms"]
    if isinstance(result.get("per_key_plans"), dict):
        c["per_key_plans"] = {}
        for k, v in result["per_key_plans"].items():
            c["per_key_plans"][k] = _pick(v, ("value", "ms_per_step"))
            r2 = v.get("roofline")
            if r2 and r2.get("frac") is not None and "invalid" not in r2:
                c["per_key_plans"][k]["kernel_ms"] = r2["avg_ms"]
                c["per_key_plans"][k]["kernel_frac"] = r2["frac"]
    if isinstance(result.get("string_keys"), dict):
        c["string_keys"] = {k: _pick(v, one + ("launches_per_batch",)) for k, v in result["string_keys"].items() if isinstance(v, dict)}
    if isinstance(result.get("per_gpu"), list):
        c["per_gpu"] = [_pick(g, ("rank", "share_of_traffic", "decisions_per_s")) for g in result["per_gpu"]][:8]
    if isinstance(result.get("abi_shape"), dict):  # requests per call -> decisions/s by mode (PCIe-inclusive)
        c["abi_shape"] = {k: ({m: vv.get("value") for m, vv in v.items()} if isinstance(v, dict) else v) for k, v in result["abi_shape"].items()}
    c["detail"] = DETAIL_PATH
    c = _r(c)
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:  # shed the optional parts, largest first, rather than break the contract
        for k in ("per_gpu", "roofline_zipf", "roofline_uniform", "string_keys", "per_key_plans", "abi_shape", "general_uniform", "general_zipf", "fixed_layout", "wide_layout", "allowed_fraction"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= COMPACT_LIMIT:
                break
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def emit(result, detail):
    """detail file (best effort) + stderr pointer, then the compact line as the LAST line on stdout"""
    full = dict(result)
    full["detail"] = detail
    try:
        path = os.path.join(ROOT, DETAIL_PATH)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        print(f"[bench] detail written to {path}", file=sys.stderr, flush=True)
    except OSError as e:
        print(f"[bench] could not write the detail file: {e}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(result), flush=True)


if __name__ == "__main__":
    main()
