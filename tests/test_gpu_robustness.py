"""Error paths and forward progress.
* A staging copy that fails inside a batch (tc_debug_fail_copy) must fail the call with TC_E_HIP, apply nothing,
  and leave the engine fully usable -- in every batch path that stages host data (synchronous, TC_B_ASYNC,
  string keys, pipelined).
* The engine's wait loops (radix look-back, direct stores of k_eval_sorted, hand-over chain of k_eval_general)
  each wait only for blocks dispatched earlier.  Two processes drive two engines on ONE device with the
  batches that lean on those waits hardest (hot keys with runs of 100 000 allowed requests crossing ~1 600
  waves, per-request timestamps, skewed uniform batches), so that their kernels contend for the same CUs -- and, round 4,
  against FILLER kernels (tc_debug_occupy) that sit on half of the CUs, on every other CU, or on every CU's LDS while the
  batches run; every result is checked against the oracle and the spin watchdog (tc_selfcheck) must stay silent."""
import os
import socket
import sys
import time

import numpy as np
import pytest

from tests.test_gpu_slots import FIELDS, T0, _oracle, assert_same, assert_state_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_failed_staging_copy_applies_nothing_and_engine_goes_on():
    import throttlecrab_amd as t
    from throttlecrab_amd import _lib as L
    cap, n = 5000, 20000
    rng = np.random.default_rng(8)
    eng, orc = t.Engine(cap, n), _oracle(cap)
    eng.check_on_close = True
    eng.register_params_uniform(5, 10, 60)

    def batch(general):
        slots = ((rng.zipf(1.3, n) * 2654435761) % cap).astype(np.uint32)
        if general:
            return slots, rng.integers(0, 3, n), T0 + rng.integers(0, 10**9, n)
        return slots, 1, T0 + int(rng.integers(0, 10**9))

    def pinned(slots, q, now):
        s = eng.host_alloc(n, np.uint32)
        s[:] = slots
        cols = []
        for v in (q, now):
            if isinstance(v, np.ndarray):
                a = eng.host_alloc(n, np.int64)
                a[:] = v
                cols.append(a)
            else:
                cols.append(v)
        return s, cols[0], cols[1]

    step = 0
    for general in (False, True):
        for path in ("sync", "async"):
            for nth in (1, 2, 3):
                slots, q, now = batch(general)
                before = eng.counters()
                state0 = eng.read_state(0, cap)
                eng.debug_fail_copy(nth)
                failed = False
                try:
                    if path == "sync":
                        eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
                    else:
                        s, qq, nn = pinned(slots, q, now)
                        out = t.BatchResult(**{f: eng.host_alloc(n, np.uint8 if f in ("allowed", "status") else np.int64) for f in FIELDS})
                        eng.rate_limit_batch_slots(s, registered=True, quantity=qq, now_ns=nn, out=out, async_=True)
                        eng.wait_batches(0)
                except t.TcError as err:
                    assert err.code == L.TC_E_HIP
                    failed = True
                eng.debug_fail_copy(0)
                eng.synchronize()
                if failed:
                    # (a failure in an INPUT copy: nothing was applied; in a result copy the batch WAS applied -- the
                    # documented exception -- which the oracle then has to follow)
                    after = eng.counters()
                    state1 = eng.read_state(0, cap)
                    applied = after["total"] != before["total"]
                    if not applied:
                        assert np.array_equal(state0[0], state1[0]) and np.array_equal(state0[1], state1[1])
                        assert after["batches"] == before["batches"]
                    else:
                        orc.batch_slots(slots, 5, 10, 60, q, now)
                else:
                    orc.batch_slots(slots, 5, 10, 60, q, now)   # fewer copies in this path than nth: the batch went through
                # the engine goes on, bit-exact
                slots, q, now = batch(general)
                ref = orc.batch_slots(slots, 5, 10, 60, q, now)
                assert_same(eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now), ref, f"after {path}/{general}/{nth}")
                assert_state_same(eng, orc, slots[::3])
                step += 1
    eng.close()

    # string keys: the key arena's staging copy fails
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    eng = t.Engine(20000, 8192, key_mode=True)
    eng.check_on_close = True
    orc = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    for nth in (1, 2):
        kb, ko = W.string_keys(rng.integers(0, 9000, 5000))
        before = eng.counters()
        eng.debug_fail_copy(nth)
        with pytest.raises(t.TcError) as ei:
            eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + nth)
        assert ei.value.code == L.TC_E_HIP
        eng.debug_fail_copy(0)
        after = eng.counters()
        assert (after["keys_inserted"], after["total"], after["batches"]) == (before["keys_inserted"], before["total"], before["batches"])
        kb, ko = W.string_keys(rng.integers(0, 9000, 5000))
        ref = orc.batch_keys(kb, ko, 5, 10, 60, 1, T0 + 10 + nth)
        res = eng.rate_limit_batch_keys(kb, ko, max_burst=5, count_per_period=10, period=60, quantity=1, now_ns=T0 + 10 + nth)
        for f in FIELDS:
            assert np.array_equal(getattr(res, f).astype(np.int64), getattr(ref, f).astype(np.int64)), (nth, f)
    eng.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _contender(rank, seconds, q):
    sys.path.insert(0, ROOT)
    import torch
    import throttlecrab_amd as t
    from oracle import oracle as O
    torch.cuda.set_device(0)
    cap, n = 200_000, 1 << 18
    rng = np.random.default_rng(100 + rank)
    eng, orc = t.Engine(cap, n), O.DenseOracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(200_000, 10**9, 1)   # burst 200 000, 1 ns apart: runs of 100 000 allowed requests
    tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
    from throttlecrab_amd import sharded
    host_counts = eng.host_alloc(3, np.uint32)
    host_counts[:] = 0
    t_end, rounds, bad = time.time() + seconds, 0, 0
    # fillers take wave slots, LDS and whole CUs away from the kernels that wait for other workgroups: half of the
    # CUs for 3 ms, every other CU, or every CU's LDS -- launched just before a batch, still resident while its kernels run
    fillers = os.environ.get("TC_STRESS_FILLERS", "1") == "1"
    masks = [[0xFFFFFFFF] * 4 + [0] * 4, [0x55555555] * 8, None]
    while time.time() < t_end:
        kind = rounds % 4
        if fillers and rounds % 3 != 2:
            m = masks[(rounds // 4) % 3]
            eng.debug_occupy(blocks=256 if m is None else 1024, microseconds=3000 if m is not None else 800, cu_mask=m,
                             lds_bytes=96 * 1024 if m is None else 0)
        if kind == 3:
            # a global batch of 2 shards routed on the grouping streams (one-pass router: tiles chained by look-back),
            # the count polled from pinned memory, what this shard owns evaluated as a pipelined batch
            gids = rng.integers(0, 2 * cap, 2 * n - 5000).astype(np.uint32)
            owner, slot = sharded.route(gids, 2, cap)
            want = slot[owner == rank]
            base = T0 + rounds * 10**9
            ref = orc.batch_slots(want, 200_000, 10**9, 1, 1, base)
            g = torch.from_numpy(gids.astype(np.int32)).cuda()
            torch.cuda.synchronize()
            slots_d, _, _ = eng.route_batch(g, 2, only=rank, ahead=True, host_counts=host_counts, tag=rounds + 1)
            while int(host_counts[2]) != rounds + 1:
                pass
            mine = int(host_counts[rank])
            bad += int(mine != len(want))
            res = eng.rate_limit_batch_slots(slots_d[:mine], registered=True, quantity=1, now_ns=base, want=("allowed", "remaining"), inputs_ready=True)
            torch.cuda.synchronize()
            if mine == len(want):
                bad += int((res.allowed.cpu().numpy() != ref.allowed).sum()) + int((res.remaining.cpu().numpy() != ref.remaining).sum())
            rounds += 1
            continue
        slots = rng.integers(0, cap, n).astype(np.uint32)
        hot = rng.random(n) < 0.4
        slots[hot] = 7 + rank            # ~105 000 requests of one key: >1 600 waves, all allowed
        base = T0 + rounds * 10**9
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        if kind == 0:    # general path: per-request timestamps, the chain carries the state through every wave
            now = base + np.sort(rng.integers(0, 10**6, n))
            ref = orc.batch_slots(slots, 200_000, 10**9, 1, 1, now)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=tt(now), want=("allowed", "remaining"), inputs_ready=True)
        elif kind == 1:  # uniform, pipelined: radix look-back + direct stores behind earlier rows
            ref = orc.batch_slots(slots, 200_000, 10**9, 1, 1, base)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=base, want=("allowed", "remaining"), inputs_ready=True)
        else:            # uniform, in order
            ref = orc.batch_slots(slots, 200_000, 10**9, 1, 1, base)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=base, want=("allowed", "remaining"))
        torch.cuda.synchronize()
        bad += int((res.allowed.cpu().numpy() != ref.allowed).sum()) + int((res.remaining.cpu().numpy() != ref.remaining).sum())
        rounds += 1
    q.put((rank, rounds, bad, eng.selfcheck()))
    eng.close()


def test_two_processes_contend_for_one_device():
    import torch.multiprocessing as mp
    seconds = float(os.environ.get("TC_STRESS_SECONDS", "20"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_contender, args=(r, seconds, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=seconds * 10 + 300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, rounds, bad, viol in got:
        assert rounds >= 4, (rank, rounds)
        assert bad == 0, (rank, bad)
        assert viol == 0, f"rank {rank}: the spin watchdog fired {viol} times"


@pytest.mark.parametrize("sync_batch", [True, False], ids=["host_batch", "device_batch"])
def test_a_tripped_watchdog_fails_the_engine_loudly(sync_batch):
    """a wait that gives up must not leave silently wrong results behind.  tc_debug_break_wait withholds
    ONE "row has read its cells" announcement; the owner of a key whose requests span several rows then waits until the
    watchdog expires (2 s), the kernel raises the engine's poison word, and the batch call (synchronous batch) or the
    next call / tc_synchronize (asynchronous batch) returns TC_E_INVARIANT -- as does everything after it."""
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import _lib as L
    eng = t.Engine(1000, 1 << 16)
    eng.register_params_uniform(1000, 1000, 60)
    slots = np.zeros(2000, np.uint32)  # one key, 2000 allowed requests: the run crosses 31 rows
    slots[1000:] = 7
    # a healthy batch first: no trip, no poison
    r = eng.rate_limit_batch_slots(slots.copy(), registered=True, quantity=1, now_ns=T0, want=("allowed",))
    assert r.allowed.all() and eng.selfcheck() == 0
    eng.debug_break_wait(True)
    t0 = time.time()
    if sync_batch:
        with pytest.raises(t.engine.TcError) as ei:
            eng.rate_limit_batch_slots(np.full(2000, 5, np.uint32), registered=True, quantity=1, now_ns=T0 + 1, want=("allowed",))  # a fresh key
        assert ei.value.code == L.TC_E_INVARIANT
    else:
        d = torch.full((8000,), 5, dtype=torch.int32, device="cuda")  # a fresh key: 1000 allowed requests, the owner sits in row 15
        eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=T0 + 1, want=("allowed",))  # enqueued: returns at once
        with pytest.raises(t.engine.TcError) as ei:
            eng.synchronize()
        assert ei.value.code == L.TC_E_INVARIANT
    assert 1.0 < time.time() - t0 < 30.0  # the watchdog's two seconds, not a hang
    assert eng.selfcheck() >= 1
    for call in (lambda: eng.rate_limit_batch_slots(slots, registered=True, quantity=1, now_ns=T0 + 2, want=("allowed",)),
                 lambda: eng.counters(), lambda: eng.sweep_expired(T0), lambda: eng.synchronize()):
        with pytest.raises(t.engine.TcError) as ei:
            call()
        assert ei.value.code == L.TC_E_INVARIANT  # sticky
    eng.close()
    # a fresh engine is unaffected
    e2 = t.Engine(1000, 1 << 16)
    e2.register_params_uniform(5, 10, 60)
    assert e2.rate_limit_batch_slots(np.arange(10, dtype=np.uint32), registered=True, quantity=1, now_ns=T0, want=("allowed",)).allowed.all()
    e2.close()


@pytest.mark.parametrize("filler", ["half_the_cus", "every_other_cu", "all_lds", "oversubscribed"])
def test_waiting_kernels_make_progress_beside_fillers(filler):
    """the look-back of the LSD passes, the direct stores of k_eval_sorted, the chain of k_eval_general and the
    one-pass router wait for workgroups dispatched EARLIER -- an assumption about the dispatcher, not a promise of HIP.  What
    can be tested is that starving them does not break it: a filler kernel (tc_debug_occupy) holds half of the CUs, every other
    CU, most of every CU's LDS, or more workgroups than the chip has room for, for milliseconds, while batches with a run of
    ~105 000 allowed requests of one key (1 600 rows behind one owner), per-request timestamps (the chain) and a skewed slot
    column (LSD passes) run pipelined.  Results == oracle, watchdog silent."""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import sharded
    cap, n = 200_000, 1 << 18
    rng = np.random.default_rng(41)
    eng, orc = t.Engine(cap, n), O.DenseOracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(200_000, 10**9, 1)
    tt = lambda a: torch.from_numpy(np.asarray(a).astype(np.int64)).cuda()
    spec = {"half_the_cus": dict(blocks=2048, microseconds=4000, cu_mask=[0xFFFFFFFF] * 4 + [0] * 4),
            "every_other_cu": dict(blocks=2048, microseconds=4000, cu_mask=[0x55555555] * 8),
            "all_lds": dict(blocks=256, microseconds=4000, lds_bytes=128 * 1024),
            "oversubscribed": dict(blocks=8192, microseconds=600)}[filler]
    host_counts = eng.host_alloc(3, np.uint32)
    host_counts[:] = 0
    pending = []
    for rnd in range(8):
        slots = rng.integers(0, cap, n).astype(np.uint32)
        slots[rng.random(n) < 0.4] = 11
        base = T0 + rnd * 10**9
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        eng.debug_occupy(**spec)                      # resident while the batch's kernels are dispatched
        if rnd % 4 == 0:
            now = base + np.sort(rng.integers(0, 10**6, n))
            ref = orc.batch_slots(slots, 200_000, 10**9, 1, 1, now)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=tt(now), want=("allowed", "remaining"), inputs_ready=True)
        elif rnd % 4 == 3:
            gids = rng.integers(0, 2 * cap, n).astype(np.uint32)
            owner, slot = sharded.route(gids, 2, cap)
            want = slot[owner == 0]
            ref = orc.batch_slots(want, 200_000, 10**9, 1, 1, base)
            g = torch.from_numpy(gids.astype(np.int32)).cuda()
            torch.cuda.synchronize()
            slots_d, _, _ = eng.route_batch(g, 2, only=0, ahead=True, host_counts=host_counts, tag=rnd + 1)
            while int(host_counts[2]) != rnd + 1:
                pass
            mine = int(host_counts[0])
            assert mine == len(want)
            res = eng.rate_limit_batch_slots(slots_d[:mine], registered=True, quantity=1, now_ns=base, want=("allowed", "remaining"), inputs_ready=True)
        else:
            ref = orc.batch_slots(slots, 200_000, 10**9, 1, 1, base)
            res = eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=base, want=("allowed", "remaining"), inputs_ready=bool(rnd % 2))
        pending.append((res, ref, d))
    torch.cuda.synchronize()
    for rnd, (res, ref, _) in enumerate(pending):
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed), (filler, rnd)
        assert np.array_equal(res.remaining.cpu().numpy(), ref.remaining), (filler, rnd)
    assert eng.selfcheck() == 0, "the spin watchdog fired"
    eng.close()


def test_pipelined_batches_do_not_depend_on_what_the_process_did_before():
    """Which hardware queue -- and which dispatch pipe -- a HIP stream lands on depends on everything the process created before.
    With GPU work on the caller's stream BEFORE the engine creates its grouping streams (any real application; not bench.py),
    the third grouping stream used to land on the main stream's pipe: its kernels were held back while an evaluation still
    handed out blocks, and pipelined 1 Mi batches took 104 us instead of 42 -- slower than in order (64).  The engine now drops
    candidates that collide with the main stream (k_probe_occupy / k_probe_stamp).  Here: after such work, pipelined batches
    must beat in-order ones on a fresh engine each (a ratio, not a time: 0.66 when healthy, 1.6 when not), results exact."""
    import time

    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    N, B = 4_000_000, 1 << 20
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.zeros(1 << 20, device=dev).sum().item()   # the caller's stream takes its queue first
        host = [W.uniform_slots(N, B, seed=2, start=i * B) for i in range(4)]
        d = [torch.from_numpy(h.astype(np.int32)).to(dev) for h in host]
        per = {}
        for piped in (True, False):
            eng = t.Engine(N, B, fixed_params=True)
            eng.use_torch_stream()
            eng.register_params_uniform(*W.REF_PARAMS)
            outs = [t.BatchResult(allowed=torch.empty(B, dtype=torch.uint8, device=dev)) for _ in range(8)]
            run = lambda i: eng.rate_limit_batch_slots(d[i % 4], registered=True, now_ns=W.T0_NS + i * 1000, want=("allowed",), out=outs[i % 8],
                                                       inputs_ready=piped, outputs_idle=piped)
            for i in range(12):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(60):
                run(12 + i)
            torch.cuda.synchronize()
            per[piped] = (time.perf_counter() - t0) / 60
            if piped:   # the last batch's decisions against the oracle applying all 72 batches
                orc = O.DenseOracle(N)
                for i in range(72):
                    ref = orc.batch_slots(host[i % 4], *W.REF_PARAMS, 1, W.T0_NS + i * 1000)
                assert np.array_equal(outs[71 % 8].allowed.cpu().numpy(), ref.allowed)
            assert eng.selfcheck() == 0
            eng.close()
    assert per[True] < 0.9 * per[False], per


_INFO_CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["TC_ROOT"])
import numpy as np, torch
import throttlecrab_amd as t
from oracle import oracle as O
from throttlecrab_amd import workload as W
cap, n = 200_000, 1 << 16
eng, orc = t.Engine(cap, n, fixed_params=True), O.DenseOracle(cap)
eng.use_torch_stream()
eng.register_params_uniform(*W.REF_PARAMS)
before = eng.info()
ok = True
for b in range(6):
    sl = W.uniform_slots(cap, n, seed=5, start=b * n)
    ref = orc.batch_slots(sl, *W.REF_PARAMS, 1, W.T0_NS + b * 10**6)
    res = eng.rate_limit_batch_slots(torch.from_numpy(sl.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=W.T0_NS + b * 10**6,
                                     want=("allowed",), inputs_ready=True)
    torch.cuda.synchronize()
    ok = ok and bool(np.array_equal(res.allowed.cpu().numpy(), ref.allowed))
print(json.dumps({"before": before, "after": eng.info(), "exact": ok, "selfcheck": eng.selfcheck()}))
"""


@pytest.mark.parametrize("queues", ["8", "2"])
def test_the_engine_says_when_its_pipeline_is_degraded(queues):
    """with too few hardware queues the side streams cannot run beside the main stream; pipelined batches then run
    in order -- same results, 1.5-2.5 x slower, no error.  tc_engine_info_get must say so (and say "healthy" when it is)."""
    import json
    import subprocess
    env = dict(os.environ, GPU_MAX_HW_QUEUES=queues, TC_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", _INFO_CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["exact"] and d["selfcheck"] == 0
    assert d["before"]["side_streams_probed"] == 0 and d["before"]["grouping_path"] == "none yet"
    a = d["after"]
    assert a["side_streams_probed"] == 1 and a["grouping_streams_wanted"] == 3 and a["batches"] == 6 and a["scratch_sets"] == 6
    assert a["grouping_path"] in ("range path", "LSD passes") and a["range_path_possible"] == 1 and a["range_hint_requests"] == 1 << 16
    assert a["candidates_tried"] >= a["grouping_streams"]
    if queues == "8":
        assert a["grouping_streams"] == 3 and a["pipelining_degraded"] == 0, a
    else:   # main + one more queue at most: not three grouping streams that run beside the main stream
        assert a["grouping_streams"] < 3 and a["pipelining_degraded"] == 1 and a["rejected_same_queue"] + a["rejected_same_pipe"] > 0, a


def test_a_second_engine_takes_the_probed_streams_from_the_pool():
    """probing sixteen stream candidates costs 3-8 ms of GPU time on every engine's first pipelined batch.  The
    streams that passed stay in a per-process pool.  An engine that runs on a stream of its own leaves the whole set behind (main
    stream + grouping streams) and the next such engine takes it as it is; engines that share a stream of the caller's share
    its probed grouping streams, while they live and afterwards.  tc_engine_info says so (probes_pooled); results stay exact."""
    import torch

    import throttlecrab_amd as t
    from tests.test_gpu_slots import T0, _oracle
    cap, n, plan = 300_000, 1 << 16, (5, 10, 60)
    rng = np.random.default_rng(4)

    def run(eng, orc, base):
        held = []
        for i in range(5):
            slots = rng.integers(0, cap, n).astype(np.uint32)
            ref = orc.batch_slots(slots, *plan, 1, T0 + (base + i) * 10**8)
            d = torch.from_numpy(slots.astype(np.int32)).cuda()
            torch.cuda.synchronize()
            held.append((eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=T0 + (base + i) * 10**8, want=("allowed",), inputs_ready=True), ref, d))
        eng.synchronize()
        torch.cuda.synchronize()
        for res, ref, _ in held:
            assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed)

    def make(stream=None):
        e = t.Engine(cap, n, fixed_params=True)
        if stream is not None:
            e.set_stream(stream.cuda_stream)
        e.register_params_uniform(*plan)
        return e, _oracle(cap)

    # engines on streams of their own: two beside each other probe a set each, their successors take the sets over
    a, oa = make()
    b, ob = make()
    run(a, oa, 0)
    run(b, ob, 0)
    assert a.info()["grouping_streams"] == 3 and b.info()["grouping_streams"] == 3
    run(a, oa, 5)
    a.close()
    b.close()
    c, oc = make()
    d, od = make()
    run(c, oc, 0)
    run(d, od, 0)
    for eng in (c, d):
        info = eng.info()
        assert info["probes_pooled"] == 1 and info["grouping_streams"] == 3 and info["pipelining_degraded"] == 0, info
        assert eng.selfcheck() == 0
    c.close()
    d.close()
    # engines on ONE stream of the caller's share its probed grouping streams
    side = torch.cuda.Stream()
    f, of = make(side)
    run(f, of, 0)
    g, og = make(side)
    run(g, og, 0)
    ig = g.info()
    assert ig["probes_pooled"] == 1 and ig["grouping_streams"] == 3, ig
    run(f, of, 5)
    assert f.selfcheck() == 0 and g.selfcheck() == 0
    f.close()
    g.close()
