"""The configuration bench.py TIMES, pinned to the oracle at the headline's size.

bench.py's headline runs 10 M keys / 1 Mi requests per batch on the 8-byte layout with one registered plan,
TC_B_INPUTS_READY | TC_B_OUTPUTS_IDLE, want=("allowed",), a ring of 8 result arrays and no host synchronisation between
batches: that is k_eval_sorted_lean<2, true> behind the three grouping streams, with preset decision bytes.  The other
full-size tests run other variants (wide layout, in order, all outputs).  Here: 25 batches of bench.py's own uniform
(seed 2) and Zipf (seed 3) streams back to back, every batch's decision bytes, the counters and the whole resident
state (tc_read_state over all 10 M keys) against the oracle applying the same requests one by one
(rate_limiter.rs:147-205 in queue order, throttlecrab-server/src/actor.rs:217-236); tc_selfcheck must stay 0.
The same for the 16-byte layout, for per-request timestamps (k_eval_general, bench.py's make_nows columns) and for
per-key rate plans (the class_by_slot path: rate_limiter.rs:102-123 takes the triple per call, the engine keeps it per key).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAP, N, BATCHES, RING = 10_000_000, 1 << 20, 25, 8


def _streams():
    import bench
    return {"uniform": bench.make_batches("uniform", CAP, N, BATCHES), "zipf": bench.make_batches("zipf", CAP, N, BATCHES)}


@pytest.fixture(scope="module")
def streams():
    return _streams()


def _plans_of(kind, cap):
    """per-key plans: tier of a slot = a hash of the slot (bench.py's per_key_plans leg uses the same function)"""
    import bench
    return bench.plan_tiers(kind, cap)


def _run(streams, stream, layout, general, plans, own_stream):
    import torch

    import bench
    import throttlecrab_amd as t
    from oracle import oracle as O
    from throttlecrab_amd import workload as W
    dev = torch.device("cuda:0")
    host = streams[stream]
    ctx = torch.cuda.stream(torch.cuda.Stream(device=dev)) if not own_stream else torch.cuda.stream(torch.cuda.default_stream(dev))
    with ctx:
        eng = t.Engine(CAP, N, fixed_params=(layout == "fixed"))
        eng.use_torch_stream()   # (under torch's default stream this is "the engine's own stream": exactly bench.py's N = 1 set-up)
        if plans is None:
            eng.register_params_uniform(*W.REF_PARAMS)
            per_slot = None
        else:
            tiers, tier_of = _plans_of(plans, CAP)
            per_slot = tiers[tier_of]                      # [CAP, 3]
            eng.register_params(per_slot[:, 0], per_slot[:, 1], per_slot[:, 2])
        d_batches = [torch.from_numpy(b.astype(np.int32)).to(dev) for b in host]
        nows = bench.make_nows(dev, N, BATCHES, W.T0_NS) if general else None
        # own stream: nothing of torch's is ordered against the engine, so every batch keeps its own array (a ring as long as
        # the run); shared stream: bench.py's ring of 8, each batch's bytes copied aside ON the engine's stream before the
        # ring entry comes round again
        ring = [t.BatchResult(allowed=torch.empty(N, dtype=torch.uint8, device=dev)) for _ in range(BATCHES if own_stream else RING)]
        kept = torch.empty((BATCHES, N), dtype=torch.uint8, device=dev) if not own_stream else None
        torch.cuda.synchronize()
        for i in range(BATCHES):            # back to back: no host synchronisation inside
            eng.rate_limit_batch_slots(d_batches[i], registered=True, quantity=1, now_ns=bench.now_of(W.T0_NS, i, nows), want=("allowed",),
                                       out=ring[i % len(ring)], inputs_ready=True, outputs_idle=True)
            if kept is not None:
                kept[i].copy_(ring[i % len(ring)].allowed)
        torch.cuda.synchronize()
        got = [(kept[i] if kept is not None else ring[i].allowed).cpu().numpy() for i in range(BATCHES)]
        counters = eng.counters()
        assert eng.selfcheck() == 0
        tat, exp = eng.read_state(0, CAP)
        eng.close()
    orc = O.DenseOracle(CAP)
    allowed, th = 0, O.host_threads()
    for i in range(BATCHES):
        sl = host[i]
        now = (W.T0_NS + i * 1_000_000 + np.arange(N, dtype=np.int64)) if general else W.T0_NS + i * 1_000_000
        if per_slot is None:
            ref = orc.batch_slots(sl, *W.REF_PARAMS, 1, now, threads=th)
        else:
            pr = per_slot[sl]
            ref = orc.batch_slots(sl, pr[:, 0], pr[:, 1], pr[:, 2], 1, now, threads=th)
        assert not ref.status.any()
        bad = np.flatnonzero(got[i] != ref.allowed)
        assert bad.size == 0, f"{stream}/{layout} batch {i}: {bad.size} decisions differ, first at request {bad[:5]} (slots {sl[bad[:5]]})"
        allowed += int(ref.allowed.sum())
    assert counters["allowed"] == allowed and counters["denied"] == BATCHES * N - allowed and counters["errors"] == 0
    otat, oexp, occ = orc.dump()
    assert not exp[~occ].any(), "a key the oracle never wrote is occupied"
    assert np.array_equal(tat[occ], otat[occ]) and np.array_equal(exp[occ], oexp[occ]), "resident state differs"
    return allowed


@pytest.mark.parametrize("own_stream", [True, False], ids=["own_stream_ring25", "torch_stream_ring8"])
@pytest.mark.parametrize("stream", ["uniform", "zipf"])
def test_headline_configuration_matches_the_oracle(streams, stream, own_stream):
    """fixed layout, lean evaluation, preset decision bytes, pipelined, 25 unsynchronised batches"""
    allowed = _run(streams, stream, "fixed", False, None, own_stream)
    if stream == "zipf":
        assert allowed < BATCHES * N   # the hot keys run out of burst: the minority-decision stores are exercised too


@pytest.mark.parametrize("stream", ["uniform", "zipf"])
def test_wide_layout_timed_configuration(streams, stream):
    _run(streams, stream, "wide", False, None, True)


@pytest.mark.parametrize("stream", ["uniform", "zipf"])
def test_general_batches_timed_configuration(streams, stream):
    """a timestamp per request (bench.py --workload general / general_zipf): k_eval_general behind the same pipeline"""
    _run(streams, stream, "fixed", True, None, True)


@pytest.mark.parametrize("plans", ["tiers4", "tiers1000"])
@pytest.mark.parametrize("stream", ["uniform", "zipf"])
def test_per_key_plans_timed_configuration(streams, stream, plans):
    """every key carries its own (burst, count, period): the evaluation reads rate_id[] and the plan dictionary"""
    _run(streams, stream, "fixed", False, plans, True)
