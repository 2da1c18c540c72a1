"""TC_B_COLUMNS_READY: a general batch's timestamps ride through the last radix pass (rs::k_onesweep<.., CARRY>) and the
evaluation reads them in its own order.  Same results as the oracle and as the gather by request index, request by
request: hot keys whose runs cross waves and blocks, non-monotone timestamps, per-request quantities beside the carried
column, every pass count (1, 2, 3 radix passes), ragged sizes, both resident-state layouts, full results."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
PLAN = (5, 10, 60)
FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")


def _stream(rng, n_keys, n, b):
    slots = rng.integers(0, n_keys, n).astype(np.uint32)
    slots[: n // 6] = rng.integers(0, min(n_keys, 48), n // 6)       # hot keys
    rng.shuffle(slots)
    now = T0 + b * 700_000_000 + np.sort(rng.integers(0, 500_000_000, n)).astype(np.int64)
    back = rng.random(n) < 0.03
    now[back] -= rng.integers(0, 10**9, int(back.sum()))             # a clock that is not monotone
    return slots, now


@pytest.mark.parametrize("n_keys", [200, 40_000, 3_000_000], ids=["1pass", "2passes", "3passes"])
@pytest.mark.parametrize("fixed", [False, True], ids=["wide", "fixed"])
def test_carried_timestamps_match_the_oracle(fixed, n_keys):
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    sizes = [50_000, 1, 63, 4097, 131_072, 77_777]
    eng = t.Engine(n_keys, max(sizes), fixed_params=fixed)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*PLAN)
    orc = O.DenseOracle(n_keys)
    rng = np.random.default_rng(n_keys + fixed)
    outs, refs, keep = [], [], []
    for b, n in enumerate(sizes):
        slots, now = _stream(rng, n_keys, n, b)
        want = FIELDS if b % 2 else ("allowed",)
        refs.append((orc.batch_slots(slots, *PLAN, 1, now), want, n))
        d, dn = torch.from_numpy(slots.astype(np.int32)).cuda(), torch.from_numpy(now).cuda()
        keep.append((d, dn))
        outs.append(eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=dn, want=want, inputs_ready=True,
                                               columns_ready=True, outputs_idle=(want == ("allowed",))))
    torch.cuda.synchronize()
    for b, (res, (ref, want, n)) in enumerate(zip(outs, refs)):
        for f in want:
            got = getattr(res, f).cpu().numpy()[:n]
            exp = getattr(ref, f)
            assert (got == exp.astype(got.dtype)).all(), f"batch {b} field {f} differs at {np.nonzero(got != exp.astype(got.dtype))[0][:6]}"
    from tests.test_gpu_slots import assert_state_same
    assert_state_same(eng, orc, np.arange(min(n_keys, 3000)))
    eng.close()


def test_per_request_quantity_and_rate_beside_the_carried_column():
    """only `now` is carried; quantity and the rate triple stay gathered by request index"""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    n_keys, n = 5000, 60_000
    eng = t.Engine(n_keys, n)
    eng.check_on_close = True
    eng.use_torch_stream()
    orc = O.DenseOracle(n_keys)
    rng = np.random.default_rng(7)
    for b in range(4):
        slots, now = _stream(rng, n_keys, n, b)
        q = rng.choice(np.array([1, 1, 2, 0, -1, 3], dtype=np.int64), n)
        burst = rng.choice(np.array([3, 10, 1, 0], dtype=np.int64), n)
        count = np.full(n, 20, np.int64)
        period = rng.choice(np.array([60, 1], dtype=np.int64), n)
        ref = orc.batch_slots(slots, burst, count, period, q, now)
        cols = [torch.from_numpy(a).cuda() for a in (slots.astype(np.int32), burst, count, period, q, now)]
        torch.cuda.synchronize()
        res = eng.rate_limit_batch_slots(cols[0], max_burst=cols[1], count_per_period=cols[2], period=cols[3], quantity=cols[4],
                                         now_ns=cols[5], want=FIELDS, inputs_ready=True, columns_ready=True)
        torch.cuda.synchronize()
        for f in FIELDS:
            got = getattr(res, f).cpu().numpy()
            assert (got == getattr(ref, f).astype(got.dtype)).all(), f"batch {b} field {f}"
    eng.close()


def test_same_results_with_and_without_the_flag_at_full_size():
    """10 M keys, 1 Mi Zipf requests with a timestamp each: carried vs gathered, decision by decision, and the state"""
    import torch

    import throttlecrab_amd as t
    from throttlecrab_amd import workload as W
    n_keys, n = 10_000_000, 1 << 20
    z = W.Zipf(n_keys)
    engs = [t.Engine(n_keys, n, fixed_params=True) for _ in range(2)]
    for e in engs:
        e.check_on_close = True
        e.use_torch_stream()
        e.register_params_uniform(*W.REF_PARAMS)
    for b in range(4):
        d = torch.from_numpy(z.slots(n, start=b * n).astype(np.int32)).cuda()
        dn = torch.arange(n, dtype=torch.int64, device="cuda") + (W.T0_NS + b * 1_000_000)
        torch.cuda.synchronize()
        res = [e.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=dn, want=("allowed",), inputs_ready=True,
                                        columns_ready=bool(i)) for i, e in enumerate(engs)]
        torch.cuda.synchronize()
        assert torch.equal(res[0].allowed, res[1].allowed), f"batch {b}"
    st = [e.read_state(0, 100_000) for e in engs]
    assert (st[0][0] == st[1][0]).all() and (st[0][1] == st[1][1]).all()
    c = [e.counters() for e in engs]
    assert c[0]["allowed"] == c[1]["allowed"] and 0 < c[0]["denied"]
    for e in engs:
        e.close()
