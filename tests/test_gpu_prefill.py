"""TC_B_OUTPUTS_IDLE (decision bytes preset on the grouping stream, the evaluation stores the minority decision only)
and the lean evaluation kernel: same decisions as the oracle, batch by batch, whatever the fill value was -- streams
whose majority flips, sizes that are not multiples of 16, unaligned output arrays, both resident-state layouts."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
PLAN = (100, 1000, 3600)


def _run(n_keys, sizes, make_slots, fixed, plan=PLAN, dt_ns=1_000_000, offset=0, env=None, monkeypatch=None, sync_each=False):
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    eng = t.Engine(n_keys, max(sizes), fixed_params=fixed)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*plan)
    orc = O.DenseOracle(n_keys)
    ring = [torch.full((max(sizes) + 64,), 7, dtype=torch.uint8, device="cuda") for _ in range(len(sizes))]
    pending = []
    for b, n in enumerate(sizes):
        slots = make_slots(b, n)
        now = T0 + b * dt_ns
        ref = orc.batch_slots(slots, *plan, 1, now)
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        out = t.BatchResult(allowed=ring[b][offset:offset + n])
        eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=now, want=("allowed",), out=out, inputs_ready=True,
                                   outputs_idle=True)
        pending.append((b, n, d, ref))
        if sync_each:  # the device's hint reaches the host before the next call: the fill value follows the stream
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for b, n, _, ref in pending:
        got = ring[b].cpu().numpy()
        bad = np.nonzero(got[offset:offset + n] != ref.allowed.astype(np.uint8))[0]
        assert bad.size == 0, f"batch {b} (n={n}): decisions differ at {bad[:8]}"
        assert (got[:offset] == 7).all() and (got[offset + n:] == 7).all(), f"batch {b}: bytes outside the batch were written"
    c = eng.counters()
    assert c["total"] == sum(sizes)
    eng.close()
    return c


@pytest.mark.parametrize("sync_each", [False, True], ids=["host_ahead", "hint_follows"])
@pytest.mark.parametrize("fixed", [False, True], ids=["wide", "fixed"])
def test_majority_flips_from_allowed_to_denied(fixed, sync_each):
    """64 hot keys, burst 100: the first batches are mostly allowed, then the keys run dry and nearly everything is
    denied -- the hint (and with it the fill value) flips on the way; every batch must match the oracle."""
    rng = np.random.default_rng(5)
    sizes = [20000] * 14

    def slots(b, n):
        return rng.integers(0, 64, n).astype(np.uint32)
    c = _run(1000, sizes, slots, fixed, sync_each=sync_each)
    assert c["allowed"] > 0 and c["denied"] > c["allowed"]


@pytest.mark.parametrize("fixed", [False, True], ids=["wide", "fixed"])
def test_mixed_decisions_uniform_and_skewed(fixed):
    from throttlecrab_amd import workload as W
    z = W.Zipf(50_000)
    rng = np.random.default_rng(9)
    sizes = [65536, 65536, 40000, 65536, 16, 1, 65536, 33333]

    def slots(b, n):
        return z.slots(n, start=b * 65536) if b % 2 else rng.integers(0, 50_000, n).astype(np.uint32)
    _run(50_000, sizes, slots, fixed, plan=(5, 10, 60), dt_ns=400_000_000)


@pytest.mark.parametrize("offset", [0, 1, 5, 16, 31])
def test_unaligned_output_and_ragged_sizes(offset):
    rng = np.random.default_rng(offset)
    sizes = [1, 15, 16, 17, 4097, 20000 + offset, 65535]

    def slots(b, n):
        return rng.integers(0, 3000, n).astype(np.uint32)
    _run(3000, sizes, slots, True, plan=(3, 6, 60), dt_ns=2_000_000_000, offset=offset)


@pytest.mark.parametrize("env", [{"TCGPU_PREFILL": "0"}, {"TCGPU_EVAL_LEAN": "0"}, {"TCGPU_STOP_EVENTS": "0"}, {"TCGPU_EVAL_ITEMS": "1"}],
                         ids=["no_prefill", "no_lean", "no_stop_events", "items1"])
def test_same_results_with_each_knob_off(env, monkeypatch):
    rng = np.random.default_rng(3)
    sizes = [30000] * 8

    def slots(b, n):
        return rng.integers(0, 200, n).astype(np.uint32)
    _run(5000, sizes, slots, True, env=env, monkeypatch=monkeypatch)


def test_out_of_range_slots_and_unregistered_plans_keep_their_zero():
    """requests that get a status (slot >= capacity) are `not allowed`, whatever the fill value"""
    import torch

    import throttlecrab_amd as t
    eng = t.Engine(1000, 1 << 16, fixed_params=True)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*PLAN)
    rng = np.random.default_rng(1)
    for b in range(4):  # (the hint says "allowed" after the first batch: fill value 1)
        slots = rng.integers(0, 1000, 50000).astype(np.uint32)
        bad = rng.random(50000) < 0.1
        slots[bad] = 1000 + rng.integers(0, 100, int(bad.sum()))
        out = t.BatchResult(allowed=torch.full((50000,), 9, dtype=torch.uint8, device="cuda"))
        eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=T0 + b * 10**12,
                                   want=("allowed",), out=out, inputs_ready=True, outputs_idle=True)
        torch.cuda.synchronize()
        got = out.allowed.cpu().numpy()
        assert (got[bad] == 0).all() and (got[~bad] == 1).all(), b
    eng.close()


@pytest.mark.parametrize("sync_each", [False, True], ids=["host_ahead", "hint_follows"])
@pytest.mark.parametrize("fixed", [False, True], ids=["wide", "fixed"])
def test_general_batches_with_preset_decision_bytes(fixed, sync_each):
    """per-request timestamps (k_eval_general) with TC_B_OUTPUTS_IDLE: hot keys that run dry (the majority flips),
    out-of-range slots (status != OK keeps its 0 under a fill value of 1), ragged sizes, an unaligned output array"""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    n_keys, plan = 2000, (20, 100, 60)
    eng = t.Engine(n_keys, 1 << 16, fixed_params=fixed)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*plan)
    orc = O.DenseOracle(n_keys)
    rng = np.random.default_rng(11)
    sizes = [30000, 65536, 1, 17, 40001, 30000, 30000, 65535, 30000, 30000]
    ring = [torch.full((70000,), 7, dtype=torch.uint8, device="cuda") for _ in sizes]
    pending, t_ns = [], T0
    for b, n in enumerate(sizes):
        slots = np.where(rng.random(n) < 0.7, rng.integers(0, 40, n), rng.integers(0, n_keys, n)).astype(np.uint32)
        bad = rng.random(n) < 0.02
        slots[bad] = n_keys + 5
        now = t_ns + np.sort(rng.integers(0, 2_000_000, n)).astype(np.int64)
        t_ns += 2_000_000 if b != 5 else 120 * 10**9   # (a long pause: the keys refill, the majority flips back)
        ref = orc.batch_slots(slots, *plan, 1, now)
        out = t.BatchResult(allowed=ring[b][3:3 + n])
        keep = (torch.from_numpy(slots.astype(np.int32)).cuda(), torch.from_numpy(now).cuda())
        eng.rate_limit_batch_slots(keep[0], registered=True, quantity=1, now_ns=keep[1], want=("allowed",), out=out, inputs_ready=True,
                                   outputs_idle=True)
        pending.append((b, n, keep, ref, bad))
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    seen = set()
    for b, n, _, ref, bad in pending:
        got = ring[b].cpu().numpy()
        want = ref.allowed.astype(np.uint8)
        assert (want[bad] == 0).all()
        diff = np.nonzero(got[3:3 + n] != want)[0]
        assert diff.size == 0, f"batch {b} (n={n}): decisions differ at {diff[:8]}"
        assert (got[:3] == 7).all() and (got[3 + n:] == 7).all(), f"batch {b}: bytes outside the batch were written"
        seen.add(bool(want.mean() > 0.5))
    assert seen == {True, False}, "the stream must have batches of either majority"
    eng.close()


def test_result_set_reused_for_a_larger_batch_gets_new_arrays():
    """the Python wrapper never lets the library write past the end of a result array that an earlier, smaller batch
    allocated (an owner's share of a routed global batch changes from step to step)"""
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    eng = t.Engine(10_000, 1 << 16)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*PLAN)
    orc = O.DenseOracle(10_000)
    rng = np.random.default_rng(3)
    out = t.BatchResult()
    guard = None
    for b, n in enumerate([1000, 999, 4097, 65536, 17]):
        slots = rng.integers(0, 10_000, n).astype(np.uint32)
        ref = orc.batch_slots(slots, *PLAN, 1, T0 + b)
        before = out.allowed
        eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=T0 + b,
                                   want=("allowed",), out=out)
        torch.cuda.synchronize()
        assert out.allowed.numel() >= n
        if before is not None and before.numel() >= n:
            assert out.allowed is before  # large enough: reused
        assert (out.allowed.cpu().numpy()[:n] == ref.allowed.astype(np.uint8)).all()
        guard = out.allowed
    assert guard is not None
    eng.close()
