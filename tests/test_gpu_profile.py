"""tc_profile_enable / tc_profile_read (the per-stage kernel times bench.py's roofline block is built from): the timed
launches carry their start/stop events on the dispatch packet itself (TC_LAUNCH_T), and profiling must neither change a
result nor lose a launch -- pipelined and in order, uniform and general batches, both resident-state layouts, and the
marker-event fallback (TCGPU_PROF_MARKERS=1)."""
import numpy as np
import pytest

from tests import kat

pytestmark = pytest.mark.gpu
T0 = kat.load()["t0_ns"]
PLAN = (5, 10, 60)


def _run(fixed, general, piped, monkeypatch=None, markers=False, n_keys=200_000, n=1 << 17, batches=6, hot=False):
    import torch

    import throttlecrab_amd as t
    from oracle import oracle as O
    if markers:
        monkeypatch.setenv("TCGPU_PROF_MARKERS", "1")
    monkeypatch.setenv("TCGPU_BUCKET", "0")  # (in-order batches stay on the sort path: the launches counted below)
    if not hot:
        monkeypatch.setenv("TCGPU_HOT", "0")  # (... and the skewed batches on the LSD passes: round 6's hot form is counted below)
    eng = t.Engine(n_keys, n, fixed_params=fixed)
    eng.check_on_close = True
    eng.use_torch_stream()
    eng.register_params_uniform(*PLAN)
    orc = O.DenseOracle(n_keys)
    rng = np.random.default_rng(11)
    outs = [t.BatchResult() for _ in range(batches)]
    refs, alive = [], []
    eng.profile_enable(True)
    for b in range(batches):
        slots = rng.integers(0, n_keys, n).astype(np.uint32)
        slots[: n // 8] = rng.integers(0, 64, n // 8)  # hot keys: runs that cross rows and waves
        now = T0 + b * 300_000_000
        if general:
            nows = now + np.arange(n, dtype=np.int64)
            refs.append(orc.batch_slots(slots, *PLAN, 1, nows))
            now_arg = torch.from_numpy(nows).cuda()
        else:
            refs.append(orc.batch_slots(slots, *PLAN, 1, now))
            now_arg = now
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=now_arg, want=("allowed",), out=outs[b],
                                   inputs_ready=piped, outputs_idle=piped)
        alive.append((d, now_arg))  # (the columns stay untouched until the results are ready)
        if hot and b % 4 == 3:
            torch.cuda.synchronize()  # (the evaluations' notes reach the host: the next batches are grouped in the hot form)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile_enable(False)
    for b in range(batches):
        got = outs[b].allowed.cpu().numpy()[:n]
        assert (got == refs[b].allowed.astype(np.uint8)).all(), f"batch {b}: profiling changed a decision"
    passes = 3 if n_keys >= (1 << 16) else 2
    if hot:
        # a batch is grouped by the LSD passes (k_hist + 3) until the host has a hot list, then in the hot form: the partition
        # and the finish
        lsd = prof["prep"][1]
        assert 1 <= lsd < batches and prof["sort"][1] == passes * lsd + 2 * (batches - lsd) and prof["eval"][1] == batches, prof
        eng.close()
        return prof
    assert prof["prep"][1] == batches and prof["sort"][1] == passes * batches and prof["eval"][1] == batches, prof
    for stage in ("prep", "sort", "eval"):
        ms, calls = prof[stage]
        assert 0.0 < ms / calls < 20.0, (stage, prof[stage])  # a kernel of a 128 Ki batch: microseconds, not seconds
    eng.close()
    return prof


@pytest.mark.parametrize("piped", [True, False], ids=["pipelined", "in_order"])
@pytest.mark.parametrize("general", [False, True], ids=["uniform", "general"])
@pytest.mark.parametrize("fixed", [False, True], ids=["wide", "fixed"])
def test_profiling_counts_every_launch_and_changes_nothing(fixed, general, piped, monkeypatch):
    _run(fixed, general, piped, monkeypatch)


def test_profiling_counts_the_hot_forms_launches(monkeypatch):
    _run(True, False, True, monkeypatch, batches=24, hot=True)


def test_marker_event_fallback(monkeypatch):
    _run(True, False, True, monkeypatch, markers=True)


def test_dispatch_events_do_not_exceed_marker_events(monkeypatch):
    """the kernel's own execution time (events on the dispatch packet) cannot be longer than the interval between a marker
    recorded before and one recorded after it on the same stream (which adds the gaps around the kernel)"""
    a = _run(True, False, False, monkeypatch, n=1 << 19, n_keys=2_000_000, batches=8)
    b = _run(True, False, False, monkeypatch, markers=True, n=1 << 19, n_keys=2_000_000, batches=8)
    for stage in ("prep", "sort", "eval"):
        ka, kb = a[stage][0] / a[stage][1], b[stage][0] / b[stage][1]
        assert ka <= kb * 1.25, (stage, ka, kb)
