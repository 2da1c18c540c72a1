"""GPU parity of the bucket path (csrc/bucket_path.hpp): uniform batches that run in order on the engine's
stream are grouped by a stable partition into key-range buckets + a per-bucket rank pass instead of a sort,
with the sort path enqueued behind a device-side gate (the partition's largest bucket).  Every mode must give
the oracle's results bit for bit, on all outputs and on the resident state:
  default       the engine's own thresholds (batches >= 16384 requests, buckets <= 1024 requests)
  all_sizes     every eligible batch, however small
  long_buckets  ... and buckets of any length stay on the bucket path (the walk in pieces with parked stores)
  gate_trips    ... and every batch trips the gate (bucket kernels leave at once, the gated sort path runs)
  gate_trips_backoff  ... with the engine's back-off: once the host has seen a tripped gate, the next 32 batches are
                sorted without being partitioned first
  off           the bucket path disabled
(The range path, which takes such batches first since round 4, is switched off here: TCGPU_RANGE=0.)"""
import numpy as np
import pytest

from tests.test_gpu_slots import FIELDS, T0, _oracle, assert_same, assert_state_same

pytestmark = pytest.mark.gpu

MODES = {
    "default": {},
    "all_sizes": {"TCGPU_BUCKET_MIN_N": "1"},
    "long_buckets": {"TCGPU_BUCKET_MIN_N": "1", "TCGPU_BUCKET_SKEW": "32767"},
    "gate_trips": {"TCGPU_BUCKET_MIN_N": "1", "TCGPU_BUCKET_SKEW": "1", "TCGPU_BUCKET_BACKOFF": "0"},
    "gate_trips_backoff": {"TCGPU_BUCKET_MIN_N": "1", "TCGPU_BUCKET_SKEW": "1"},
    "off": {"TCGPU_BUCKET": "0"},
}
ENV = ("TCGPU_BUCKET", "TCGPU_BUCKET_MIN_N", "TCGPU_BUCKET_SKEW", "TCGPU_BUCKET_PIPED", "TCGPU_BUCKET_BACKOFF")


@pytest.fixture(params=list(MODES), ids=list(MODES))
def mode(request, monkeypatch):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[request.param].items():
        monkeypatch.setenv(k, v)
    # (round 4: in-order batches of a stream the range path takes no longer come here -- these tests are about the bucket
    # path and its gated sort, so the range path is off for them; tests/test_gpu_range.py and the rest of the suite run with it)
    monkeypatch.setenv("TCGPU_RANGE", "0")
    return request.param


def _engine(capacity, max_batch):
    import throttlecrab_amd as t
    e = t.Engine(capacity, max_batch)  # (reads the TCGPU_BUCKET* variables now)
    e.check_on_close = True
    return e


def _stream(rng, kind, n, cap):
    if kind == "uniform":
        s = rng.integers(0, cap, n)
    elif kind == "dense":      # every key several times: long buckets everywhere
        s = rng.integers(0, max(cap // 64, 1), n) * 61 % cap
    else:                      # one key = a third of the batch
        s = np.where(rng.random(n) < 0.33, cap // 3, rng.integers(0, cap, n))
    s = s.astype(np.uint32)
    bad = rng.random(n) < 0.005
    s[bad] = cap + rng.integers(0, 3, bad.sum()).astype(np.uint32)  # out-of-range slots -> Internal
    return s


@pytest.mark.parametrize("kind", ["uniform", "dense", "hot_key"])
@pytest.mark.parametrize("params", ["registered", "scalar", "per_slot"])
def test_uniform_batches_match_the_sequence(mode, kind, params):
    cap, n = 300_000 + 77, 70_000
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}/{params}".encode()))
    eng, orc = _engine(cap, n), _oracle(cap)
    plans = np.array([(100, 1000, 3600), (5, 10, 60), (3, 7, 60), (20, 600, 60)], dtype=np.int64)
    if params == "registered":
        eng.register_params_uniform(5, 10, 60)
    elif params == "per_slot":
        idx = rng.integers(0, len(plans), cap)
        eng.register_params(plans[idx[: cap - 500], 0], plans[idx[: cap - 500], 1], plans[idx[: cap - 500], 2])  # the last 500: none
    for rnd in range(4):
        slots = _stream(rng, kind, n if rnd != 2 else 20_011, cap)
        now = T0 + rnd * 900_000_000 if rnd != 3 else T0 - 10**9  # also goes back once
        q = 1 if rnd != 1 else 2
        if params == "per_slot":
            ok = (slots < cap - 500) | (slots >= cap)  # (out of range: any valid triple takes the oracle as far as its store)
            pl = plans[idx[np.minimum(slots, cap - 1)]]
            b, c, p = np.where(ok, pl[:, 0], 0), np.where(ok, pl[:, 1], 0), np.where(ok, pl[:, 2], 0)
            ref = orc.batch_slots(slots, b, c, p, q, now)
            res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
        elif params == "registered":
            ref = orc.batch_slots(slots, 5, 10, 60, q, now)
            res = eng.rate_limit_batch_slots(slots, registered=True, quantity=q, now_ns=now)
        else:
            ref = orc.batch_slots(slots, 100, 1000, 3600, q, now)
            res = eng.rate_limit_batch_slots(slots, max_burst=100, count_per_period=1000, period=3600, quantity=q, now_ns=now)
        assert_same(res, ref, f"{mode}/{kind}/{params} round {rnd}")
        assert_state_same(eng, orc, slots[::9])
    eng.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8193])
def test_edge_sizes(mode, n):
    """wave / step / tile boundaries, a key space that is not a multiple of the bucket width, decisions only"""
    import torch
    cap = 12_345
    rng = np.random.default_rng(n)
    eng, orc = _engine(cap, 10_000), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(3, 7, 60)
    for rnd in range(3):
        slots = rng.integers(0, cap, n).astype(np.uint32)
        slots[0] = cap - 1
        if n > 2:
            slots[-1], slots[n // 2] = cap, slots[0]  # the clamp value itself; a duplicate of the first
        ref = orc.batch_slots(slots, 3, 7, 60, 1, T0 + rnd)
        res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=T0 + rnd,
                                         want=("allowed", "allowed_bits", "status"))
        torch.cuda.synchronize()
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed.astype(np.uint8)), (mode, n, rnd)
        assert np.array_equal(res.status.cpu().numpy(), ref.status.astype(np.uint8)), (mode, n, rnd)
        bits = np.unpackbits(res.allowed_bits.cpu().numpy().view(np.uint8), bitorder="little")[:n]
        assert np.array_equal(bits, ref.allowed.astype(np.uint8)), (mode, n, rnd)
        assert_state_same(eng, orc, slots)
    eng.close()


REGIMES = [
    (1, 1, 1, 1, "burst 1"),
    (1, 1, 1, 0, "q=0 burst 1"),
    (10, 100, 60, 0, "q=0"),
    (10, 2**62, 60, 1, "ei=0"),
    (10, 10, 60, 2**62, "huge q"),
    (2**63 - 1, 2**63 - 1, 2**63 - 1, 1, "i64::MAX"),
    (4, 10, 60, 1, "regular"),
]


@pytest.mark.parametrize("burst,count,period,q,label", REGIMES, ids=[c[4] for c in REGIMES])
@pytest.mark.parametrize("grouped", [False, True], ids=["by_index", "grouped_rows"])
def test_irregular_runs_and_grouped_rows(mode, burst, count, period, q, label, grouped):
    """runs the host cannot prove regular and TC_B_GROUPED_OUTPUT rows (both stay on the sort path whatever the mode),
    keys with ~10 requests each, all outputs"""
    import torch
    import zlib
    cap, n = 40_000, 24_000
    rng = np.random.default_rng(zlib.crc32(label.encode()))
    eng, orc = _engine(cap, n), _oracle(cap)
    eng.use_torch_stream()
    for rnd in range(3):
        slots = (rng.integers(0, 2400, n) * 17 % cap).astype(np.uint32)
        slots[rng.random(n) < 0.01] = cap + 1
        now = T0 + rnd * 300_000_000
        ref = orc.batch_slots(slots, burst, count, period, q, now)
        res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), max_burst=burst, count_per_period=count, period=period,
                                         quantity=q, now_ns=now, want=FIELDS, grouped=grouped, inputs_ready=(rnd == 1))
        torch.cuda.synchronize()
        if not grouped:
            assert_same(res, ref, f"{mode}/{label} round {rnd}")
        else:
            order = res.order.cpu().numpy().astype(np.int64)
            assert np.array_equal(np.sort(order), np.arange(n))
            sk = np.minimum(slots[order].astype(np.int64), cap)
            assert np.all(np.diff(sk) >= 0) and np.all((np.diff(sk) > 0) | (np.diff(order) > 0)), "rows grouped by key, index order inside"
            for f in FIELDS:
                back = np.empty(n, np.int64)
                back[order] = getattr(res, f).cpu().numpy().astype(np.int64)
                assert np.array_equal(back, getattr(ref, f).astype(np.int64)), (mode, label, f, rnd)
        assert_state_same(eng, orc, slots[::5])
    eng.close()


def test_in_order_batches_between_pipelined_ones(mode):
    """device batches: in-order ones and TC_B_INPUTS_READY ones (grouped on the auxiliary streams) mixed
    without a host synchronisation in between; counters and state at the end"""
    import torch
    cap, n = 200_000, 40_000
    rng = np.random.default_rng(12)
    eng, orc = _engine(cap, n), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(10, 100, 60)
    outs, refs, keep = [], [], []
    for rnd in range(10):
        slots = _stream(rng, "uniform" if rnd % 3 else "dense", n, cap)
        now = T0 + rnd * 400_000_000
        refs.append(orc.batch_slots(slots, 10, 100, 60, 1, now))
        d = torch.from_numpy(slots.astype(np.int32)).cuda()
        keep.append(d)
        outs.append(eng.rate_limit_batch_slots(d, registered=True, quantity=1, now_ns=now, want=FIELDS, inputs_ready=(rnd % 2 == 1)))
    torch.cuda.synchronize()
    for rnd, (res, ref) in enumerate(zip(outs, refs)):
        assert_same(res, ref, f"{mode} batch {rnd}")
    c = eng.counters()
    assert c["total"] == 10 * n
    assert c["allowed"] == sum(int(r.allowed.sum()) for r in refs)
    assert_state_same(eng, orc, np.arange(0, cap, 37))
    eng.close()


def test_denied_counters(mode, monkeypatch):
    """TC_CFG_TRACK_DENIED through the bucket path: per-key denials == the oracle's"""
    import throttlecrab_amd as t
    cap, n = 50_000, 30_000
    rng = np.random.default_rng(3)
    eng = t.Engine(cap, n, track_denied=True)
    orc = _oracle(cap)
    denied = np.zeros(cap, np.int64)
    for rnd in range(3):
        slots = (rng.integers(0, 4000, n) * 11 % cap).astype(np.uint32)
        ref = orc.batch_slots(slots, 2, 10, 60, 1, T0 + rnd * 10**6)
        eng.rate_limit_batch_slots(slots, max_burst=2, count_per_period=10, period=60, quantity=1, now_ns=T0 + rnd * 10**6)
        np.add.at(denied, slots[(ref.allowed == 0) & (ref.status == 0)], 1)
    top = eng.top_denied(50)
    want = sorted(((int(c), -s) for s, c in enumerate(denied) if c), reverse=True)[:50]
    assert [(s, c) for s, c in top] == [(-ns, c) for c, ns in want]
    eng.close()


@pytest.mark.parametrize("kind", ["uniform", "hot_key"])
def test_full_size_in_order(kind):
    """BASELINE configs[1] shape in order on one stream: 10 M keys, 1 Mi requests per batch (2048-slot buckets)"""
    import torch
    cap, n = 10_000_000, 1 << 20
    rng = np.random.default_rng(99)
    eng, orc = _engine(cap, n), _oracle(cap)
    eng.use_torch_stream()
    eng.register_params_uniform(100, 1000, 3600)
    for rnd in range(3):
        slots = _stream(rng, kind, n, cap)
        now = T0 + rnd * 1_000_000
        ref = orc.batch_slots(slots, 100, 1000, 3600, 1, now)
        res = eng.rate_limit_batch_slots(torch.from_numpy(slots.astype(np.int32)).cuda(), registered=True, quantity=1, now_ns=now,
                                         want=("allowed", "status"))
        torch.cuda.synchronize()
        assert np.array_equal(res.allowed.cpu().numpy(), ref.allowed.astype(np.uint8)), (kind, rnd)
        assert np.array_equal(res.status.cpu().numpy(), ref.status.astype(np.uint8)), (kind, rnd)
    tat, exp = eng.read_state(0, cap)
    pick = rng.integers(0, cap, 200_000)
    for s in pick[:2000]:
        ot, oe, occ = orc.peek(int(s))
        assert (int(tat[s]), int(exp[s])) == ((ot, oe) if occ else (0, 0)), s
    assert eng.counters()["total"] == 3 * n
    eng.close()
