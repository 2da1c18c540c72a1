"""String keys, round 6: an explicit sweep (tc_sweep_expired without a wait) right behind pipelined key batches can run on the KEY
stream (TCGPU_SWEEP_ASIDE=1; exact, but measured no faster and therefore off by default), BESIDE the newest batches' evaluations (keys.hip: sweep_keys_device; maintenance_kernels.hpp: k_touch_mark, k_sweep_keys'
`touched` column, k_sweep_fixup) -- the reference's cleanup (adaptive_cleanup.rs:173-203) at that point of the sequence all the same.

Unsynchronised pipelined batches with sweeps between them, against the AdaptiveStore port with force_cleanup at the same points:
every field of every request, the number removed, the store's size, the table from both sides.  The streams are built so that the
newest batch's own slots matter to the sweep: sweeps at a LATER time than the batch in front of them (its freshly written cells
have expired by then and must go), requests that are denied or errors on new and on expired keys (a bound slot whose cell was
never written), quantities above the burst, duplicates of one key in a batch, sweeps that unbind nearly everything (table
rebuild inside the sweep), two sweeps in a row, a sweep after a single batch (goes behind, as before), other calls between the batch
and the sweep (goes behind)."""
import numpy as np
import pytest

from tests.test_gpu_keys import T0, _engine, assert_same

pytestmark = pytest.mark.gpu

MS = 10**6


def _stream(rng, n, hi, new_from, n_new):
    idx = np.where(rng.random(n) < 0.25, np.minimum(rng.zipf(1.4, n) - 1, hi - 1), rng.integers(0, hi, n))
    idx[:n_new] = new_from + np.arange(n_new)          # keys never seen before
    return rng.permutation(idx)


@pytest.mark.parametrize("track_denied", [False, True], ids=["plain", "track_denied"])
@pytest.mark.parametrize("aside", [True, False], ids=["aside", "behind"])
def test_sweeps_between_unsynchronised_pipelined_key_batches(aside, track_denied, monkeypatch):
    import torch
    from oracle import oracle as O
    monkeypatch.setenv("TCGPU_SWEEP_ASIDE", "1" if aside else "0")
    rng = np.random.default_rng(20260930 + track_denied)
    n, nb, cap = 30000, 16, 150000
    # (the arena: a sweep beside an evaluation compacts while the newest batch's long keys are still bound -- those that k_sweep_fixup
    # then unbinds stay in the arena until the next compaction; the default arena of this capacity, 1 MiB, is too tight for that here)
    eng = _engine(cap, n, track_denied=track_denied, key_arena_bytes=4 << 20)
    eng.use_torch_stream()
    orc = O.AdaptiveOracle(capacity=cap, created_ns=T0, auto_cleanup=False)
    # (2, 10 per second): emission interval 100 ms, tolerance 100 ms -- a written cell lives 100-200 ms
    burst, count, period = 2, 10, 1
    staged, nows, sweeps = [], [], {}
    hi, fresh = 20000, 50000
    for b in range(nb):
        idx = _stream(rng, n, hi, fresh, 3000)
        fresh += 3000
        hi += 1500
        keys = [b"sk:%d" % i if i % 7 else b"a-longer-key-for-the-overflow-arena-" + b"z" * 90 + b"%d" % i for i in idx]
        kb, ko = O.pack_keys(keys)
        q = np.ones(n, dtype=np.int64)
        q[rng.random(n) < 0.10] = 3    # above the burst: denied, the cell stays as it was
        q[rng.random(n) < 0.03] = -1   # an error: nothing written
        q[rng.random(n) < 0.05] = 0    # a probe
        staged.append((kb, ko, q, torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda(), torch.from_numpy(q).cuda()))
        nows.append(T0 + b * 120 * MS)
    # where the sweeps fall and at what time: at the batch's own time, a little later (part of its cells expired), much later (all)
    sweeps = {1: 0, 2: 60 * MS, 3: 150 * MS, 5: 400 * MS, 6: 0, 9: 90 * MS, 10: 0, 12: 10**10, 14: 130 * MS, 15: 0}
    torch.cuda.synchronize()
    outs, refs, removed_ref = [], [], 0
    for b, (kb, ko, q, dkb, dko, dq) in enumerate(staged):
        refs.append(orc.batch_keys(kb, ko, burst, count, period, q, nows[b]))
        outs.append(eng.rate_limit_batch_keys(dkb, dko, max_burst=burst, count_per_period=count, period=period, quantity=dq, now_ns=nows[b],
                                              inputs_ready=True))
        if b in sweeps:
            t = nows[b] + sweeps[b]
            if b == 9:       # another call between the batch and its sweep: the sweep goes behind everything
                assert eng.get(b"sk:1", nows[b]) == orc.get(b"sk:1", nows[b])
            before = len(orc)
            orc.force_cleanup(t)
            removed_ref += before - len(orc)
            eng.sweep_expired_async(t)
            if b == 6:       # two in a row (the second finds nothing to do at the same time)
                orc.force_cleanup(t)
                eng.sweep_expired_async(t)
    torch.cuda.synchronize()
    eng.synchronize()
    for b in range(nb):
        assert_same(outs[b], refs[b], f"batch {b}")
    c = eng.counters()
    assert c["swept"] == removed_ref, (c["swept"], removed_ref)
    assert c["live_slots"] == len(orc), (c["live_slots"], len(orc))   # (the last batch has a sweep behind it)
    assert eng.debug_check_keys() == 0
    info = eng.info()
    if aside:
        assert info["key_stream"] == 1 and info["pipelining_degraded"] == 0, info
        # (behind: the second of the pair at 6, the one at 9 with a call in front of it, the one at 10 after a lone batch)
        assert info["sweeps_aside"] == 8, info
    else:
        assert info["sweeps_aside"] == 0, info
    # the synchronous form, and the table keeps working: a last batch, a counted sweep
    t_end = nows[-1] + 10**9
    kb, ko, q, dkb, dko, dq = staged[3]
    ref = orc.batch_keys(kb, ko, burst, count, period, q, t_end)
    res = eng.rate_limit_batch_keys(dkb, dko, max_burst=burst, count_per_period=count, period=period, quantity=dq, now_ns=t_end, inputs_ready=True)
    kb2, ko2, q2, dkb2, dko2, dq2 = staged[4]
    ref2 = orc.batch_keys(kb2, ko2, burst, count, period, q2, t_end + 50 * MS)
    res2 = eng.rate_limit_batch_keys(dkb2, dko2, max_burst=burst, count_per_period=count, period=period, quantity=dq2, now_ns=t_end + 50 * MS, inputs_ready=True)
    before = len(orc)
    orc.force_cleanup(t_end + 160 * MS)
    assert eng.sweep_expired(t_end + 160 * MS) == before - len(orc)
    assert_same(res, ref, "after")
    assert_same(res2, ref2, "after 2")
    assert eng.counters()["live_slots"] == len(orc)
    assert eng.debug_check_keys() == 0
    sample = [b"sk:%d" % i for i in range(1, 60)] + [b"sk:%d" % (fresh - 1 - i) for i in range(60)]
    for k in sample:
        assert eng.get(k, t_end + 170 * MS) == orc.get(k, t_end + 170 * MS), k
    eng.close()


def test_sweep_beside_an_evaluation_with_the_table_nearly_full(monkeypatch):
    """Capacity for little more than what is live: every sweep must hand the slots back in time for the next batches' new keys (a slot
    the sweep frees while the newest batch is still being evaluated is on the free stack before the next key stage pops), and a key
    that loses its slot in k_sweep_fixup is found again as a new key by the next batch."""
    import torch
    from oracle import oracle as O
    monkeypatch.setenv("TCGPU_SWEEP_ASIDE", "1")   # (off by default: no faster, profiles/r06_v45_sweep_aside_ab.txt)
    rng = np.random.default_rng(77)
    n, nb = 20000, 24
    cap = 3 * n + 2000
    eng = _engine(cap, n)
    eng.use_torch_stream()
    orc = O.AdaptiveOracle(capacity=cap, created_ns=T0, auto_cleanup=False)
    burst, count, period = 3, 10, 1   # a written cell lives 200-400 ms
    staged = []
    for b in range(nb):
        idx = np.concatenate([b * n // 2 + rng.permutation(n // 2), rng.integers(max(0, (b - 2) * n // 2), b * n // 2 + 1, n - n // 2)])  # half new, half of the last two batches' keys
        kb, ko = O.pack_keys([b"nf:%d" % i for i in rng.permutation(idx)])
        staged.append((kb, ko, torch.from_numpy(kb).cuda(), torch.from_numpy(ko.astype(np.int32)).cuda()))
    torch.cuda.synchronize()
    outs, refs, removed_ref = [], [], 0
    for b, (kb, ko, dkb, dko) in enumerate(staged):
        now = T0 + b * 250 * MS
        refs.append(orc.batch_keys(kb, ko, burst, count, period, 1, now))
        outs.append(eng.rate_limit_batch_keys(dkb, dko, max_burst=burst, count_per_period=count, period=period, quantity=1, now_ns=now, inputs_ready=True))
        if b % 2 == 1:
            before = len(orc)
            orc.force_cleanup(now + (b % 3) * 100 * MS)
            removed_ref += before - len(orc)
            eng.sweep_expired_async(now + (b % 3) * 100 * MS)
    torch.cuda.synchronize()
    eng.synchronize()
    for b in range(nb):
        assert (refs[b].status == 0).all()
        assert_same(outs[b], refs[b], f"batch {b}")
    c = eng.counters()
    assert c["swept"] == removed_ref and c["live_slots"] == len(orc) and c["errors"] == 0, c
    assert eng.info()["sweeps_aside"] == nb // 2
    assert eng.debug_check_keys() == 0
    eng.close()


def test_engine_info_serves_callers_built_against_the_struct_without_sweeps_aside():
    """tc_engine_info grew by `sweeps_aside` (appended): a caller that passes the old, smaller struct_size gets the fields it knows and
    nothing written behind them; a struct_size below the old one is still refused."""
    import ctypes as C
    from throttlecrab_amd import _lib as L
    eng = _engine(1000, 256)
    lib = L.load()
    old_size = L.tc_engine_info.sweeps_aside.offset
    buf = (C.c_uint8 * (C.sizeof(L.tc_engine_info) + 16))()
    C.memset(buf, 0xAB, len(buf))
    info = L.tc_engine_info.from_buffer(buf)
    info.struct_size = old_size
    assert lib.tc_engine_info_get(eng._h, C.byref(info)) == 0
    assert info.struct_size == old_size and info.scratch_sets >= 1
    assert bytes(buf[old_size:]) == b"\xab" * (len(buf) - old_size)   # nothing written past what the caller said it has
    info.struct_size = old_size - 8
    assert lib.tc_engine_info_get(eng._h, C.byref(info)) != 0
    info.struct_size = C.sizeof(L.tc_engine_info)
    assert lib.tc_engine_info_get(eng._h, C.byref(info)) == 0 and info.sweeps_aside == 0
    eng.close()
