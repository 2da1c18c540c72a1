import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_keys import _keyset, _stream, PSETS, T0, _engine, _oracle, FIELDS
seed = 1
rng = np.random.default_rng(seed)
keys = _keyset(rng, 700)
eng, orc = _engine(2048), _oracle()
for rnd in range(4):
    n = 20000
    idx, kb, ko = _stream(rng, keys[: 200 + 150 * rnd], n)
    ps = PSETS[idx % len(PSETS)]
    q = rng.choice(np.array([0, 1, 1, 2, -1], dtype=np.int64), n)
    now = T0 + rnd * 3 * 10**9 + rng.integers(0, 2 * 10**9, n)
    ref = orc.batch_keys(kb, ko, ps[:, 0].copy(), ps[:, 1].copy(), ps[:, 2].copy(), q, now)
    res = eng.rate_limit_batch_keys(kb, ko, max_burst=ps[:, 0].copy(), count_per_period=ps[:, 1].copy(),
                                    period=ps[:, 2].copy(), quantity=q, now_ns=now)
    bad = np.nonzero(res.allowed != ref.allowed)[0]
    print("round", rnd, "bad", bad[:10])
    if bad.size:
        slots = np.array([eng.lookup_slot(keys[i]) for i in range(200 + 150 * rnd)])
        print("distinct slots", len(set(slots.tolist())), "of", len(slots), "min", slots.min())
        for b in bad[:3]:
            k = idx[b]
            print(" req", b, "keyidx", k, "key", keys[k][:40], "len", len(keys[k]), "slot", slots[k], "params", ps[b], "q", q[b], "now", now[b]-T0)
            same = np.nonzero(idx == k)[0]
            print("  same-key requests:", len(same), "positions", same[:10], "...", same[-3:])
            for f in FIELDS:
                print("   ", f, "got", getattr(res, f)[same][:12], "want", getattr(ref, f)[same][:12])
            dup = np.nonzero(slots == slots[k])[0]
            print("  keys sharing slot:", dup)
        break
