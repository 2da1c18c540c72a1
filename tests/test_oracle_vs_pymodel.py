"""The C oracle against an independent pure-Python restatement of the reference (oracle/pymodel.py):
random request streams with extreme parameters, store operations and cleanups, all six outputs."""
import numpy as np
import pytest

from tests import kat

T0 = kat.load()["t0_ns"]
BURST = [1, 2, 3, 5, 10, 100, 2**32, 2**32 + 1, 2**63 - 1, (2**63 - 1) // 1000, 0, -4]
COUNT = [1, 6, 7, 10, 100, 1000, 120, 2**62, 2**63 - 1, 0, -1]
PERIOD = [1, 60, 3600, 86400, 2**62, 2**63 - 1, 0, -9]
QTY = [0, 1, 1, 1, 2, 5, 1000, 2**62, 2**63 - 1, -1, -2**63]
VAL = [0, 1, -1, 2**63 - 1, -2**63, T0, T0 + 10**12]
TTL = [0, 1, 10**9, 60 * 10**9, 2**63, 2**64 - 1]


@pytest.mark.parametrize("seed", range(8))
def test_c_oracle_matches_python_model(seed):
    from oracle import oracle as O
    from oracle.pymodel import PyModel
    rng = np.random.default_rng(seed)
    keys = [b"k%d" % i for i in range(40)]
    c = O.AdaptiveOracle(capacity=100000, created_ns=T0, auto_cleanup=False)
    m = PyModel()
    now = T0
    pick = lambda xs: int(xs[rng.integers(0, len(xs))])
    for step in range(6000):
        r = rng.random()
        now = min(2**63 - 1, max(0, now + int(rng.integers(-10**9, 4 * 10**9)))) if rng.random() < 0.97 else pick([-5, 0, 2**62, 2**63 - 1, T0])
        key = keys[rng.integers(0, len(keys))]
        if r < 0.8:
            sticky = rng.random() < 0.7
            b, cn, p = (pick(BURST[:6]), pick(COUNT[:7]), pick(PERIOD[:4])) if sticky else (pick(BURST), pick(COUNT), pick(PERIOD))
            q = pick(QTY[:6]) if sticky else pick(QTY)
            got = c.rate_limit(key, b, cn, p, q, now)
            exp = m.rate_limit(key, b, cn, p, q, now)
            assert tuple(got) == tuple(exp), (seed, step, key, b, cn, p, q, now, got, exp)
        elif now >= 0:
            if r < 0.86:
                assert c.get(key, now) == m.get(key, now)
            elif r < 0.92:
                v, ttl = pick(VAL), pick(TTL)
                assert c.set_if_not_exists_with_ttl(key, v, ttl, now) == m.set_if_not_exists_with_ttl(key, v, ttl, now)
            elif r < 0.97:
                cur = m.get(key, now)
                old = cur if (cur is not None and rng.random() < 0.7) else pick(VAL)
                new, ttl = pick(VAL), pick(TTL)
                assert c.compare_and_swap_with_ttl(key, old, new, ttl, now) == m.compare_and_swap_with_ttl(key, old, new, ttl, now)
            else:
                before = len(c)
                c.force_cleanup(now)
                assert before - len(c) == m.cleanup(now)
    t_end = max(now, T0)
    for k in keys:
        assert c.get(k, t_end) == m.get(k, t_end)


KAT = kat.load()


@pytest.mark.parametrize("sc", KAT["scenarios"], ids=[s["name"] for s in KAT["scenarios"]])
def test_python_model_replays_the_reference_known_answers(sc):
    """The second restatement is pinned to the reference's own tests as well."""
    from oracle.pymodel import PyModel
    kat.replay_scenario(sc, PyModel())
