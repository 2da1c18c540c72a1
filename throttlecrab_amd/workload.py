"""Synthetic request streams for the BASELINE.json configs.

Deterministic (SplitMix64 counter-based), numpy only; the bench uploads the
result to HBM once, outside the timed region.  Shapes follow the reference's
own benchmarks (throttlecrab-server/benches/store_performance.rs:7-365,
examples/store_comparison.rs:17: burst 100, 1000 per 3600 s).
"""
from __future__ import annotations

import numpy as np

T0_NS = 1_700_000_000 * 10**9
REF_PARAMS = (100, 1000, 3600)  # store_comparison.rs:17

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised SplitMix64 finaliser over uint64."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, start: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        ctr = np.arange(start, start + n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
    return splitmix64(ctr)


def uniform_slots(n_keys: int, n: int, seed: int = 2, start: int = 0) -> np.ndarray:
    """config 2: n requests, slot uniform in [0, n_keys)."""
    r = _stream(seed, start, n)
    # 64x64->hi multiply-shift would need u128; n_keys << 2^32 so use the top 32 bits
    return (((r >> np.uint64(32)) * np.uint64(n_keys)) >> np.uint64(32)).astype(np.uint32)


class Zipf:
    """config 3: Zipf(s) ranks over n_keys via an inverse-CDF table, rank -> slot
    through a fixed pseudo-random permutation (so hot keys are scattered)."""

    def __init__(self, n_keys: int, s: float = 1.1, perm_seed: int = 33):
        self.n_keys = n_keys
        w = np.arange(1, n_keys + 1, dtype=np.float64) ** (-s)
        self.cdf = np.cumsum(w)
        self.cdf /= self.cdf[-1]
        # permutation by sorting hashed ids (deterministic)
        self.perm = np.argsort(splitmix64(np.arange(n_keys, dtype=np.uint64) ^ np.uint64(perm_seed)),
                               kind="stable").astype(np.uint32)

    def slots(self, n: int, seed: int = 3, start: int = 0) -> np.ndarray:
        r = _stream(seed, start, n)
        u = (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        rank = np.searchsorted(self.cdf, u, side="left")
        np.minimum(rank, self.n_keys - 1, out=rank)
        return self.perm[rank]


def shard_of(key_id: np.ndarray, world: int) -> np.ndarray:
    """config 4: owner GPU of a global key id = mix64(id) mod world."""
    return (splitmix64(key_id.astype(np.uint64) ^ np.uint64(0xA5A5A5A5)) % np.uint64(world)).astype(np.uint32)


def string_keys(ids: np.ndarray, prefix: bytes = b"key_"):
    """`format!("key_{}", i)` key arena for ids -> (bytes uint8[], offsets uint32[n+1]);
    vectorised (the benches format millions of keys)."""
    ids = np.asarray(ids, dtype=np.uint64)
    n, plen = len(ids), len(prefix)
    nd = np.ones(n, dtype=np.int64)
    p10 = np.uint64(10)
    lim = p10
    for _ in range(19):
        nd += ids >= lim
        lim = lim * p10 if lim < np.uint64(10**18) else np.uint64(2**64 - 1)
    lens = nd + plen
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum(lens)
    buf = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    base = off[:-1].astype(np.int64)
    for j, ch in enumerate(prefix):
        buf[base + j] = ch
    rem = ids.copy()
    # least significant digit goes to the last position
    for d in range(int(nd.max()) if n else 0):
        m = nd > d
        buf[base[m] + plen + nd[m] - 1 - d] = (rem[m] % p10).astype(np.uint8) + ord("0")
        rem //= p10
    return buf, off
