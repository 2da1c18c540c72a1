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


def pack_keys(keys):
    """A list of bytes keys -> key arena (bytes uint8[total], offsets uint32[n+1])."""
    off = np.zeros(len(keys) + 1, dtype=np.uint32)
    if len(keys):
        off[1:] = np.cumsum([len(k) for k in keys], dtype=np.uint64).astype(np.uint32)
    buf = np.frombuffer(b"".join(keys), dtype=np.uint8).copy() if len(keys) else np.zeros(0, np.uint8)
    return (buf if buf.size else np.zeros(1, np.uint8)), off


def long_keys(ids: np.ndarray, device=None):
    """SURVEY.md section 8(d) cfg 5, second key set: one 32..64-byte key of printable ASCII per id (a
    function of the id alone), as a key arena.  Half of them are longer than the 48 bytes a slot's
    key record holds inline.  device: a torch device -- the arena is then built there (same bytes; compacting
    64 candidate bytes per key into the arena takes seconds per million keys in numpy) and returned as
    CUDA tensors (uint8 bytes, int32 offsets)."""
    if device is not None:
        return _long_keys_torch(ids, device)
    ids = np.asarray(ids, dtype=np.uint64)
    n = len(ids)
    with np.errstate(over="ignore"):
        lens = (32 + (splitmix64(ids ^ np.uint64(0x51ED2701)) % np.uint64(33))).astype(np.int64)
        words = np.empty((n, 8), dtype=np.uint64)
        for j in range(8):
            words[:, j] = splitmix64(ids * np.uint64(8) + np.uint64(j) + np.uint64(0x1234567))
    chars = (words.view(np.uint8).reshape(n, 64) % np.uint8(94) + np.uint8(33))  # '!' .. '~'
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum(lens)
    keep = np.arange(64)[None, :] < lens[:, None]
    buf = chars[keep]  # row-major: every key's first `len` bytes, in id order
    return (buf if buf.size else np.zeros(1, np.uint8)), off


def _long_keys_torch(ids, device):
    import torch

    def i64(v):  # a uint64 constant as the int64 with the same bits (torch has no unsigned 64-bit arithmetic)
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix(x):  # splitmix64 on int64 tensors: wrap-around multiplication has the same low 64 bits
        z = x + i64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)
    x = torch.as_tensor(np.asarray(ids, dtype=np.int64), device=device)
    n = x.numel()
    h = mix(x ^ 0x51ED2701)
    # h mod 33 for the UNSIGNED value of h: (h_hi * 2^32 + h_lo) mod 33 with 2^32 mod 33 == 4
    lens = 32 + (((lsr(h, 32) % 33) * 4 + ((h & 0xFFFFFFFF) % 33)) % 33)
    words = torch.stack([mix(x * 8 + j + 0x1234567) for j in range(8)], dim=1).contiguous()
    chars = (words.view(torch.uint8).reshape(n, 64) % 94 + 33).to(torch.uint8)
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    off[1:] = torch.cumsum(lens, 0)
    keep = torch.arange(64, device=device)[None, :] < lens[:, None]
    buf = chars[keep]
    off = off.to(torch.int32)
    if buf.is_cuda:
        # the arena is COMPLETE in device memory on return: what TC_B_INPUTS_READY requires of its caller (the
        # engine reads it on a stream of its own, which is not ordered behind torch's)
        torch.cuda.current_stream(buf.device).synchronize()
    return (buf if buf.numel() else torch.zeros(1, dtype=torch.uint8, device=device)), off


class Config4Stream:
    """BASELINE configs[4] / SURVEY.md section 8(d) cfg 5: 10 M string keys, then mixed batches of
    70 % hits of live keys, 20 % new keys, 10 % re-hits of keys whose entries have expired, rate (10, 100 / 60 s)
    so that an entry outlives its last allowed request by 5.4 .. 11 s, `now` one second per batch, an expiry
    sweep every `sweep_every` batches.

    prefill(k), k < n_prefill: batch of `batch` new keys.  Batches 0 .. n_prefill-2 are stamped within one
      second at T0 (they are all expired when the mixed phase starts 12 s later and feed its expired re-hits),
      the last one is stamped one second before the mixed phase (live: it feeds the first hits).
    mixed(s): ids of step s, now = T1 + s seconds.
    Deterministic; ids are dense from 0 in order of first appearance."""
    PARAMS = (10, 100, 60)

    def __init__(self, batch: int = 1 << 20, n_prefill: int = 10, seed: int = 5, long: bool = False, sweep_every: int = 4):
        self.batch, self.n_prefill, self.long, self.sweep_every = batch, n_prefill, long, sweep_every
        self.rng = np.random.default_rng(seed)
        self.t1 = T0_NS + 13 * 10**9
        self.next_id = batch * n_prefill
        old = self.rng.permutation(batch * (n_prefill - 1))  # the expired pool, drawn without replacement
        self._expired, self._expired_at = old, 0
        self._recent = [np.arange(batch * (n_prefill - 1), batch * n_prefill, dtype=np.int64)]  # live pools, newest last

    def keys(self, ids, device=None):
        """key arena of `ids`: numpy (bytes uint8, offsets uint32), or CUDA tensors (uint8, int32) with `device`"""
        if self.long:
            return long_keys(ids, device)
        kb, ko = string_keys(ids)
        if device is None:
            return kb, ko
        import torch
        return torch.from_numpy(kb).to(device), torch.from_numpy(ko.astype(np.int32)).to(device)

    def prefill(self, k: int):
        ids = np.arange(k * self.batch, (k + 1) * self.batch, dtype=np.int64)
        now = T0_NS + k * 10**8 if k < self.n_prefill - 1 else self.t1 - 10**9
        return ids, now

    def mixed(self, s: int):
        B = self.batch
        n_new, n_exp = B // 5, B // 10
        new = np.arange(self.next_id, self.next_id + n_new, dtype=np.int64)
        self.next_id += n_new
        exp = self._expired[self._expired_at:self._expired_at + n_exp].astype(np.int64)
        self._expired_at += n_exp
        assert len(exp) == n_exp, "expired pool exhausted: fewer mixed steps or a larger prefill"
        pool = np.concatenate(self._recent[-3:])
        n_hit = B - n_new - n_exp
        # hits: mostly spread over the live keys, a fifth of them on 1000 hot ones (these run out of burst: denials)
        pick = self.rng.integers(0, len(pool), n_hit)
        hot = self.rng.random(n_hit) < 0.2
        pick[hot] = self.rng.integers(0, 1000, int(hot.sum()))
        hit = pool[pick]
        ids = np.concatenate([hit, new, exp])
        self.rng.shuffle(ids)
        self._recent.append(np.concatenate([new, exp]))
        return ids, self.t1 + s * 10**9

    def sweep_due(self, s: int) -> bool:
        return s % self.sweep_every == self.sweep_every - 1
