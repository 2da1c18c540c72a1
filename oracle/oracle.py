"""ctypes binding of the CPU oracle (oracle/gcra_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgcra_oracle.so")

OK, NEGATIVE_QUANTITY, INVALID_RATE_LIMIT, INTERNAL = 0, 1, 2, 3


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gcra_oracle.c")
    hdr = os.path.join(_HERE, "gcra_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgcra_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class _Result(C.Structure):
    _fields_ = [("limit", C.c_int64), ("remaining", C.c_int64),
                ("reset_after_ns", C.c_uint64), ("retry_after_ns", C.c_uint64),
                ("allowed", C.c_uint8), ("status", C.c_uint8)]


class _Store(C.Structure):
    _fields_ = [("vt", C.c_void_p), ("self", C.c_void_p)]


class _BatchIO(C.Structure):
    _fields_ = [("n", C.c_size_t),
                ("max_burst", C.c_void_p), ("burst_stride", C.c_size_t),
                ("count_per_period", C.c_void_p), ("count_stride", C.c_size_t),
                ("period", C.c_void_p), ("period_stride", C.c_size_t),
                ("quantity", C.c_void_p), ("quantity_stride", C.c_size_t),
                ("now_ns", C.c_void_p), ("now_stride", C.c_size_t),
                ("allowed", C.c_void_p), ("limit", C.c_void_p), ("remaining", C.c_void_p),
                ("reset_after_ns", C.c_void_p), ("retry_after_ns", C.c_void_p),
                ("status", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.tco_emission_interval.restype = C.c_uint64
    L.tco_emission_interval.argtypes = [C.c_int64, C.c_int64]
    L.tco_derive.restype = C.c_int
    L.tco_derive.argtypes = [C.c_int64] * 4 + [C.POINTER(C.c_int64)] * 2
    L.tco_rate_limit.restype = C.c_int
    L.tco_rate_limit.argtypes = [C.POINTER(_Store), C.c_char_p, C.c_size_t] + [C.c_int64] * 5 + [C.POINTER(_Result)]
    L.tco_adaptive_new.restype = C.c_void_p
    L.tco_adaptive_new.argtypes = [C.c_size_t, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int64]
    L.tco_adaptive_with_capacity.restype = C.c_void_p
    L.tco_adaptive_with_capacity.argtypes = [C.c_size_t, C.c_int64]
    L.tco_adaptive_free.argtypes = [C.c_void_p]
    L.tco_adaptive_as_store.restype = _Store
    L.tco_adaptive_as_store.argtypes = [C.c_void_p]
    L.tco_adaptive_len.restype = C.c_size_t
    L.tco_adaptive_len.argtypes = [C.c_void_p]
    L.tco_adaptive_cleanups.restype = C.c_uint64
    L.tco_adaptive_cleanups.argtypes = [C.c_void_p]
    L.tco_adaptive_force_cleanup.argtypes = [C.c_void_p, C.c_int64]
    L.tco_adaptive_set_auto_cleanup.argtypes = [C.c_void_p, C.c_int]
    L.tco_dense_new.restype = C.c_void_p
    L.tco_dense_new.argtypes = [C.c_size_t]
    L.tco_dense_free.argtypes = [C.c_void_p]
    L.tco_dense_as_store.restype = _Store
    L.tco_dense_as_store.argtypes = [C.c_void_p]
    L.tco_dense_peek.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.tco_dense_dump.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.tco_dense_sweep.restype = C.c_uint64
    L.tco_dense_sweep.argtypes = [C.c_void_p, C.c_int64]
    L.tco_dense_live.restype = C.c_size_t
    L.tco_dense_live.argtypes = [C.c_void_p]
    L.tco_batch_keys.argtypes = [C.POINTER(_Store), C.c_void_p, C.c_void_p, C.POINTER(_BatchIO)]
    L.tco_batch_slots.argtypes = [C.POINTER(_Store), C.c_void_p, C.POINTER(_BatchIO)]
    L.tco_batch_slots_mt.argtypes = [C.POINTER(_Store), C.c_void_p, C.POINTER(_BatchIO), C.c_int]
    L.tco_batch_keys_mt.restype = C.c_double
    L.tco_batch_keys_mt.argtypes = [C.c_int, C.c_size_t, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(_BatchIO)]
    L.tco_reference_shape.restype = C.c_double
    L.tco_reference_shape.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tco_format_keys.restype = C.c_size_t
    L.tco_format_keys.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    L.tco_hash_bytes.restype = C.c_uint64
    L.tco_hash_bytes.argtypes = [C.c_char_p, C.c_size_t]
    _lib = L
    return L


def emission_interval(count: int, period: int) -> int:
    return int(lib().tco_emission_interval(count, period))


def derive(burst: int, count: int, period: int, now: int):
    ei, dvt = C.c_int64(), C.c_int64()
    st = lib().tco_derive(burst, count, period, now, C.byref(ei), C.byref(dvt))
    return st, ei.value, dvt.value


def pack_keys(keys: Sequence[bytes]):
    """-> (bytes uint8[total], offsets uint32[n+1]) key arena."""
    off = np.zeros(len(keys) + 1, dtype=np.uint32)
    if len(keys):
        off[1:] = np.cumsum([len(k) for k in keys], dtype=np.uint64).astype(np.uint32)
    buf = np.frombuffer(b"".join(keys), dtype=np.uint8).copy() if len(keys) else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, off


class BatchOut:
    def __init__(self, n: int):
        self.allowed = np.zeros(n, np.uint8)
        self.limit = np.zeros(n, np.int64)
        self.remaining = np.zeros(n, np.int64)
        self.reset_after_ns = np.zeros(n, np.int64)
        self.retry_after_ns = np.zeros(n, np.int64)
        self.status = np.zeros(n, np.uint8)

    def fields(self):
        return {"allowed": self.allowed, "limit": self.limit, "remaining": self.remaining,
                "reset_after_ns": self.reset_after_ns, "retry_after_ns": self.retry_after_ns,
                "status": self.status}


def _param(a, n):
    """scalar or array -> (contiguous int64 array, stride)."""
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))
    if arr.size == 1 and n != 1:
        return arr, 0
    if arr.size != n:
        raise ValueError(f"parameter length {arr.size} != batch {n}")
    return arr, 1


def _make_io(n, burst, count, period, quantity, now):
    out = BatchOut(n)
    keep = []
    io = _BatchIO()
    io.n = n
    for name, sname, val in (("max_burst", "burst_stride", burst), ("count_per_period", "count_stride", count),
                             ("period", "period_stride", period), ("quantity", "quantity_stride", quantity),
                             ("now_ns", "now_stride", now)):
        arr, stride = _param(val, n)
        keep.append(arr)
        setattr(io, name, arr.ctypes.data)
        setattr(io, sname, stride)
    io.allowed = out.allowed.ctypes.data
    io.limit = out.limit.ctypes.data
    io.remaining = out.remaining.ctypes.data
    io.reset_after_ns = out.reset_after_ns.ctypes.data
    io.retry_after_ns = out.retry_after_ns.ctypes.data
    io.status = out.status.ctypes.data
    return io, out, keep


class _StoreBase:
    """RateLimiter<S> over one oracle store (rate_limiter.rs:42-58)."""

    def __init__(self):
        self._st: Optional[_Store] = None

    # RateLimiter::rate_limit -- returns (status, allowed, limit, remaining, reset_ns, retry_ns)
    def rate_limit(self, key: bytes, max_burst: int, count_per_period: int, period: int, quantity: int, now_ns: int):
        r = _Result()
        lib().tco_rate_limit(C.byref(self._st), key, len(key), max_burst, count_per_period, period,
                             quantity, now_ns, C.byref(r))
        return (r.status, bool(r.allowed), r.limit, r.remaining, r.reset_after_ns, r.retry_after_ns)

    # Store trait (store/mod.rs:85-133) -- direct calls through the vtable
    def _vt(self):
        vt_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int))
        cas_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.POINTER(C.c_int))
        set_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_uint64, C.c_int64, C.POINTER(C.c_int))
        ptrs = C.cast(self._st.vt, C.POINTER(C.c_void_p))
        return vt_t(ptrs[0]), cas_t(ptrs[1]), set_t(ptrs[2])

    def get(self, key: bytes, now_ns: int):
        g, _, _ = self._vt()
        v, f = C.c_int64(), C.c_int()
        rc = g(self._st.self, key, len(key), now_ns, C.byref(v), C.byref(f))
        if rc != 0:
            raise RuntimeError("store error")
        return v.value if f.value else None

    def compare_and_swap_with_ttl(self, key: bytes, old: int, new: int, ttl_ns: int, now_ns: int) -> bool:
        _, c, _ = self._vt()
        ok = C.c_int()
        if c(self._st.self, key, len(key), old, new, ttl_ns, now_ns, C.byref(ok)) != 0:
            raise RuntimeError("store error")
        return bool(ok.value)

    def set_if_not_exists_with_ttl(self, key: bytes, val: int, ttl_ns: int, now_ns: int) -> bool:
        _, _, s = self._vt()
        ok = C.c_int()
        if s(self._st.self, key, len(key), val, ttl_ns, now_ns, C.byref(ok)) != 0:
            raise RuntimeError("store error")
        return bool(ok.value)


class AdaptiveOracle(_StoreBase):
    """RateLimiter<AdaptiveStore> -- string keys."""

    def __init__(self, capacity: int = 1000, created_ns: int = 0, *, min_interval_ns: int = 10**9,
                 max_interval_ns: int = 300 * 10**9, max_operations: int = 100_000, auto_cleanup: bool = True):
        super().__init__()
        self._h = lib().tco_adaptive_new(capacity, min_interval_ns, max_interval_ns, max_operations, created_ns)
        if not auto_cleanup:
            lib().tco_adaptive_set_auto_cleanup(self._h, 0)
        self._st = lib().tco_adaptive_as_store(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tco_adaptive_free(self._h)
            self._h = None

    def __len__(self):
        return int(lib().tco_adaptive_len(self._h))

    @property
    def cleanups(self):
        return int(lib().tco_adaptive_cleanups(self._h))

    def force_cleanup(self, now_ns: int):
        lib().tco_adaptive_force_cleanup(self._h, now_ns)

    def batch_keys(self, key_bytes: np.ndarray, key_off: np.ndarray, burst, count, period, quantity, now) -> BatchOut:
        n = len(key_off) - 1
        io, out, keep = _make_io(n, burst, count, period, quantity, now)
        kb = np.ascontiguousarray(key_bytes, dtype=np.uint8)
        ko = np.ascontiguousarray(key_off, dtype=np.uint32)
        lib().tco_batch_keys(C.byref(self._st), kb.ctypes.data, ko.ctypes.data, C.byref(io))
        return out


class DenseOracle(_StoreBase):
    """RateLimiter over a slot-indexed store (key = u32 slot id)."""

    def __init__(self, capacity: int):
        super().__init__()
        self.capacity = capacity
        self._h = lib().tco_dense_new(capacity)
        self._st = lib().tco_dense_as_store(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tco_dense_free(self._h)
            self._h = None

    def batch_slots(self, slots: np.ndarray, burst, count, period, quantity, now, threads: int = 1) -> BatchOut:
        """threads > 1: requests partitioned by slot over that many threads -- same results (keys are independent, a
        key's requests stay in index order), for the full-size tests"""
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        io, out, keep = _make_io(len(sl), burst, count, period, quantity, now)
        if threads > 1:
            lib().tco_batch_slots_mt(C.byref(self._st), sl.ctypes.data, C.byref(io), threads)
        else:
            lib().tco_batch_slots(C.byref(self._st), sl.ctypes.data, C.byref(io))
        return out

    def peek(self, slot: int):
        t, e, o = C.c_int64(), C.c_uint64(), C.c_int()
        lib().tco_dense_peek(self._h, slot, C.byref(t), C.byref(e), C.byref(o))
        return t.value, e.value, bool(o.value)

    def dump(self, first: int = 0, n: Optional[int] = None):
        """-> (tat int64[n], expiry uint64[n] saturated, occupied bool[n]) of slots [first, first + n)"""
        n = self.capacity - first if n is None else n
        tat, exp, occ = np.zeros(n, np.int64), np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        lib().tco_dense_dump(self._h, first, n, tat.ctypes.data, exp.ctypes.data, occ.ctypes.data)
        return tat, exp, occ.astype(bool)

    def sweep(self, now_ns: int) -> int:
        return int(lib().tco_dense_sweep(self._h, now_ns))

    def live(self) -> int:
        return int(lib().tco_dense_live(self._h))


def host_threads(limit: int = 16) -> int:
    """threads this process may actually run (CPU set and cgroup quota), at most `limit`"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, limit))


def batch_keys_mt(threads: int, capacity_per_thread: int, created_ns: int, key_bytes, key_off,
                  burst, count, period, quantity, now, max_operations: int = 1_000_000):
    """Hash-sharded multi-thread AdaptiveStore baseline -> (seconds, BatchOut)."""
    n = len(key_off) - 1
    io, out, keep = _make_io(n, burst, count, period, quantity, now)
    kb = np.ascontiguousarray(key_bytes, dtype=np.uint8)
    ko = np.ascontiguousarray(key_off, dtype=np.uint32)
    secs = lib().tco_batch_keys_mt(threads, capacity_per_thread, max_operations, created_ns, kb.ctypes.data, ko.ctypes.data, C.byref(io))
    return float(secs), out


def reference_shape(num_keys: int = 2000, iterations: int = 400_000):
    """The reference's own library benchmark loop (throttlecrab-server/examples/store_comparison.rs:4-34) on the
    AdaptiveStore port -> (seconds, allowed, blocked)."""
    a, b = C.c_uint64(0), C.c_uint64(0)
    secs = lib().tco_reference_shape(num_keys, iterations, C.byref(a), C.byref(b))
    return float(secs), int(a.value), int(b.value)


def format_keys(ids: np.ndarray, prefix: bytes = b"key_"):
    """`format!("key_{}", id)` arena for a slot-id stream -> (bytes, offsets[n+1])."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    cap = len(ids) * (len(prefix) + 10) + 1
    buf = np.zeros(cap, np.uint8)
    off = np.zeros(len(ids) + 1, np.uint32)
    used = lib().tco_format_keys(prefix, ids.ctypes.data, len(ids), buf.ctypes.data, cap, off.ctypes.data)
    return buf[:max(int(used), 1)], off
