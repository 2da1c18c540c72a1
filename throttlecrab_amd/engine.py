"""Thin object wrapper over the C ABI (include/tcgpu.h).

Inputs may be numpy arrays (host pointers: staged by the library, the call
returns with results) or torch CUDA tensors (device pointers: asynchronous on
the engine's stream, results written into CUDA tensors).
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L


class TcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"tcgpu error {code}: {msg}")
        self.code = code


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class BatchResult:
    """Columnar (bool, RateLimitResult) + status for one batch."""
    __slots__ = ("allowed", "allowed_bits", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status",
                 "result4", "decisions", "order")

    def __init__(self, **arrays):
        for s in self.__slots__:
            setattr(self, s, arrays.pop(s, None))
        if arrays:
            raise TypeError(f"unknown result fields: {sorted(arrays)}")


_NP_DTYPES = {"allowed": np.uint8, "allowed_bits": np.uint64, "limit": np.int64, "remaining": np.int64,
              "reset_after_ns": np.int64, "retry_after_ns": np.int64, "status": np.uint8, "result4": np.int64,
              "decisions": np.int64}


class _PinnedBlock:
    """Frees one tc_host_alloc block when the last numpy view of it is gone (it hangs off the ctypes
    array that is the views' base)."""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        try:
            self._lib.tc_host_free(C.c_void_p(self._ptr))
        except Exception:
            pass


class Engine:
    ALL_FIELDS = ("allowed", "limit", "remaining", "reset_after_ns", "retry_after_ns", "status")
    # the same information with RateLimitResult as one 32-byte record per request
    # (result4[i] = limit, remaining, reset_after_ns, retry_after_ns)
    RECORD_FIELDS = ("allowed", "status", "result4")
    # tc_decision records: decisions[i] = remaining, reset_after_ns, retry_after_ns, allowed | status << 8
    # (everything but `limit`, which the caller knows) -- ONE scattered store per request
    DECISION_FIELDS = ("decisions",)

    @staticmethod
    def unpack_decisions(dec):
        """[n*4] int64 decisions column -> dict of the five fields."""
        d = (dec.cpu().numpy() if _is_torch(dec) else np.asarray(dec)).reshape(-1, 4)
        return {"remaining": d[:, 0], "reset_after_ns": d[:, 1], "retry_after_ns": d[:, 2],
                "allowed": (d[:, 3] & 0xFF).astype(np.uint8), "status": ((d[:, 3] >> 8) & 0xFF).astype(np.uint8)}

    def __init__(self, capacity: int, max_batch: int = 1 << 20, device: int = 0, key_mode: bool = False,
                 key_arena_bytes: int = 0, track_denied: bool = False, fixed_params: bool = False):
        self._lib = L.load()
        cfg = L.tc_config()
        cfg.struct_size = C.sizeof(L.tc_config)
        cfg.flags = ((L.TC_CFG_KEY_MODE if key_mode else 0) | (L.TC_CFG_TRACK_DENIED if track_denied else 0) |
                     (L.TC_CFG_FIXED_PARAMS if fixed_params else 0))
        cfg.device_id = device
        cfg.capacity = capacity
        cfg.max_batch = max_batch
        cfg.key_arena_bytes = key_arena_bytes
        err = C.c_int(0)
        self._async_keep = collections.deque()
        self._h = self._lib.tc_engine_create(C.byref(cfg), C.byref(err))
        if not self._h:
            raise TcError(err.value, "tc_engine_create failed (is an MI355X visible and libtcgpu.so built for gfx950?)")
        self.capacity = capacity
        self.max_batch = max_batch
        self.device = device
        self.key_mode = key_mode
        self.check_on_close = False
        self._children = []   # objects that hold a pointer to this engine (sharded.ExchangeRank): closed before it

    def debug_fail_copy(self, nth: int):
        """test hook: the nth staging copy from now fails (tc_debug_fail_copy)"""
        self._check(self._lib.tc_debug_fail_copy(self._h, nth))

    def debug_break_wait(self, on: bool = True):
        """test hook: the next uniform batch withholds one cross-row announcement (tc_debug_break_wait)"""
        self._check(self._lib.tc_debug_break_wait(self._h, 1 if on else 0))

    def debug_occupy(self, blocks: int, microseconds: int, cu_mask=None, lds_bytes: int = 0):
        """test hook: a filler kernel of `blocks` workgroups that stay resident for `microseconds` each, on a stream limited to
        the CUs of `cu_mask` (8 uint32 words; None: all) -- tc_debug_occupy; returns at once"""
        m = None
        if cu_mask is not None:
            m = (C.c_uint32 * 8)(*[int(w) & 0xFFFFFFFF for w in cu_mask])
        self._check(self._lib.tc_debug_occupy(self._h, m, blocks, lds_bytes, microseconds))

    def debug_check_keys(self) -> int:
        """test hook (string keys): number of inconsistencies between the key table's entries, records, position column and free
        stack (tc_debug_check_keys); 0 on a healthy table"""
        v = C.c_uint64(0)
        self._check(self._lib.tc_debug_check_keys(self._h, C.byref(v)))
        return int(v.value)

    GROUPING_PATHS = ("none yet", "range path", "LSD passes", "bucket path", "no grouping", "range path, hot slots peeled")

    def info(self) -> dict:
        """tc_engine_info_get: the pipeline's health -- side streams wanted / kept, what the probe rejected and why, whether
        pipelined batches currently overlap (pipelining_degraded), the grouping path of the last batch and the range hint"""
        r = L.tc_engine_info()
        r.struct_size = C.sizeof(L.tc_engine_info)
        self._check(self._lib.tc_engine_info_get(self._h, C.byref(r)))
        d = {k: int(getattr(r, k)) for k, _ in L.tc_engine_info._fields_ if k != "struct_size"}
        d["grouping_path"] = self.GROUPING_PATHS[min(r.grouping_path, 5)]
        return d

    def selfcheck(self) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.tc_selfcheck(self._h, C.byref(v)))
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            for ch in list(getattr(self, "_children", ())):
                ch.close()
            self._children = []
            if self.check_on_close:
                bad = self.selfcheck()
                assert bad == 0, f"engine flagged {bad} internal invariant violations"
            self._lib.tc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.tc_last_error(self._h)
            raise TcError(rc, msg.decode() if msg else "")

    # ---- configuration ----
    def set_stream(self, hip_stream: Optional[int]):
        self._check(self._lib.tc_engine_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def use_torch_stream(self):
        """Run on torch's CURRENT stream.  torch's default stream has handle 0, which tc_set_stream reads as "the
        engine's own stream": work torch enqueues afterwards is then NOT ordered behind the engine's -- synchronise
        (torch.cuda.synchronize()) before reading results, or call this inside `with torch.cuda.stream(s)`."""
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        self._check(self._lib.tc_synchronize(self._h))

    def register_params_uniform(self, max_burst: int, count_per_period: int, period: int):
        self._check(self._lib.tc_register_params_uniform(self._h, max_burst, count_per_period, period))

    def register_params(self, max_burst, count_per_period, period, slots=None):
        b = np.ascontiguousarray(max_burst, dtype=np.int64)
        c = np.ascontiguousarray(count_per_period, dtype=np.int64)
        p = np.ascontiguousarray(period, dtype=np.int64)
        s = None if slots is None else np.ascontiguousarray(slots, dtype=np.uint32)
        self._check(self._lib.tc_register_params(self._h, len(b), None if s is None else s.ctypes.data,
                                                 b.ctypes.data, c.ctypes.data, p.ctypes.data))

    # ---- batches ----
    def _column(self, batch, name, val, n, dev, keep):
        """Set batch.<name> (pointer) or batch.<name>_scalar."""
        if val is None:
            return
        if _is_torch(val):
            if not dev:
                raise ValueError("torch CUDA columns need CUDA slots/keys (device-pointer batch)")
            assert val.is_cuda and val.is_contiguous() and val.numel() == n, name
            keep.append(val)
            setattr(batch, name, val.data_ptr())
            return
        arr = np.asarray(val)
        if arr.ndim == 0 or (arr.size == 1 and n != 1):
            setattr(batch, name + "_scalar", int(arr.reshape(-1)[0]))
            return
        if dev:
            raise ValueError(f"{name}: per-request column must be a CUDA tensor in a device-pointer batch")
        arr = np.ascontiguousarray(arr, dtype=np.int64)
        if arr.size != n:
            raise ValueError(f"{name}: length {arr.size} != batch size {n}")
        keep.append(arr)
        setattr(batch, name, arr.ctypes.data)

    @staticmethod
    def encode_plans(max_burst, count_per_period, period):
        """What the shims do for TC_B_PLAN_DICT: the three per-request columns -> (dictionary int64[k, 3], ids uint16[n]), or
        None when the batch holds more than 65 536 distinct triples (the wide columns then)."""
        rows = np.stack([np.asarray(max_burst, np.int64), np.asarray(count_per_period, np.int64), np.asarray(period, np.int64)], axis=1)
        dic, ids = np.unique(rows, axis=0, return_inverse=True)
        if len(dic) > 65536:
            return None
        return np.ascontiguousarray(dic), np.ascontiguousarray(ids.reshape(-1), dtype=np.uint16)

    def _prepare(self, n, dev, max_burst, count_per_period, period, quantity, now_ns, registered, unique,
                 want, out: Optional[BatchResult], inputs_ready=False, grouped=False, async_=False, outputs_idle=False, plan_dict=None):
        keep = []
        b = L.tc_batch()
        b.struct_size = C.sizeof(L.tc_batch)
        b.n = n
        flags = 0
        if dev:
            flags |= L.TC_B_DEVICE_PTRS
        if registered:
            flags |= L.TC_B_REGISTERED_PARAMS
        if unique:
            flags |= L.TC_B_UNIQUE_SLOTS
        if inputs_ready:
            if not dev:
                raise ValueError("inputs_ready applies to device-pointer batches")
            flags |= L.TC_B_INPUTS_READY
        if grouped:
            flags |= L.TC_B_GROUPED_OUTPUT
        if outputs_idle:
            if not inputs_ready:
                raise ValueError("outputs_idle goes with inputs_ready")
            flags |= L.TC_B_OUTPUTS_IDLE
        if async_:
            if dev:
                raise ValueError("async_ applies to host-array batches (CUDA-tensor batches are asynchronous anyway)")
            flags |= L.TC_B_ASYNC
        if plan_dict is not None:
            # TC_B_PLAN_DICT: (dictionary int64[k, 3], ids uint16[n], quantities uint32[n] or None) -- numpy for host batches, CUDA
            # tensors (int64 / int16 / int32 views of the same bits) for device batches
            flags |= L.TC_B_PLAN_DICT
            dic, ids, q32 = plan_dict
            if max_burst is not None or count_per_period is not None or period is not None or (q32 is not None and quantity is not None):
                raise ValueError("plan_dict replaces max_burst / count_per_period / period (and quantity32 replaces quantity)")
            for name, arr, dt in (("plan_dict", dic, np.int64), ("plan_id", ids, np.uint16), ("quantity32", q32, np.uint32)):
                if arr is None:
                    continue
                if _is_torch(arr):
                    assert dev and arr.is_cuda and arr.is_contiguous(), name
                    keep.append(arr)
                    setattr(b, name, arr.data_ptr())
                else:
                    if dev:
                        raise ValueError(f"{name}: a CUDA tensor in a device-pointer batch")
                    if arr.dtype != dt or not arr.flags["C_CONTIGUOUS"]:
                        if async_:
                            raise ValueError(f"async_: {name} must be a C-contiguous {np.dtype(dt).name} array (no hidden copies)")
                        arr = np.ascontiguousarray(arr, dtype=dt)
                    keep.append(arr)
                    setattr(b, name, arr.ctypes.data)
            b.n_plans = (dic.numel() if _is_torch(dic) else dic.size) // 3
        b.flags = flags
        b.quantity_scalar = 1
        if not registered and plan_dict is None:
            if max_burst is None or count_per_period is None or period is None:
                raise ValueError("max_burst/count_per_period/period required unless registered=True")
            self._column(b, "max_burst", max_burst, n, dev, keep)
            self._column(b, "count_per_period", count_per_period, n, dev, keep)
            self._column(b, "period", period, n, dev, keep)
        self._column(b, "quantity", quantity, n, dev, keep)
        if now_ns is None:
            raise ValueError("now_ns is required")
        self._column(b, "now_ns", now_ns, n, dev, keep)
        res = out or BatchResult()
        if grouped:  # output row k belongs to request res.order[k]
            if res.order is None or (len(res.order) if not _is_torch(res.order) else res.order.numel()) != n:
                if dev:
                    import torch
                    res.order = torch.empty(n, dtype=torch.int32, device=f"cuda:{self.device}")
                else:
                    res.order = np.zeros(n, dtype=np.uint32)
            keep.append(res.order)
            b.order = res.order.data_ptr() if _is_torch(res.order) else res.order.ctypes.data
        for name in want:
            cur = getattr(res, name)
            ln = (n + 63) // 64 if name == "allowed_bits" else (4 * n if name in ("result4", "decisions") else n)
            if cur is not None and (cur.numel() if _is_torch(cur) else cur.size) < ln:
                if async_:
                    raise ValueError(f"async_: the {name} array holds fewer than {ln} entries (no hidden reallocation may be made)")
                cur = None  # a result set reused for a larger batch: a new array, never a write past the old one's end
            if cur is None:
                if dev:
                    import torch
                    tdt = torch.uint8 if _NP_DTYPES[name] == np.uint8 else torch.int64
                    cur = torch.empty(ln, dtype=tdt, device=f"cuda:{self.device}")
                else:
                    cur = np.zeros(ln, dtype=_NP_DTYPES[name])
                setattr(res, name, cur)
            keep.append(cur)
            setattr(b, name, cur.data_ptr() if _is_torch(cur) else cur.ctypes.data)
        return b, res, keep

    def rate_limit_batch_slots(self, slots, *, max_burst=None, count_per_period=None, period=None, quantity=None,
                               now_ns=None, registered=False, unique=False, want=ALL_FIELDS,
                               out: Optional[BatchResult] = None, inputs_ready=False, grouped=False,
                               async_=False, outputs_idle=False, segments=None, plan_dict=None) -> BatchResult:
        """rate_limit_batch over pre-resolved slots (sequential semantics, index order).
        segments=[(tensor, count), ...] (with slots=None): the slot column in pieces, e.g. one per source GPU.
        async_=True (TC_B_ASYNC, host arrays): only enqueue -- transfers and evaluation overlap with
        other batches; `slots`, per-request columns and the arrays of `out` must be uint32 / int64 /
        uint8 numpy arrays (pinned: host_alloc) that stay untouched until wait_batches() says so.
        grouped=True (TC_B_GROUPED_OUTPUT): output rows come in the engine's evaluation order and
        res.order[k] is the request index of row k.
        inputs_ready=True (TC_B_INPUTS_READY): the CUDA `slots` tensor is already complete and
        stays untouched until the results are ready, so the engine may group this batch on its
        auxiliary stream while earlier batches are still being evaluated.
        outputs_idle=True (TC_B_OUTPUTS_IDLE, with inputs_ready): nothing enqueued earlier reads or writes the
        arrays of `out` (every batch in flight has its own), so the engine may initialise them early."""
        dev = _is_torch(slots) or segments is not None
        keep = []
        seg_arrays = None
        if segments is not None:
            # a slot column in pieces (tc_batch.seg_slot): [(CUDA int32 tensor, count), ...] taken one after the other
            import torch
            assert slots is None and 0 < len(segments) <= 64
            if __debug__ and len(segments) <= 4:  # (a rank of a large deployment passes dozens of pieces every few microseconds)
                for tns, cnt in segments:
                    assert tns.is_cuda and tns.is_contiguous() and tns.dtype in (torch.int32, torch.uint32) and 0 <= cnt <= tns.numel()
            ptrs = (C.c_void_p * len(segments))(*[tns.data_ptr() for tns, _ in segments])
            cnts = (C.c_uint32 * len(segments))(*[int(cnt) for _, cnt in segments])
            keep += [ptrs, cnts] + [tns for tns, _ in segments]
            seg_arrays = (ptrs, cnts)
            n = int(sum(cnt for _, cnt in segments))
            sp = None
        elif dev:
            import torch
            assert slots.is_cuda and slots.is_contiguous() and slots.dtype in (torch.int32, torch.uint32)
            n = slots.numel()
            sp = slots.data_ptr()
        else:
            if async_ and not (isinstance(slots, np.ndarray) and slots.dtype == np.uint32 and slots.flags.c_contiguous):
                raise ValueError("async_: slots must be a C-contiguous uint32 numpy array (no hidden copy may be made)")
            sl = np.ascontiguousarray(slots, dtype=np.uint32)
            keep.append(sl)
            n = sl.size
            sp = sl.ctypes.data
        b, res, k2 = self._prepare(n, dev, max_burst, count_per_period, period, quantity, now_ns, registered,
                                   unique, want, out, inputs_ready, grouped, async_, outputs_idle, plan_dict=plan_dict)
        b.slot = sp
        if seg_arrays is not None:
            b.n_segments = len(segments)
            b.seg_slot = C.cast(seg_arrays[0], C.c_void_p)
            b.seg_n = C.cast(seg_arrays[1], C.c_void_p)
        if n:
            self._check(self._lib.tc_rate_limit_batch_slots(self._h, C.byref(b)))
            if async_:
                self._async_keep.append((keep, k2))  # arrays stay referenced while the batch is in flight
        return res

    def wait_batches(self, max_in_flight: int = 0) -> None:
        """Block until at most `max_in_flight` async_ batches are still incomplete (tc_wait_batches)."""
        self._check(self._lib.tc_wait_batches(self._h, max_in_flight))
        while len(self._async_keep) > max_in_flight:
            self._async_keep.popleft()

    def host_alloc(self, n: int, dtype) -> np.ndarray:
        """Pinned host array for async_ batches (tc_host_alloc); freed when the array is garbage-collected."""
        dt = np.dtype(dtype)
        nbytes = max(1, int(n) * dt.itemsize)
        ptr = self._lib.tc_host_alloc(nbytes)
        if not ptr:
            raise MemoryError(f"tc_host_alloc({nbytes}) failed")
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        buf._owner = _PinnedBlock(self._lib, ptr)
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def rate_limit_batch_keys(self, key_bytes, key_off, *, max_burst=None, count_per_period=None, period=None,
                              quantity=None, now_ns=None, want=ALL_FIELDS,
                              out: Optional[BatchResult] = None, inputs_ready=False, async_=False, plan_dict=None) -> BatchResult:
        """rate_limit_batch over string keys (arena bytes + offsets[n+1]).  inputs_ready: see
        rate_limit_batch_slots (here it covers the key arena and the offsets).  async_: see
        rate_limit_batch_slots (key_bytes: uint8, key_off: uint32 numpy arrays, no hidden copies)."""
        dev = _is_torch(key_bytes)
        keep = []
        if dev:
            assert key_bytes.is_cuda and key_off.is_cuda
            n = key_off.numel() - 1
            kb, ko = key_bytes.data_ptr(), key_off.data_ptr()
        else:
            if async_:
                for a, dt in ((key_bytes, np.uint8), (key_off, np.uint32)):
                    if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous):
                        raise ValueError("async_: key_bytes / key_off must be C-contiguous uint8 / uint32 numpy arrays")
            kbytes = np.ascontiguousarray(key_bytes, dtype=np.uint8)
            koff = np.ascontiguousarray(key_off, dtype=np.uint32)
            keep += [kbytes, koff]
            n = koff.size - 1
            kb, ko = kbytes.ctypes.data, koff.ctypes.data
        b, res, k2 = self._prepare(n, dev, max_burst, count_per_period, period, quantity, now_ns, False, False,
                                   want, out, inputs_ready, False, async_, plan_dict=plan_dict)
        b.key_bytes = kb
        b.key_off = ko
        if n:
            self._check(self._lib.tc_rate_limit_batch_keys(self._h, C.byref(b)))
            if async_:
                self._async_keep.append((keep, k2))
        return res

    def route_batch(self, global_ids, world: int, only: int = -1, want_pos: bool = False, out=None, stream=None, ahead: bool = False,
                    host_counts=None, tag: int = 0, no_readers: bool = False, out_dst=None):
        """tc_route_batch: of a CUDA tensor of global key ids (int32 / uint32), keep what destination `only` owns
        (only = -1: every destination's segment, one after the other) as shard-local slots, in request order.
        -> (slots int32[n], pos int32[n] or None, counts int32[world]) CUDA tensors; the first counts[only]
        (resp. sum(counts)) entries are valid.  Asynchronous on the engine's stream: read `counts` after a sync.
        `out`: (slots, pos or None, counts) tensors to reuse.  `stream`: a torch.cuda.Stream to run the router on
        instead; the caller orders it before the batch call that reads `slots`.
        ahead=True (TC_ROUTE_AHEAD): the engine runs the router on one of its grouping streams, beside the
        evaluations of earlier batches.  host_counts: a pinned uint32 array of world + 1 words (host_alloc) that
        receives the counts and then `tag` in its last word -- poll that word instead of synchronising."""
        import torch
        assert global_ids.is_cuda and global_ids.is_contiguous() and global_ids.dtype in (torch.int32, torch.uint32)
        n = global_ids.numel()
        dev = global_ids.device
        keep_dst = None
        if out_dst is not None:
            # segment d goes straight to out_dst[d] (CUDA int32 tensors; peer memory allowed) instead of into `slots`
            assert only == -1 and len(out_dst) == world
            keep_dst = (C.c_void_p * world)(*[d.data_ptr() for d in out_dst])
            if out is None:
                out = (None, None, torch.empty(world, dtype=torch.int32, device=dev))
        if out is None:
            # (torch.empty, not zeros: a fill enqueued on torch's stream is not ordered with the engine's stream when
            # that is the engine's own -- see use_torch_stream -- and the router writes every entry anyway)
            out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev) if want_pos else None,
                   torch.empty(world, dtype=torch.int32, device=dev))
        slots, pos, counts = out
        r = L.tc_route()
        r.struct_size = C.sizeof(L.tc_route)
        r.world, r.keys_per_shard, r.n, r.only = world, self.capacity, n, only
        r.global_id, r.out_slot, r.out_count = global_ids.data_ptr(), (slots.data_ptr() if slots is not None else None), counts.data_ptr()
        if keep_dst is not None:
            r.out_dst = C.cast(keep_dst, C.c_void_p)
        r.out_pos = pos.data_ptr() if pos is not None else None
        r.stream = stream.cuda_stream if stream is not None else None
        r.flags = (L.TC_ROUTE_AHEAD if ahead else 0) | (L.TC_ROUTE_NO_READERS if no_readers else 0)
        if host_counts is not None:
            assert host_counts.dtype == np.uint32 and host_counts.size >= world + 1
            r.out_count_host, r.tag = host_counts.ctypes.data, tag
        self._check(self._lib.tc_route_batch(self._h, C.byref(r)))
        return slots, pos, counts

    def forward_segments(self, src, counts, dsts, stream=None):
        """tc_forward_segments: the router's segments (tc_route_batch with only = -1; `src` = its slot tensor, `counts` = the
        host copy of its counts) -> dsts[d][:counts[d]] for every destination d, in one launch (dsts: CUDA int32 tensors,
        peer memory allowed).  stream: a torch.cuda.Stream (default: the engine's stream)."""
        world = len(counts)
        f = L.tc_forward()
        f.struct_size = C.sizeof(L.tc_forward)
        f.world = world
        f.src = src.data_ptr()
        cnt = (C.c_uint32 * world)(*[int(c) for c in counts])
        dst = (C.c_void_p * world)(*[d.data_ptr() for d in dsts])
        f.count, f.dst = C.cast(cnt, C.c_void_p), C.cast(dst, C.c_void_p)
        f.stream = stream.cuda_stream if stream is not None else None
        self._check(self._lib.tc_forward_segments(self._h, C.byref(f)))

    def rate_limit(self, key: bytes, max_burst: int, count_per_period: int, period: int, quantity: int, now_ns: int):
        """RateLimiter::rate_limit -> (status, allowed, limit, remaining, reset_after_ns, retry_after_ns)."""
        r = L.tc_result()
        self._check(self._lib.tc_rate_limit(self._h, key, len(key), max_burst, count_per_period, period, quantity,
                                            now_ns, C.byref(r)))
        return (r.status, bool(r.allowed), r.limit, r.remaining, r.reset_after_ns, r.retry_after_ns)

    # ---- maintenance / introspection ----
    def sweep_expired_async(self, now_ns: int) -> None:
        """Enqueue the sweep on the engine's stream without waiting (count -> counters()["swept"])."""
        self._check(self._lib.tc_sweep_expired(self._h, now_ns, None))

    def sweep_expired(self, now_ns: int) -> int:
        removed = C.c_uint64(0)
        self._check(self._lib.tc_sweep_expired(self._h, now_ns, C.byref(removed)))
        return int(removed.value)

    def set_sweep_policy(self, kind="adaptive", created_ns: int = 0, min_interval_ns: int = 0, max_interval_ns: int = 0,
                         interval_ns: int = 0, max_operations: int = 0, map_capacity: int = 0, cleanup_probability: int = 0):
        """tc_set_sweep_policy: the engine cleans by itself in front of its own mutating calls, when the reference's store of
        that kind would (adaptive_cleanup.rs:138-211 / periodic.rs:128-142 / probabilistic.rs:110-125).  kind: "adaptive" |
        "periodic" | "probabilistic" | None (off, the default of a fresh engine).  Zeros mean the reference's defaults."""
        if kind is None or kind == "none":
            self._check(self._lib.tc_set_sweep_policy(self._h, None))
            return
        p = L.tc_sweep_policy()
        p.struct_size = C.sizeof(L.tc_sweep_policy)
        p.kind = {"adaptive": L.TC_SWEEP_ADAPTIVE, "periodic": L.TC_SWEEP_PERIODIC, "probabilistic": L.TC_SWEEP_PROBABILISTIC}[kind]
        p.created_ns, p.min_interval_ns, p.max_interval_ns, p.interval_ns = created_ns, min_interval_ns, max_interval_ns, interval_ns
        p.max_operations, p.map_capacity, p.cleanup_probability = max_operations, map_capacity, cleanup_probability
        self._check(self._lib.tc_set_sweep_policy(self._h, C.byref(p)))

    def sweep_stats(self) -> dict:
        """tc_sweep_stats: what the engine's own cleanups did (sweeps by trigger, retries, the policy's state)."""
        r = L.tc_sweep_info()
        r.struct_size = C.sizeof(L.tc_sweep_info)
        self._check(self._lib.tc_sweep_stats(self._h, C.byref(r)))
        d = {k: int(getattr(r, k)) for k, _ in L.tc_sweep_info._fields_ if k != "struct_size"}
        d["kind"] = ("none", "adaptive", "periodic", "probabilistic")[r.kind]
        return d

    def counters(self) -> dict:
        arr = (C.c_uint64 * L.TC_CNT_COUNT)()
        self._check(self._lib.tc_counters(self._h, arr))
        return {k: int(arr[i]) for i, k in enumerate(L.TC_CNT_NAMES)}

    def counters_refresh(self):
        """Fold the sharded device counters into the device counter block (async)."""
        self._check(self._lib.tc_counters_refresh(self._h))

    def profile_enable(self, on: bool = True):
        self._check(self._lib.tc_profile_enable(self._h, 1 if on else 0))

    def profile_read(self) -> dict:
        """-> {stage: (total_ms, launches)} measured with HIP events on the engine's stream."""
        ms = (C.c_double * L.TC_STAGE_COUNT)()
        calls = (C.c_uint64 * L.TC_STAGE_COUNT)()
        self._check(self._lib.tc_profile_read(self._h, ms, calls))
        return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(L.TC_STAGE_NAMES)}

    def counters_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self._lib.tc_counters_device_ptr(self._h, C.byref(p)))
        return int(p.value)

    def read_state(self, first: int, n: int):
        tat = np.zeros(n, np.int64)
        exp = np.zeros(n, np.uint64)
        self._check(self._lib.tc_read_state(self._h, first, n, tat.ctypes.data, exp.ctypes.data))
        return tat, exp

    def top_denied(self, k: int = 100):
        """Top denied keys (metrics.rs:24-76): [(slot, count)] most denied first; in key mode
        [(key bytes, count)]."""
        slots = np.zeros(max(k, 1), np.uint32)
        counts = np.zeros(max(k, 1), np.uint64)
        n = C.c_uint32(0)
        self._check(self._lib.tc_top_denied(self._h, k, slots.ctypes.data, counts.ctypes.data, C.byref(n)))
        slots, counts = slots[: n.value], counts[: n.value]
        if not self.key_mode:
            return [(int(s), int(c)) for s, c in zip(slots, counts)]
        # key mode: by KEY, the keys a sweep has unbound included (tc_top_denied_keys)
        kk = max(k, 1)
        off = np.zeros(kk + 1, np.uint32)
        cnt = np.zeros(kk, np.uint64)
        buf = np.zeros(max(1, 64 * kk), np.uint8)
        while True:
            rc = self._lib.tc_top_denied_keys(self._h, k, buf.ctypes.data, buf.size, off.ctypes.data, cnt.ctypes.data, C.byref(n))
            if rc == L.TC_E_INVALID_ARG and buf.size < (1 << 30):
                buf = np.zeros(buf.size * 8, np.uint8)
                continue
            self._check(rc)
            break
        return [(bytes(buf[off[i]:off[i + 1]]), int(cnt[i])) for i in range(n.value)]

    def snapshot_save(self, path: str):
        self._check(self._lib.tc_snapshot_save(self._h, path.encode()))

    def snapshot_load(self, path: str):
        self._check(self._lib.tc_snapshot_load(self._h, path.encode()))

    def denied_reset(self):
        self._check(self._lib.tc_denied_reset(self._h))

    def lookup_slot(self, key: bytes) -> int:
        s = C.c_int64(-1)
        self._check(self._lib.tc_lookup_slot(self._h, key, len(key), C.byref(s)))
        return int(s.value)

    # ---- `trait Store` shims (store/mod.rs:85-133) ----
    def get(self, key: bytes, now_ns: int):
        v, f = C.c_int64(), C.c_int()
        self._check(self._lib.tc_store_get(self._h, key, len(key), now_ns, C.byref(v), C.byref(f)))
        return v.value if f.value else None

    def compare_and_swap_with_ttl(self, key: bytes, old: int, new: int, ttl_ns: int, now_ns: int) -> bool:
        ok = C.c_int()
        self._check(self._lib.tc_store_compare_and_swap_with_ttl(self._h, key, len(key), old, new, ttl_ns, now_ns,
                                                                 C.byref(ok)))
        return bool(ok.value)

    def set_if_not_exists_with_ttl(self, key: bytes, value: int, ttl_ns: int, now_ns: int) -> bool:
        ok = C.c_int()
        self._check(self._lib.tc_store_set_if_not_exists_with_ttl(self._h, key, len(key), value, ttl_ns, now_ns,
                                                                  C.byref(ok)))
        return bool(ok.value)
