"""ctypes loader for libtcgpu.so (HIP kernels + C ABI, include/tcgpu.h).

There is NO CPU fallback: if the shared library is missing the import fails
loudly -- build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C throttlecrab_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtcgpu.so")

# per-request status (CellError, throttlecrab/src/core/mod.rs:49-56)
TC_OK, TC_NEGATIVE_QUANTITY, TC_INVALID_RATE_LIMIT, TC_INTERNAL = 0, 1, 2, 3
# call-level return codes
(TC_E_OK, TC_E_INVALID_ARG, TC_E_HIP, TC_E_NOMEM, TC_E_BATCH_TOO_LARGE, TC_E_TABLE_FULL, TC_E_NO_DEVICE,
 TC_E_UNSUPPORTED, TC_E_INVARIANT, TC_E_AGAIN) = (0, -1, -2, -3, -4, -5, -6, -7, -8, -9)
TC_X_NONBLOCKING = 0x1
TC_CFG_KEY_MODE = 0x1
TC_CFG_TRACK_DENIED = 0x2
TC_CFG_FIXED_PARAMS = 0x4
TC_B_DEVICE_PTRS, TC_B_REGISTERED_PARAMS, TC_B_UNIQUE_SLOTS, TC_B_INPUTS_READY, TC_B_GROUPED_OUTPUT = 0x1, 0x2, 0x4, 0x8, 0x10
TC_B_ASYNC = 0x20
TC_B_PLAN_DICT = 0x80
TC_B_OUTPUTS_IDLE = 0x40
TC_ROUTE_AHEAD = 0x1
TC_ROUTE_NO_READERS = 0x2
TC_CNT_NAMES = ("total", "allowed", "denied", "errors", "swept", "batches", "keys_inserted", "live_slots")
TC_CNT_COUNT = 8
TC_STAGE_NAMES = ("prep", "sort", "eval", "commit", "pack", "hash", "bucket_hist", "bucket_scan", "bucket_scatter",
                  "bucket_eval")
TC_STAGE_COUNT = 10


class tc_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("device_id", C.c_int32),
                ("reserved0", C.c_int32), ("capacity", C.c_uint64), ("max_batch", C.c_uint64),
                ("key_arena_bytes", C.c_uint64)]


class tc_batch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("n", C.c_uint64),
                ("slot", C.c_void_p), ("key_bytes", C.c_void_p), ("key_off", C.c_void_p),
                ("max_burst", C.c_void_p), ("count_per_period", C.c_void_p), ("period", C.c_void_p),
                ("quantity", C.c_void_p), ("now_ns", C.c_void_p),
                ("max_burst_scalar", C.c_int64), ("count_per_period_scalar", C.c_int64),
                ("period_scalar", C.c_int64), ("quantity_scalar", C.c_int64), ("now_ns_scalar", C.c_int64),
                ("allowed", C.c_void_p), ("allowed_bits", C.c_void_p), ("limit", C.c_void_p),
                ("remaining", C.c_void_p), ("reset_after_ns", C.c_void_p), ("retry_after_ns", C.c_void_p),
                ("status", C.c_void_p), ("result4", C.c_void_p), ("decisions", C.c_void_p), ("order", C.c_void_p),
                ("n_segments", C.c_uint32), ("reserved_seg", C.c_uint32), ("seg_slot", C.c_void_p), ("seg_n", C.c_void_p),
                ("plan_dict", C.c_void_p), ("plan_id", C.c_void_p), ("quantity32", C.c_void_p), ("n_plans", C.c_uint32), ("reserved_dict", C.c_uint32)]


class tc_forward(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("world", C.c_uint32), ("src", C.c_void_p), ("count", C.c_void_p),
                ("dst", C.c_void_p), ("stream", C.c_void_p)]


class tc_route(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("world", C.c_uint32), ("keys_per_shard", C.c_uint64), ("n", C.c_uint64),
                ("global_id", C.c_void_p), ("only", C.c_int32), ("flags", C.c_uint32), ("out_slot", C.c_void_p),
                ("out_pos", C.c_void_p), ("out_count", C.c_void_p), ("stream", C.c_void_p), ("out_count_host", C.c_void_p), ("tag", C.c_uint32), ("reserved1", C.c_uint32), ("out_dst", C.c_void_p)]


class tc_exchange_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32), ("ring", C.c_uint32), ("seg_cap", C.c_uint32),
                ("flags", C.c_uint32), ("keys_per_shard", C.c_uint64), ("inbox", C.c_void_p), ("mail", C.c_void_p), ("done", C.c_void_p)]


TC_SWEEP_NONE, TC_SWEEP_ADAPTIVE, TC_SWEEP_PERIODIC, TC_SWEEP_PROBABILISTIC = 0, 1, 2, 3


class tc_sweep_policy(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_uint32), ("created_ns", C.c_int64), ("min_interval_ns", C.c_int64),
                ("max_interval_ns", C.c_int64), ("interval_ns", C.c_int64), ("max_operations", C.c_uint64),
                ("map_capacity", C.c_uint64), ("cleanup_probability", C.c_uint64)]


class tc_sweep_info(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_uint32), ("sweeps", C.c_uint64), ("sweeps_by_time", C.c_uint64),
                ("sweeps_by_operations", C.c_uint64), ("sweeps_by_size", C.c_uint64), ("sweeps_for_room", C.c_uint64),
                ("retries", C.c_uint64), ("feed_waits", C.c_uint64), ("operations", C.c_uint64), ("entries", C.c_uint64),
                ("last_removed", C.c_uint64), ("current_interval_ns", C.c_int64), ("next_cleanup_ns", C.c_int64)]


class tc_engine_info(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("side_streams_probed", C.c_uint32), ("grouping_streams_wanted", C.c_uint32),
                ("grouping_streams", C.c_uint32), ("key_stream", C.c_uint32), ("candidates_tried", C.c_uint32),
                ("rejected_same_queue", C.c_uint32), ("rejected_same_pipe", C.c_uint32), ("kept_second_best", C.c_uint32),
                ("probes_assumed", C.c_uint32), ("pipelining_degraded", C.c_uint32), ("scratch_sets", C.c_uint32),
                ("grouping_path", C.c_uint32), ("range_path_possible", C.c_uint32), ("range_hint_requests", C.c_uint64),
                ("range_hint_largest", C.c_uint64), ("host_chunk_requests", C.c_uint64), ("batches", C.c_uint64),
                ("hot_slots", C.c_uint64), ("hot_batches", C.c_uint64), ("probes_pooled", C.c_uint64), ("sweeps_aside", C.c_uint64)]


class tc_shard_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32), ("ring", C.c_uint32), ("keys_per_shard", C.c_uint64),
                ("max_global", C.c_uint64)]


class tc_result(C.Structure):
    _fields_ = [("limit", C.c_int64), ("remaining", C.c_int64), ("reset_after_ns", C.c_int64),
                ("retry_after_ns", C.c_int64), ("allowed", C.c_uint8), ("status", C.c_uint8)]


# every symbol include/tcgpu.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tc_abi_version": (C.c_uint32, []),
    "tc_engine_create": (C.c_void_p, [C.POINTER(tc_config), C.POINTER(C.c_int)]),
    "tc_engine_destroy": (None, [C.c_void_p]),
    "tc_engine_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tc_synchronize": (C.c_int, [C.c_void_p]),
    "tc_last_error": (C.c_char_p, [C.c_void_p]),
    "tc_register_params": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tc_register_params_uniform": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "tc_rate_limit_batch_slots": (C.c_int, [C.c_void_p, C.POINTER(tc_batch)]),
    "tc_rate_limit_batch_keys": (C.c_int, [C.c_void_p, C.POINTER(tc_batch)]),
    "tc_wait_batches": (C.c_int, [C.c_void_p, C.c_uint32]),
    "tc_host_alloc": (C.c_void_p, [C.c_size_t]),
    "tc_host_free": (None, [C.c_void_p]),
    "tc_rate_limit": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.c_int64, C.POINTER(tc_result)]),
    "tc_sweep_expired": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_uint64)]),
    "tc_set_sweep_policy": (C.c_int, [C.c_void_p, C.POINTER(tc_sweep_policy)]),
    "tc_sweep_stats": (C.c_int, [C.c_void_p, C.POINTER(tc_sweep_info)]),
    "tc_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tc_counters_refresh": (C.c_int, [C.c_void_p]),
    "tc_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tc_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "tc_counters_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "tc_store_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.POINTER(C.c_int64),
                               C.POINTER(C.c_int)]),
    "tc_store_compare_and_swap_with_ttl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_int64,
                                                     C.c_uint64, C.c_int64, C.POINTER(C.c_int)]),
    "tc_store_set_if_not_exists_with_ttl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_uint64,
                                                      C.c_int64, C.POINTER(C.c_int)]),
    "tc_read_state": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tc_lookup_slot": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "tc_top_denied": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "tc_denied_reset": (C.c_int, [C.c_void_p]),
    "tc_top_denied_keys": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "tc_debug_fail_copy": (C.c_int, [C.c_void_p, C.c_uint32]),
    "tc_debug_break_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "tc_debug_occupy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "tc_debug_check_keys": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tc_selfcheck": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tc_engine_info_get": (C.c_int, [C.c_void_p, C.POINTER(tc_engine_info)]),
    "tc_snapshot_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tc_snapshot_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "tc_route_batch": (C.c_int, [C.c_void_p, C.POINTER(tc_route)]),
    "tc_forward_segments": (C.c_int, [C.c_void_p, C.POINTER(tc_forward)]),
    "tc_exchange_create": (C.c_int, [C.c_void_p, C.POINTER(tc_exchange_config), C.POINTER(C.c_void_p)]),
    "tc_exchange_destroy": (C.c_int, [C.c_void_p]),
    "tc_exchange_route": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "tc_exchange_post": (C.c_int, [C.c_void_p, C.c_uint64]),
    "tc_exchange_collect": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "tc_exchange_evaluate": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(tc_batch), C.POINTER(C.c_uint64)]),
    "tc_exchange_step": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(tc_batch), C.POINTER(C.c_uint64)]),
    "tc_exchange_poll": (C.c_int, [C.c_void_p]),
    "tc_exchange_wait_ns": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tc_shard_create": (C.c_int, [C.c_void_p, C.POINTER(tc_shard_config), C.POINTER(C.c_void_p)]),
    "tc_shard_destroy": (C.c_int, [C.c_void_p]),
    "tc_shard_route": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "tc_shard_evaluate": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(tc_batch), C.POINTER(C.c_uint64)]),
    "tc_shard_step": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(tc_batch), C.POINTER(C.c_uint64)]),
    "tc_shard_wait_ns": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tc_route_host": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tc_route_inverse": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tc_route_keys_host": (C.c_int, [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tc_slot_keys": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


def load():
    """Load libtcgpu.so and bind every declared symbol.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch ships its own libamdhip64 and libtcgpu.so is linked against
    # the system one (same SONAME).  Whichever is loaded first serves both, and with the system
    # runtime loaded first torch's device bookkeeping and ours disagree (tc_engine_create then fails
    # with TC_E_NO_DEVICE on a GPU box).  So torch, when present, is imported before the library.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built.  There is no CPU fallback; "
            "run `make -C throttlecrab_amd/csrc` (or __graft_entry__.build()).")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
