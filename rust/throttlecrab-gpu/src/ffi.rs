//! include/tcgpu.h, declaration for declaration.  Field order, widths and constant values are checked against
//! the header by tests/test_rust_shim.py (no Rust toolchain needed for that).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const TCGPU_ABI_VERSION: u32 = 1;

// per-request status == CellError (throttlecrab/src/core/mod.rs:49-56)
pub const TC_OK: u8 = 0;
pub const TC_NEGATIVE_QUANTITY: u8 = 1;
pub const TC_INVALID_RATE_LIMIT: u8 = 2;
pub const TC_INTERNAL: u8 = 3;

// call-level return codes
pub const TC_E_OK: c_int = 0;
pub const TC_E_INVALID_ARG: c_int = -1;
pub const TC_E_HIP: c_int = -2;
pub const TC_E_NOMEM: c_int = -3;
pub const TC_E_BATCH_TOO_LARGE: c_int = -4;
pub const TC_E_TABLE_FULL: c_int = -5;
pub const TC_E_NO_DEVICE: c_int = -6;
pub const TC_E_UNSUPPORTED: c_int = -7;
pub const TC_E_INVARIANT: c_int = -8;
pub const TC_E_AGAIN: c_int = -9;

pub const TC_CFG_KEY_MODE: u32 = 0x1;
pub const TC_CFG_TRACK_DENIED: u32 = 0x2;
pub const TC_CFG_FIXED_PARAMS: u32 = 0x4;

pub const TC_B_DEVICE_PTRS: u32 = 0x1;
pub const TC_B_REGISTERED_PARAMS: u32 = 0x2;
pub const TC_B_UNIQUE_SLOTS: u32 = 0x4;
pub const TC_B_INPUTS_READY: u32 = 0x8;
pub const TC_B_GROUPED_OUTPUT: u32 = 0x10;
pub const TC_B_ASYNC: u32 = 0x20;
pub const TC_B_OUTPUTS_IDLE: u32 = 0x40;
pub const TC_B_PLAN_DICT: u32 = 0x80;

pub const TC_ROUTE_AHEAD: u32 = 0x1;
pub const TC_ROUTE_NO_READERS: u32 = 0x2;
pub const TC_X_NONBLOCKING: u32 = 0x1;

// tc_sweep_policy.kind: which of the reference's stores the engine cleans like
pub const TC_SWEEP_NONE: u32 = 0;
pub const TC_SWEEP_ADAPTIVE: u32 = 1;
pub const TC_SWEEP_PERIODIC: u32 = 2;
pub const TC_SWEEP_PROBABILISTIC: u32 = 3;

pub const TC_CNT_TOTAL: usize = 0;
pub const TC_CNT_ALLOWED: usize = 1;
pub const TC_CNT_DENIED: usize = 2;
pub const TC_CNT_ERRORS: usize = 3;
pub const TC_CNT_SWEPT: usize = 4;
pub const TC_CNT_BATCHES: usize = 5;
pub const TC_CNT_KEYS_INSERTED: usize = 6;
pub const TC_CNT_LIVE_SLOTS: usize = 7;
pub const TC_CNT_COUNT: usize = 8;

#[repr(C)]
pub struct tc_engine {
    _private: [u8; 0],
}

/// one rank's side of the multi-GPU exchange (opaque)
#[repr(C)]
pub struct tc_exchange {
    _private: [u8; 0],
}

/// tc_set_sweep_policy: maybe_clean_expired (adaptive_cleanup.rs:205-211) inside the engine's own mutating calls
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct tc_sweep_policy {
    pub struct_size: u32,
    pub kind: u32,
    pub created_ns: i64,
    pub min_interval_ns: i64,
    pub max_interval_ns: i64,
    pub interval_ns: i64,
    pub max_operations: u64,
    pub map_capacity: u64,
    pub cleanup_probability: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct tc_sweep_info {
    pub struct_size: u32,
    pub kind: u32,
    pub sweeps: u64,
    pub sweeps_by_time: u64,
    pub sweeps_by_operations: u64,
    pub sweeps_by_size: u64,
    pub sweeps_for_room: u64,
    pub retries: u64,
    pub feed_waits: u64,
    pub operations: u64,
    pub entries: u64,
    pub last_removed: u64,
    pub current_interval_ns: i64,
    pub next_cleanup_ns: i64,
}

/// tc_engine_info_get: do pipelined batches overlap, and which grouping path is in use
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct tc_engine_info {
    pub struct_size: u32,
    pub side_streams_probed: u32,
    pub grouping_streams_wanted: u32,
    pub grouping_streams: u32,
    pub key_stream: u32,
    pub candidates_tried: u32,
    pub rejected_same_queue: u32,
    pub rejected_same_pipe: u32,
    pub kept_second_best: u32,
    pub probes_assumed: u32,
    pub pipelining_degraded: u32,
    pub scratch_sets: u32,
    pub grouping_path: u32,
    pub range_path_possible: u32,
    pub range_hint_requests: u64,
    pub range_hint_largest: u64,
    pub host_chunk_requests: u64,
    pub batches: u64,
    pub hot_slots: u64,
    pub hot_batches: u64,
    pub probes_pooled: u64,
    pub sweeps_aside: u64,
}

/// one rank's side of `replicate` mode (opaque)
#[repr(C)]
pub struct tc_shard {
    _private: [u8; 0],
}

#[repr(C)]
pub struct tc_shard_config {
    pub struct_size: u32,
    pub rank: u32,
    pub world: u32,
    pub ring: u32,
    pub keys_per_shard: u64,
    pub max_global: u64,
}

#[repr(C)]
pub struct tc_exchange_config {
    pub struct_size: u32,
    pub rank: u32,
    pub world: u32,
    pub ring: u32,
    pub seg_cap: u32,
    pub flags: u32,
    pub keys_per_shard: u64,
    pub inbox: *const *mut u32,
    pub mail: *mut u32,
    pub done: *mut i64,
}

#[repr(C)]
pub struct tc_config {
    pub struct_size: u32,
    pub flags: u32,
    pub device_id: i32,
    pub reserved0: i32,
    pub capacity: u64,
    pub max_batch: u64,
    pub key_arena_bytes: u64,
}

#[repr(C)]
pub struct tc_batch {
    pub struct_size: u32,
    pub flags: u32,
    pub n: u64,
    pub slot: *const u32,
    pub key_bytes: *const u8,
    pub key_off: *const u32,
    pub max_burst: *const i64,
    pub count_per_period: *const i64,
    pub period: *const i64,
    pub quantity: *const i64,
    pub now_ns: *const i64,
    pub max_burst_scalar: i64,
    pub count_per_period_scalar: i64,
    pub period_scalar: i64,
    pub quantity_scalar: i64,
    pub now_ns_scalar: i64,
    pub allowed: *mut u8,
    pub allowed_bits: *mut u64,
    pub limit: *mut i64,
    pub remaining: *mut i64,
    pub reset_after_ns: *mut i64,
    pub retry_after_ns: *mut i64,
    pub status: *mut u8,
    pub result4: *mut i64,
    pub decisions: *mut tc_decision,
    pub order: *mut u32,
    pub n_segments: u32,
    pub reserved_seg: u32,
    pub seg_slot: *const *const u32,
    pub seg_n: *const u32,
    // TC_B_PLAN_DICT (round 6)
    pub plan_dict: *const i64,
    pub plan_id: *const u16,
    pub quantity32: *const u32,
    pub n_plans: u32,
    pub reserved_dict: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct tc_decision {
    pub remaining: i64,
    pub reset_after_ns: i64,
    pub retry_after_ns: i64,
    pub allowed: u8,
    pub status: u8,
    pub pad: [u8; 6],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct tc_result {
    pub limit: i64,
    pub remaining: i64,
    pub reset_after_ns: i64,
    pub retry_after_ns: i64,
    pub allowed: u8,
    pub status: u8,
}

#[repr(C)]
pub struct tc_route {
    pub struct_size: u32,
    pub world: u32,
    pub keys_per_shard: u64,
    pub n: u64,
    pub global_id: *const u32,
    pub only: i32,
    pub flags: u32,
    pub out_slot: *mut u32,
    pub out_pos: *mut u32,
    pub out_count: *mut u32,
    pub stream: *mut c_void,
    pub out_count_host: *mut u32,
    pub tag: u32,
    pub reserved1: u32,
    pub out_dst: *const *mut u32,
}

#[repr(C)]
pub struct tc_forward {
    pub struct_size: u32,
    pub world: u32,
    pub src: *const u32,
    pub count: *const u32,
    pub dst: *const *mut u32,
    pub stream: *mut c_void,
}

extern "C" {
    pub fn tc_abi_version() -> u32;
    pub fn tc_engine_create(cfg: *const tc_config, err: *mut c_int) -> *mut tc_engine;
    pub fn tc_engine_destroy(e: *mut tc_engine);
    pub fn tc_synchronize(e: *mut tc_engine) -> c_int;
    pub fn tc_last_error(e: *const tc_engine) -> *const c_char;
    pub fn tc_register_params_uniform(e: *mut tc_engine, max_burst: i64, count_per_period: i64, period: i64) -> c_int;
    pub fn tc_rate_limit_batch_slots(e: *mut tc_engine, b: *const tc_batch) -> c_int;
    pub fn tc_rate_limit_batch_keys(e: *mut tc_engine, b: *const tc_batch) -> c_int;
    pub fn tc_wait_batches(e: *mut tc_engine, max_in_flight: u32) -> c_int;
    pub fn tc_host_alloc(bytes: usize) -> *mut c_void;
    pub fn tc_host_free(p: *mut c_void);
    pub fn tc_rate_limit(
        e: *mut tc_engine,
        key: *const u8,
        key_len: usize,
        max_burst: i64,
        count_per_period: i64,
        period: i64,
        quantity: i64,
        now_ns: i64,
        out: *mut tc_result,
    ) -> c_int;
    pub fn tc_sweep_expired(e: *mut tc_engine, now_ns: i64, removed: *mut u64) -> c_int;
    pub fn tc_shard_create(e: *mut tc_engine, c: *const tc_shard_config, out: *mut *mut tc_shard) -> c_int;
    pub fn tc_shard_destroy(x: *mut tc_shard) -> c_int;
    pub fn tc_shard_route(x: *mut tc_shard, step: u64, global_id: *const u32, n: u64) -> c_int;
    pub fn tc_shard_evaluate(x: *mut tc_shard, step: u64, tmpl: *const tc_batch, decided: *mut u64) -> c_int;
    pub fn tc_shard_step(
        x: *mut tc_shard,
        step: u64,
        global_id_ahead: *const u32,
        n_ahead: u64,
        route_ahead: u32,
        tmpl: *const tc_batch,
        decided: *mut u64,
    ) -> c_int;
    pub fn tc_shard_wait_ns(x: *mut tc_shard, out: *mut u64) -> c_int;
    pub fn tc_engine_info_get(e: *mut tc_engine, out: *mut tc_engine_info) -> c_int;
    pub fn tc_set_sweep_policy(e: *mut tc_engine, p: *const tc_sweep_policy) -> c_int;
    pub fn tc_sweep_stats(e: *mut tc_engine, out: *mut tc_sweep_info) -> c_int;
    pub fn tc_counters(e: *mut tc_engine, out: *mut u64) -> c_int;
    pub fn tc_store_get(e: *mut tc_engine, key: *const u8, key_len: usize, now_ns: i64, value: *mut i64, found: *mut c_int) -> c_int;
    pub fn tc_store_compare_and_swap_with_ttl(
        e: *mut tc_engine,
        key: *const u8,
        key_len: usize,
        old_value: i64,
        new_value: i64,
        ttl_ns: u64,
        now_ns: i64,
        swapped: *mut c_int,
    ) -> c_int;
    pub fn tc_store_set_if_not_exists_with_ttl(
        e: *mut tc_engine,
        key: *const u8,
        key_len: usize,
        value: i64,
        ttl_ns: u64,
        now_ns: i64,
        was_set: *mut c_int,
    ) -> c_int;
    pub fn tc_route_batch(e: *mut tc_engine, r: *const tc_route) -> c_int;
    pub fn tc_forward_segments(e: *mut tc_engine, f: *const tc_forward) -> c_int;
    pub fn tc_route_host(world: u32, keys_per_shard: u64, n: u64, global_id: *const u32, owner: *mut u32, slot: *mut u32) -> c_int;
    pub fn tc_exchange_create(e: *mut tc_engine, c: *const tc_exchange_config, out: *mut *mut tc_exchange) -> c_int;
    pub fn tc_exchange_destroy(x: *mut tc_exchange) -> c_int;
    pub fn tc_exchange_route(x: *mut tc_exchange, step: u64, global_id: *const u32, n: u32) -> c_int;
    pub fn tc_exchange_post(x: *mut tc_exchange, step: u64) -> c_int;
    pub fn tc_exchange_collect(x: *mut tc_exchange, step: u64, counts: *mut u32) -> c_int;
    pub fn tc_exchange_evaluate(x: *mut tc_exchange, step: u64, tmpl: *const tc_batch, decided: *mut u64) -> c_int;
    pub fn tc_exchange_step(x: *mut tc_exchange, step: u64, global_id_ahead: *const u32, n_ahead: u32, route_ahead: u32, post_ahead: u32, tmpl: *const tc_batch, decided: *mut u64) -> c_int;
    pub fn tc_exchange_poll(x: *mut tc_exchange) -> c_int;
    pub fn tc_exchange_wait_ns(x: *mut tc_exchange, out: *mut u64) -> c_int;
    pub fn tc_route_keys_host(world: u32, n: u64, key_bytes: *const u8, key_off: *const u32, owner: *mut u32) -> c_int;
    pub fn tc_route_inverse(world: u32, keys_per_shard: u64, n: u64, owner: *const u32, slot: *const u32, global_id: *mut u64) -> c_int;
    pub fn tc_engine_set_stream(e: *mut tc_engine, hip_stream: *mut c_void) -> c_int;
    pub fn tc_register_params(
        e: *mut tc_engine,
        n: u64,
        slots: *const u32,
        max_burst: *const i64,
        count_per_period: *const i64,
        period: *const i64,
    ) -> c_int;
    pub fn tc_lookup_slot(e: *mut tc_engine, key: *const u8, key_len: usize, slot: *mut i64) -> c_int;
    pub fn tc_counters_refresh(e: *mut tc_engine) -> c_int;
    pub fn tc_counters_device_ptr(e: *mut tc_engine, dptr: *mut *mut c_void) -> c_int;
    pub fn tc_top_denied(e: *mut tc_engine, k: u32, slots: *mut u32, counts: *mut u64, n_out: *mut u32) -> c_int;
    pub fn tc_denied_reset(e: *mut tc_engine) -> c_int;
    pub fn tc_top_denied_keys(e: *mut tc_engine, k: u32, key_bytes: *mut u8, key_bytes_cap: usize, key_off: *mut u32, counts: *mut u64, n_out: *mut u32) -> c_int;
    pub fn tc_slot_keys(e: *mut tc_engine, n: u32, slots: *const u32, key_bytes: *mut u8, key_bytes_cap: usize, key_off: *mut u32) -> c_int;
    pub fn tc_read_state(e: *mut tc_engine, first: u64, n: u64, tat: *mut i64, expiry: *mut u64) -> c_int;
    pub fn tc_profile_enable(e: *mut tc_engine, on: c_int) -> c_int;
    pub fn tc_profile_read(e: *mut tc_engine, total_ms: *mut f64, calls: *mut u64) -> c_int;
    pub fn tc_selfcheck(e: *mut tc_engine, violations: *mut u64) -> c_int;
    pub fn tc_debug_fail_copy(e: *mut tc_engine, nth: u32) -> c_int;
    pub fn tc_debug_break_wait(e: *mut tc_engine, on: u32) -> c_int;
    pub fn tc_debug_occupy(e: *mut tc_engine, cu_mask: *const u32, blocks: u32, lds_bytes: u32, microseconds: u64) -> c_int;
    pub fn tc_debug_check_keys(e: *mut tc_engine, inconsistencies: *mut u64) -> c_int;
    pub fn tc_snapshot_save(e: *mut tc_engine, path: *const c_char) -> c_int;
    pub fn tc_snapshot_load(e: *mut tc_engine, path: *const c_char) -> c_int;
}
