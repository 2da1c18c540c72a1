//! throttlecrab on an AMD MI355X: the reference's `Store` trait and `RateLimiter` call surface over
//! libtcgpu.so (include/tcgpu.h).
//!
//! * [`GpuStore`] implements `throttlecrab::Store` (throttlecrab/src/core/store/mod.rs:85-133) call for call, so
//!   `RateLimiter<GpuStore>` works unchanged -- three store operations per request, one kernel launch each: correct,
//!   and slow.
//! * [`GpuRateLimiter`] keeps `RateLimiter::rate_limit`'s signature (rate_limiter.rs:102-110) but runs the whole
//!   decision on the device in one call, and adds `rate_limit_batch`: a slice of requests applied exactly as if
//!   `rate_limit` had been called on them one by one, in order -- what the server's actor
//!   (throttlecrab-server/src/actor.rs:217-236) drains its queue into, see `examples` in INTEGRATION.md.
//!
//! Threading: one owner, like every store of the reference (store/mod.rs:40-43).
pub mod ffi;

use std::time::{Duration, SystemTime, UNIX_EPOCH};

use throttlecrab::{CellError, RateLimitResult, Store};

/// `SystemTime` as the engine's i64 nanoseconds since the epoch; times before 1970 become -1, which the engine
/// answers with status Internal (the reference reads the wall clock there, rate_limiter.rs:126-144).
fn ns(t: SystemTime) -> i64 {
    t.duration_since(UNIX_EPOCH).map(|d| d.as_nanos() as i64).unwrap_or(-1)
}

fn call_error(e: *const ffi::tc_engine, rc: i32) -> String {
    let msg = unsafe {
        let p = ffi::tc_last_error(e);
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    format!("tcgpu error {rc}: {msg}")
}

/// The GPU-resident key store: an engine handle in string-key mode.
pub struct GpuStore {
    e: *mut ffi::tc_engine,
    max_batch: usize,
}

// single owner; the handle may move between threads like any other store
unsafe impl Send for GpuStore {}

impl GpuStore {
    /// `AdaptiveStore::with_capacity` (adaptive_cleanup.rs:93-106): room for `capacity` keys.
    pub fn with_capacity(capacity: usize) -> Self {
        Self::new(capacity, 1 << 20, 0)
    }

    /// The store cleans itself like `AdaptiveStore` (adaptive_cleanup.rs:138-211): the engine runs the reference's
    /// `should_clean` in front of every mutating call and sweeps when it says so (tcgpu.h: tc_set_sweep_policy), so a
    /// `RateLimiter<GpuStore>` does not fill up with expired keys.  Intervals and the operation limit are the server's
    /// defaults (throttlecrab-server/src/config.rs:285-304); `with_sweep_policy` changes them, `TC_SWEEP_NONE` switches
    /// the behaviour off (`cleanup` is then the owner's to call).
    pub fn new(capacity: usize, max_batch: usize, device_id: i32) -> Self {
        let mut s = Self::new_without_policy(capacity, max_batch, device_id);
        let p = ffi::tc_sweep_policy {
            struct_size: std::mem::size_of::<ffi::tc_sweep_policy>() as u32,
            kind: ffi::TC_SWEEP_ADAPTIVE,
            created_ns: ns(SystemTime::now()).max(0),
            min_interval_ns: 5_000_000_000,
            max_interval_ns: 300_000_000_000,
            interval_ns: 0,
            max_operations: 1_000_000,
            map_capacity: 0, // the engine's own table: clean when 3/4 of its slots are taken
            cleanup_probability: 0,
        };
        s.with_sweep_policy(&p).expect("tc_set_sweep_policy");
        s
    }

    pub fn with_sweep_policy(&mut self, p: &ffi::tc_sweep_policy) -> Result<(), String> {
        let rc = unsafe { ffi::tc_set_sweep_policy(self.e, p) };
        if rc == 0 { Ok(()) } else { Err(call_error(self.e, rc)) }
    }

    /// What the engine's own cleanups have done so far (sweeps by trigger, retries after a full table, the policy's state).
    pub fn sweep_stats(&mut self) -> Result<ffi::tc_sweep_info, String> {
        let mut out = ffi::tc_sweep_info { struct_size: std::mem::size_of::<ffi::tc_sweep_info>() as u32, ..Default::default() };
        let rc = unsafe { ffi::tc_sweep_stats(self.e, &mut out) };
        if rc == 0 { Ok(out) } else { Err(call_error(self.e, rc)) }
    }

    pub fn new_without_policy(capacity: usize, max_batch: usize, device_id: i32) -> Self {
        let cfg = ffi::tc_config {
            struct_size: std::mem::size_of::<ffi::tc_config>() as u32,
            flags: ffi::TC_CFG_KEY_MODE,
            device_id,
            reserved0: 0,
            capacity: capacity as u64,
            max_batch: max_batch as u64,
            key_arena_bytes: 0,
        };
        let mut err = 0;
        let e = unsafe { ffi::tc_engine_create(&cfg, &mut err) };
        assert!(!e.is_null(), "tc_engine_create failed: {err}");
        GpuStore { e, max_batch }
    }

    /// `AdaptiveStore::cleanup` (adaptive_cleanup.rs:173-203), explicitly (the store also cleans by itself: see `new`).
    pub fn cleanup(&mut self, now: SystemTime) -> Result<u64, String> {
        let mut removed = 0u64;
        let rc = unsafe { ffi::tc_sweep_expired(self.e, ns(now), &mut removed) };
        if rc == 0 { Ok(removed) } else { Err(call_error(self.e, rc)) }
    }

    pub fn counters(&mut self) -> Result<[u64; ffi::TC_CNT_COUNT], String> {
        let mut out = [0u64; ffi::TC_CNT_COUNT];
        let rc = unsafe { ffi::tc_counters(self.e, out.as_mut_ptr()) };
        if rc == 0 { Ok(out) } else { Err(call_error(self.e, rc)) }
    }
}

impl Drop for GpuStore {
    fn drop(&mut self) {
        unsafe { ffi::tc_engine_destroy(self.e) }
    }
}

impl Store for GpuStore {
    fn compare_and_swap_with_ttl(&mut self, key: &str, old: i64, new: i64, ttl: Duration, now: SystemTime) -> Result<bool, String> {
        let mut ok = 0;
        let rc = unsafe {
            ffi::tc_store_compare_and_swap_with_ttl(self.e, key.as_ptr(), key.len(), old, new, ttl.as_nanos() as u64, ns(now), &mut ok)
        };
        if rc == 0 { Ok(ok != 0) } else { Err(call_error(self.e, rc)) }
    }

    fn get(&self, key: &str, now: SystemTime) -> Result<Option<i64>, String> {
        let (mut value, mut found) = (0i64, 0);
        let rc = unsafe { ffi::tc_store_get(self.e, key.as_ptr(), key.len(), ns(now), &mut value, &mut found) };
        if rc == 0 { Ok((found != 0).then_some(value)) } else { Err(call_error(self.e, rc)) }
    }

    fn set_if_not_exists_with_ttl(&mut self, key: &str, value: i64, ttl: Duration, now: SystemTime) -> Result<bool, String> {
        let mut ok = 0;
        let rc = unsafe {
            ffi::tc_store_set_if_not_exists_with_ttl(self.e, key.as_ptr(), key.len(), value, ttl.as_nanos() as u64, ns(now), &mut ok)
        };
        if rc == 0 { Ok(ok != 0) } else { Err(call_error(self.e, rc)) }
    }
}

/// One request of a batch: the argument list of `RateLimiter::rate_limit`.
pub struct Request<'a> {
    pub key: &'a str,
    pub max_burst: i64,
    pub count_per_period: i64,
    pub period: i64,
    pub quantity: i64,
    pub now: SystemTime,
}

/// `RateLimiter<GpuStore>` with the decision on the device.
pub struct GpuRateLimiter {
    // (fields drop in declaration order: the pinned staging first -- tc_host_free -- then the store, which destroys the engine)
    staging: Staging,
    store: GpuStore,
}

fn decode(r: &ffi::tc_decision, limit: i64, quantity: i64) -> Result<(bool, RateLimitResult), CellError> {
    match r.status {
        ffi::TC_OK => Ok((
            r.allowed != 0,
            RateLimitResult {
                limit,
                remaining: r.remaining,
                reset_after: Duration::from_nanos(r.reset_after_ns as u64),
                retry_after: Duration::from_nanos(r.retry_after_ns as u64),
            },
        )),
        ffi::TC_NEGATIVE_QUANTITY => Err(CellError::NegativeQuantity(quantity)),
        ffi::TC_INVALID_RATE_LIMIT => Err(CellError::InvalidRateLimit),
        _ => Err(CellError::Internal("outside the engine's validated domain".into())),
    }
}

// single owner, like the store (the pinned staging is plain memory owned by this value)
unsafe impl Send for GpuRateLimiter {}

impl GpuRateLimiter {
    pub fn new(store: GpuStore) -> Self {
        GpuRateLimiter { store, staging: Staging::new() }
    }

    pub fn store_mut(&mut self) -> &mut GpuStore {
        &mut self.store
    }

    /// rate_limiter.rs:102-250 in one call.
    pub fn rate_limit(
        &mut self,
        key: &str,
        max_burst: i64,
        count_per_period: i64,
        period: i64,
        quantity: i64,
        now: SystemTime,
    ) -> Result<(bool, RateLimitResult), CellError> {
        let mut r = ffi::tc_result::default();
        let rc = unsafe {
            ffi::tc_rate_limit(self.store.e, key.as_ptr(), key.len(), max_burst, count_per_period, period, quantity, ns(now), &mut r)
        };
        if rc != 0 {
            return Err(CellError::Internal(call_error(self.store.e, rc)));
        }
        let d = ffi::tc_decision {
            remaining: r.remaining,
            reset_after_ns: r.reset_after_ns,
            retry_after_ns: r.retry_after_ns,
            allowed: r.allowed,
            status: r.status,
            pad: [0; 6],
        };
        decode(&d, r.limit, quantity)
    }

    /// The requests of `reqs` applied in order -- bit for bit what calling `rate_limit` on each of them in turn
    /// returns, duplicates of a key inside the slice included -- as one pass through the engine.
    pub fn rate_limit_batch(&mut self, reqs: &[Request]) -> Vec<Result<(bool, RateLimitResult), CellError>> {
        let mut out = Vec::with_capacity(reqs.len());
        for chunk in reqs.chunks(self.store.max_batch.max(1)) {
            self.batch_chunk(chunk, &mut out);
        }
        out
    }

    fn batch_chunk(&mut self, reqs: &[Request], out: &mut Vec<Result<(bool, RateLimitResult), CellError>>) {
        let n = reqs.len();
        if n == 0 {
            return;
        }
        // The request columns are marshalled straight into PINNED staging (tc_host_alloc) that lives as long as the limiter:
        // the engine's transfers to and from pinned memory run at PCIe speed, and a large batch from pinned arrays is
        // pipelined in chunks (its inputs cross the link while earlier chunks are evaluated and their results go back);
        // pageable `Vec`s would be staged once more by the runtime, with the caller blocked meanwhile.
        let key_bytes: usize = reqs.iter().map(|r| r.key.len()).sum();
        self.staging.reserve(n, key_bytes.max(1));
        let st = &mut self.staging;
        let mut at = 0usize;
        // Round 6, TC_B_PLAN_DICT: a server's requests carry a handful of distinct (max_burst, count_per_period, period) triples.
        // The shim -- not the caller: `rate_limit_batch(&[Request])` keeps the reference's shape -- sends each triple once, in a
        // dictionary, and a 16-bit index per request; quantities that fit go as u32.  6 bytes per request cross PCIe instead of
        // 32.  More than 65 536 distinct triples in one chunk, or a quantity outside u32: the wide columns, as before.
        st.plans.clear();
        st.plan_of.clear();
        let mut compact = true;
        unsafe {
            *st.off.ptr = 0;
            for (i, r) in reqs.iter().enumerate() {
                std::ptr::copy_nonoverlapping(r.key.as_ptr(), st.arena.ptr.add(at), r.key.len());
                at += r.key.len();
                *st.off.ptr.add(i + 1) = at as u32;
                *st.burst.ptr.add(i) = r.max_burst;
                *st.count.ptr.add(i) = r.count_per_period;
                *st.period.ptr.add(i) = r.period;
                *st.qty.ptr.add(i) = r.quantity;
                *st.now.ptr.add(i) = ns(r.now);
                if compact {
                    let triple = (r.max_burst, r.count_per_period, r.period);
                    let next = st.plan_of.len();
                    let id = *st.plan_of.entry(triple).or_insert(next);
                    if id == next {
                        st.plans.extend_from_slice(&[triple.0, triple.1, triple.2]);
                    }
                    compact = id < 65536 && r.quantity >= 0 && r.quantity <= u32::MAX as i64;
                    if compact {
                        *st.plan_id.ptr.add(i) = id as u16;
                        *st.qty32.ptr.add(i) = r.quantity as u32;
                    }
                }
            }
            if compact {
                st.dict.reserve(st.plans.len());
                std::ptr::copy_nonoverlapping(st.plans.as_ptr(), st.dict.ptr, st.plans.len());
            }
        }
        let null64: *const i64 = std::ptr::null();
        let b = ffi::tc_batch {
            struct_size: std::mem::size_of::<ffi::tc_batch>() as u32,
            flags: if compact { ffi::TC_B_PLAN_DICT } else { 0 },
            n: n as u64,
            slot: std::ptr::null(),
            key_bytes: st.arena.ptr,
            key_off: st.off.ptr,
            max_burst: if compact { null64 } else { st.burst.ptr },
            count_per_period: if compact { null64 } else { st.count.ptr },
            period: if compact { null64 } else { st.period.ptr },
            quantity: if compact { null64 } else { st.qty.ptr },
            now_ns: st.now.ptr,
            max_burst_scalar: 0,
            count_per_period_scalar: 0,
            period_scalar: 0,
            quantity_scalar: 1,
            now_ns_scalar: 0,
            allowed: std::ptr::null_mut(),
            allowed_bits: std::ptr::null_mut(),
            limit: std::ptr::null_mut(),
            remaining: std::ptr::null_mut(),
            reset_after_ns: std::ptr::null_mut(),
            retry_after_ns: std::ptr::null_mut(),
            status: std::ptr::null_mut(),
            result4: std::ptr::null_mut(),
            decisions: st.dec.ptr,
            order: std::ptr::null_mut(),
            n_segments: 0,
            reserved_seg: 0,
            seg_slot: std::ptr::null(),
            seg_n: std::ptr::null(),
            plan_dict: if compact { st.dict.ptr } else { null64 },
            plan_id: if compact { st.plan_id.ptr } else { std::ptr::null() },
            quantity32: if compact { st.qty32.ptr } else { std::ptr::null() },
            n_plans: if compact { (st.plans.len() / 3) as u32 } else { 0 },
            reserved_dict: 0,
        };
        let rc = unsafe { ffi::tc_rate_limit_batch_keys(self.store.e, &b) };
        if rc != 0 && rc != ffi::TC_E_TABLE_FULL {
            // nothing was applied: every request of the chunk reports the call's failure
            let msg = call_error(self.store.e, rc);
            out.extend((0..n).map(|_| Err(CellError::Internal(msg.clone()))));
            return;
        }
        // TC_E_TABLE_FULL: the keys that fit were applied, the others carry status Internal (with the store's cleanup policy
        // on, the engine has already swept and applied them once more before it reports this: the table is full of LIVE keys)
        out.extend((0..n).map(|i| unsafe { decode(&*st.dec.ptr.add(i), *st.burst.ptr.add(i), *st.qty.ptr.add(i)) }));
    }
}

/// One pinned host array (tc_host_alloc / tc_host_free), grown by doubling and kept across calls.
struct Pinned<T> {
    ptr: *mut T,
    cap: usize,
}

impl<T> Pinned<T> {
    const fn new() -> Self {
        Pinned { ptr: std::ptr::null_mut(), cap: 0 }
    }
    fn reserve(&mut self, n: usize) {
        if n <= self.cap {
            return;
        }
        let cap = n.next_power_of_two().max(1024);
        //  nothing is left behind that a later, smaller batch could write through: cap goes to 0 with the old
        // buffer and comes back only with a new one (a server that catches the panic below keeps a consistent Pinned)
        let old = std::mem::replace(&mut self.ptr, std::ptr::null_mut());
        self.cap = 0;
        let fresh = unsafe {
            ffi::tc_host_free(old as *mut std::os::raw::c_void);
            ffi::tc_host_alloc(cap * std::mem::size_of::<T>()) as *mut T
        };
        assert!(!fresh.is_null(), "tc_host_alloc failed");
        self.ptr = fresh;
        self.cap = cap;
    }
}

impl<T> Drop for Pinned<T> {
    fn drop(&mut self) {
        unsafe { ffi::tc_host_free(self.ptr as *mut std::os::raw::c_void) }
    }
}

/// The marshalling buffers of `rate_limit_batch`: key arena, offsets, the five request columns, the decision records.
struct Staging {
    arena: Pinned<u8>,
    off: Pinned<u32>,
    burst: Pinned<i64>,
    count: Pinned<i64>,
    period: Pinned<i64>,
    qty: Pinned<i64>,
    now: Pinned<i64>,
    dec: Pinned<ffi::tc_decision>,
    // TC_B_PLAN_DICT: the chunk's dictionary (pinned), its 16-bit plan ids and u32 quantities; the encoder's map
    dict: Pinned<i64>,
    plan_id: Pinned<u16>,
    qty32: Pinned<u32>,
    plans: Vec<i64>,
    plan_of: std::collections::HashMap<(i64, i64, i64), usize>,
}

impl Staging {
    fn new() -> Self {
        Staging { arena: Pinned::new(), off: Pinned::new(), burst: Pinned::new(), count: Pinned::new(), period: Pinned::new(), qty: Pinned::new(),
                  now: Pinned::new(), dec: Pinned::new(), dict: Pinned::new(), plan_id: Pinned::new(), qty32: Pinned::new(), plans: Vec::new(),
                  plan_of: std::collections::HashMap::new() }
    }
    fn reserve(&mut self, n: usize, key_bytes: usize) {
        self.arena.reserve(key_bytes);
        self.off.reserve(n + 1);
        self.burst.reserve(n);
        self.count.reserve(n);
        self.period.reserve(n);
        self.qty.reserve(n);
        self.now.reserve(n);
        self.dec.reserve(n);
        self.plan_id.reserve(n);
        self.qty32.reserve(n);
    }
}
