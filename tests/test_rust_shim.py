"""The Rust shim (rust/throttlecrab-gpu) against include/tcgpu.h, without a Rust toolchain: every #[repr(C)]
struct (fields, order, widths), every constant and every `extern "C"` declaration in src/ffi.rs must say what the
header says.  The header is the contract; a field added there and forgotten here fails this test.
Also: the shim implements the reference's Store trait with the reference's signatures
(throttlecrab/src/core/store/mod.rs:85-133)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "tcgpu.h")).read()
FFI = open(os.path.join(ROOT, "rust", "throttlecrab-gpu", "src", "ffi.rs")).read()
LIB = open(os.path.join(ROOT, "rust", "throttlecrab-gpu", "src", "lib.rs")).read()

C_SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int32_t": "i32",
             "int64_t": "i64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void", "double": "f64",
             "tc_engine": "tc_engine", "tc_config": "tc_config", "tc_batch": "tc_batch", "tc_result": "tc_result",
             "tc_decision": "tc_decision", "tc_route": "tc_route", "tc_forward": "tc_forward", "tc_exchange": "tc_exchange",
             "tc_exchange_config": "tc_exchange_config", "tc_sweep_policy": "tc_sweep_policy", "tc_sweep_info": "tc_sweep_info", "tc_engine_info": "tc_engine_info", "tc_shard": "tc_shard", "tc_shard_config": "tc_shard_config"}


def strip_comments(c):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", c, flags=re.S))


def c_type_to_rust(t):
    """'const uint32_t*' -> '*const u32', 'struct tc_decision*' -> '*mut tc_decision', 'int64_t' -> 'i64',
    'const uint32_t* const*' -> '*const *const u32', 'uint32_t* const*' -> '*const *mut u32'"""
    parts = [x.strip() for x in t.replace("struct ", "").strip().split("*")]
    base = parts[0]
    pointee_const = base.startswith("const ") or base.endswith(" const")
    base = base.replace("const", "").strip()
    rust = C_SCALARS[base]
    for qual in parts[1:]:  # one pointer level per '*': what it points to is const if the level to its left said so
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = "const" in qual.split()
    return rust


def c_structs():
    out = {}
    src = strip_comments(HEADER)
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\}\s*\1;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?[\s\*])(\w+)(\[(\d+)\])?$", decl)
            ctype, name, arr = fm.group(1).strip(), fm.group(2), fm.group(4)
            rust = c_type_to_rust(ctype)
            fields.append((name, f"[{rust}; {arr}]" if arr else rust))
        out[m.group(1)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+) \{(.*?)\n\}", FFI, flags=re.S):
        fields = [(f.group(1), " ".join(f.group(2).split())) for f in re.finditer(r"(?:pub )?(\w+): ([^,\n]+),", m.group(2))]
        out[m.group(1)] = fields
    return out


def test_repr_c_structs_match_the_header():
    c, r = c_structs(), rust_structs()
    for name in ("tc_config", "tc_batch", "tc_decision", "tc_result", "tc_route", "tc_forward", "tc_exchange_config", "tc_sweep_policy",
                 "tc_sweep_info", "tc_engine_info", "tc_shard_config"):
        assert name in c and name in r, name
        assert r[name] == c[name], f"{name}: rust {r[name]} != header {c[name]}"
    assert r["tc_engine"] == [("_private", "[u8; 0]")] and r["tc_exchange"] == [("_private", "[u8; 0]")] and r["tc_shard"] == [("_private", "[u8; 0]")]  # opaque


def c_constants():
    src = strip_comments(HEADER)
    vals = {}
    for m in re.finditer(r"#define\s+(TC\w+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", src):
        vals[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"enum\s*\{(.*?)\};", src, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = (x.strip() for x in item.split("="))
                nxt = int(v, 0)
            else:
                k = item
            vals[k] = nxt
            nxt += 1
    return vals


def test_constants_match_the_header():
    c = c_constants()
    rust = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub const (TC\w+): \w+ = (-?(?:0x[0-9a-fA-F]+|\d+));", FFI)}
    assert len(rust) >= 30
    for k, v in rust.items():
        assert k in c, f"{k} is not in tcgpu.h"
        assert c[k] == v, f"{k}: rust {v} != header {c[k]}"
    for k in c:  # every flag and status of the header is bound
        if k.startswith(("TC_B_", "TC_CFG_", "TC_E_", "TC_CNT_", "TC_SWEEP_")) or k in ("TC_OK", "TC_NEGATIVE_QUANTITY", "TC_INVALID_RATE_LIMIT", "TC_INTERNAL"):
            assert k in rust, f"{k} of tcgpu.h is missing in ffi.rs"


def c_functions():
    src = strip_comments(HEADER)
    src = src[src.index("tc_abi_version") - 20:]
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(tc_\w+)\(([^;{]*?)\);", src, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.+?[\s\*])(\w+)(\[\w*\])?$", a)
                t = am.group(1).strip() + ("*" if am.group(3) else "")
                alist.append(c_type_to_rust(t))
        out[name] = (alist, None if ret == "void" else c_type_to_rust(ret))
    return out


def test_extern_declarations_match_the_header():
    c = c_functions()
    block = FFI[FFI.index('extern "C" {'):]
    n = 0
    for m in re.finditer(r"pub fn (tc_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        rust_args = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if ":" in a]
        assert name in c, f"{name} is not declared in tcgpu.h"
        c_args, c_ret = c[name]
        # void* stream / buffers: the header's `void*` is `*mut c_void`; out arrays `uint64_t out[N]` are pointers
        assert rust_args == c_args, f"{name}: rust {rust_args} != header {c_args}"
        assert (ret.strip() if ret else None) == c_ret, f"{name}: return {ret} != {c_ret}"
        n += 1
    bound = set(re.findall(r"pub fn (tc_\w+)", block))
    missing = sorted(set(c) - bound)
    assert not missing, f"tcgpu.h declares {missing}, which ffi.rs does not bind"
    for needed in ("tc_engine_create", "tc_engine_destroy", "tc_rate_limit", "tc_rate_limit_batch_keys", "tc_store_get",
                   "tc_store_compare_and_swap_with_ttl", "tc_store_set_if_not_exists_with_ttl", "tc_sweep_expired"):
        assert re.search(rf"pub fn {needed}\(", block), needed


def test_store_trait_surface():
    """impl Store for GpuStore: the three methods with the reference's signatures (store/mod.rs:96-132)"""
    assert "impl Store for GpuStore" in LIB
    for sig in ("fn compare_and_swap_with_ttl(&mut self, key: &str, old: i64, new: i64, ttl: Duration, now: SystemTime) -> Result<bool, String>",
                "fn get(&self, key: &str, now: SystemTime) -> Result<Option<i64>, String>",
                "fn set_if_not_exists_with_ttl(&mut self, key: &str, value: i64, ttl: Duration, now: SystemTime) -> Result<bool, String>"):
        assert sig in LIB, sig
    assert re.search(r"pub fn rate_limit\(\s*&mut self,\s*key: &str,\s*max_burst: i64,\s*count_per_period: i64,\s*period: i64,\s*quantity: i64,\s*now: SystemTime,?\s*\) -> Result<\(bool, RateLimitResult\), CellError>", LIB)
    assert "pub fn rate_limit_batch(&mut self, reqs: &[Request]) -> Vec<Result<(bool, RateLimitResult), CellError>>" in LIB


def test_the_store_cleans_itself():
    """GpuStore::new hands the engine AdaptiveStore's policy with the server's defaults (config.rs:285-304): the drop-in
    cleans inside its own compare_and_swap / set_if_not_exists, like adaptive_cleanup.rs:229,262"""
    new = LIB[LIB.index("pub fn new(capacity"):LIB.index("pub fn with_sweep_policy")]
    assert "kind: ffi::TC_SWEEP_ADAPTIVE" in new and "min_interval_ns: 5_000_000_000" in new and "max_interval_ns: 300_000_000_000" in new
    assert "max_operations: 1_000_000" in new and "with_sweep_policy(&p)" in new
    assert "ffi::tc_set_sweep_policy(self.e, p)" in LIB
