"""CPU-side boundary checks: the C-ABI library loads and exports every symbol
include/tcgpu.h declares (no compute calls without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tcgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import throttlecrab_amd as t
    from throttlecrab_amd import _lib
    lib = t.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libtcgpu.so does not export {name}"
    assert set(_lib.SYMBOLS) == set(declared), set(_lib.SYMBOLS) ^ set(declared)
    assert lib.tc_abi_version() == 1
    assert ctypes.sizeof(_lib.tc_batch) == 8 + 8 + 8 * 8 + 5 * 8 + 8 * 8  # incl. result4
    assert ctypes.sizeof(_lib.tc_config) == 40


def test_engine_create_fails_loudly_without_gpu():
    import pytest
    import torch
    import throttlecrab_amd as t
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(t.TcError) as ei:
        t.Engine(1000, 1000)
    assert ei.value.code == -6  # TC_E_NO_DEVICE: no silent CPU fallback


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "throttlecrab_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"
