"""Runs the C++ host-mirror test program (tests/cpp/test_host_mirror.cpp, written
after the reference's own Rust tests) against libtcgpu.so on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_program():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build_host_mirror_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
