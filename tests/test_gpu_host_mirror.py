"""Runs the C++ host-mirror test program (tests/cpp/test_host_mirror.cpp, written
after the reference's own Rust tests) against libtcgpu.so on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stale(exe, *sources):
    """the program is missing, or older than its source, a header under include/ or libtcgpu.so (the tree travels with its built
    programs: one built before the last header change must not be what is tested)"""
    import glob
    if not os.path.exists(exe):
        return True
    deps = list(sources) + glob.glob(os.path.join(ROOT, "include", "*")) + [os.path.join(ROOT, "throttlecrab_amd", "libtcgpu.so")]
    return any(os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(exe) for d in deps)


@pytest.mark.gpu
def test_cpp_host_mirror_program():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    if _stale(exe, exe + ".cpp"):
        import __graft_entry__ as g
        g.build_host_mirror_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


@pytest.mark.gpu
def test_cpp_actor_and_resp_pipeline_program():
    """Batch-draining actor (actor.rs mirror) and the RESP THROTTLE pipeline on the GPU."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_actor_resp")
    if _stale(exe, exe + ".cpp"):
        import __graft_entry__ as g
        g.build_actor_resp_tests()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_resp_parser_program():
    """RESP parsing / argument validation / hardening limits: no GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_resp_parse")
    src = os.path.join(ROOT, "tests", "cpp", "test_resp_parse.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_metrics_text_program():
    """Prometheus text == throttlecrab-server/src/metrics.rs:236-311: no GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_metrics_text")
    src = os.path.join(ROOT, "tests", "cpp", "test_metrics_text.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_actor_logic_program():
    """The actor's channel / batching / pipelining logic over a stand-in limiter (queue order == evaluation
    order, replies matched, at most FLIGHTS batches in flight, back-pressure, errors, shutdown): no GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_actor_logic")
    src = os.path.join(ROOT, "tests", "cpp", "test_actor_logic.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_sweep_policy_program():
    """Periodic / probabilistic / adaptive sweep schedulers == the reference stores' cleanup cadence
    (periodic.rs:128-142, probabilistic.rs:110-125, adaptive_cleanup.rs:138-211): no GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_sweep_policy")
    src = os.path.join(ROOT, "tests", "cpp", "test_sweep_policy.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_cpp_kernel_math_properties_on_host():
    """gcra_math.hpp compiled for the host: closed form == sequence, the host-side proof behind the
    direct-store evaluation, late readers harmless (tests/cpp/test_math_host.hip). No GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_math_host")
    src = os.path.join(ROOT, "tests", "cpp", "test_math_host.hip")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-ffp-contract=off", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
