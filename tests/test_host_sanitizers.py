"""Host-side code of libxeng under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build, g++): the triangular
tiling, the persistent kernel's work lists (every (channel, tile group) exactly once, K slices cover K), GetOrder and
Reorder -- the same header (csrc/xcorr_tiling.h) that hipcc compiles into the library.  SURVEY.md section 5 plans this run;
GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_tiling_and_maps_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "tiling_check")
    src = os.path.join(ROOT, "tests", "host", "tiling_check.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-Wall", "-Wextra", "-o", exe, src])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "all properties hold" in r.stdout


# ---- round 5: the native host layer of the rings (csrc/ring.hip + csrc/xeng_util.hip) under ThreadSanitizer and under
# AddressSanitizer + UndefinedBehaviorSanitizer, driven by tests/host/ring_stress.cpp through the C ABI.  Host code only (no
# kernels in those two files): compiled as C++ by the ROCm clang (its ThreadSanitizer runtime knows pthread_cond_clockwait, which
# libstdc++'s condition_variable::wait_for uses; gcc 11's does not and reports a false "double lock").
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CSRC = os.path.join(ROOT, "caltech-bifrost-dsp_amd", "csrc")


def _build_ring_stress(tmp_path, name, sanitize):
    exe = str(tmp_path / name)
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wall", "-Wno-unused-function",
           "-Wno-unused-result"] + sanitize + [os.path.join(CSRC, "ring.hip"), os.path.join(CSRC, "xeng_util.hip"),
                                               os.path.join(ROOT, "tests", "host", "ring_stress.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64",
                                               "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not available")
def test_ring_c_abi_under_thread_sanitizer(tmp_path):
    exe = _build_ring_stress(tmp_path, "ring_stress_tsan", ["-fsanitize=thread"])
    r = subprocess.run([exe, "3"], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ring stress: all scenarios passed" in r.stdout and "ThreadSanitizer" not in r.stderr


@pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang not available")
def test_ring_c_abi_under_asan_ubsan(tmp_path):
    exe = _build_ring_stress(tmp_path, "ring_stress_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    r = subprocess.run([exe, "3"], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ring stress: all scenarios passed" in r.stdout
