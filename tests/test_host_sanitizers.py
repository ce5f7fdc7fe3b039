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
