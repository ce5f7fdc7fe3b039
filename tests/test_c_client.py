"""The boundary is a C ABI: a plain-C client (examples/c_abi_demo.c, no HIP headers, gcc) builds against
include/xeng.h + libxeng.so, and on a GPU replays the reference's golden vectors with the reference's call
sequence (xgpu_test.py:76-89)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "caltech-bifrost-dsp_amd")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "examples", "c_abi_demo.c"),
                    "-L", LIBDIR, "-lxeng", "-Wl,-rpath," + LIBDIR], check=True, capture_output=True)
    return exe


def test_c_client_builds_with_gcc(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libxeng.so")):
        pytest.skip("libxeng.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = _build(tmp_path)
    out = subprocess.run(["ldd", exe], check=True, capture_output=True, text=True).stdout
    assert "libxeng.so" in out and "not found" not in out.split("libxeng.so")[1].splitlines()[0]


@pytest.mark.gpu
def test_c_client_replays_golden(tmp_path, golden_dir):
    exe = _build(tmp_path)
    r = subprocess.run([exe, os.path.join(golden_dir, "in_8t_4c_16s_2p_deadbeef.dat"),
                        os.path.join(golden_dir, "corr_8t_4a_4c_16s_2p_deadbeef.dat")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "4352 visibilities checked" in r.stdout and "PASS" in r.stdout
