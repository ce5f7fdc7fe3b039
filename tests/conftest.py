import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The ring protocol has two implementations (caltech_bifrost_dsp_amd/ring.py: NativeRing on csrc/ring.hip, PyRing): the
# modules that exercise the protocol run on both.
RING_MODULES = ("test_ring", "test_blocks_cpu", "test_ring_lifetime", "test_blocks_round4_cpu")


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in RING_MODULES:
        if "ring_impl" not in metafunc.fixturenames:
            metafunc.fixturenames.append("ring_impl")
        metafunc.parametrize("ring_impl", ["native", "python"], indirect=True)


@pytest.fixture
def ring_impl(request):
    from caltech_bifrost_dsp_amd import ring
    was = ring.IMPLEMENTATION
    ring.IMPLEMENTATION = request.param
    try:
        yield request.param
    finally:
        ring.IMPLEMENTATION = was
