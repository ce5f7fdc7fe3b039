"""Results must not depend on what else the GPU is doing.

Every parity test runs its kernel alone; in the pipeline (BASELINE config 5) the contraction, the beamformer, the power
sums, the CorrAcc adds, the consumers and the ingest scatter share the CUs.  profiles/soak.py repeats that concurrent
pattern on fixed inputs and folds every result into a device-side sum, which must equal N x the result computed alone.
(Round 2: beam_integrate_kernel returned wrong cross-power sums beside the contraction while passing every stand-alone
test -- DESIGN.md 4.10.)"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402


def _load_soak():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "soak.py")
    spec = importlib.util.spec_from_file_location("xeng_soak", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_results_do_not_depend_on_concurrency():
    sk = _load_soak()
    rng = np.random.default_rng(77)
    vin = rng.integers(0, 256, (sk.NTIME_GULP, sk.NCHAN, sk.NSTAND, 2), dtype=np.uint8)
    seq0 = 10 ** 12 + 11
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    order = rng.permutation(len(pk))
    slab = b"".join(pk[i] for i in order)
    lines = []
    res = sk.soak(N=240, packets=(slab, len(pk), len(pk[0]), seq0, vin.reshape(-1)), log=lines.append)
    print("\n".join(lines))
    assert len(res) >= 9
    for name, n, bad in res:
        assert n > 0 and bad == 0, "%s: %d differing words / drops over %d results\n%s" % (name, bad, n, "\n".join(lines))
