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


@pytest.mark.parametrize("env,rounds", [
    ({}, 240),                          # the shipped kernels
    ({"XENG_BEAM": "bf16x3"}, 60),      # bf16 MFMA beamformer beside the contraction's epilogue
    ({"XENG_BEAM": "f32"}, 60),         # fp32 MFMA beamformer
    ({"XENG_TILING": "64"}, 60),        # the 64x64-tile tiling of the fused kernel (17 groups, 7-vs-6 items)
    ({"XENG_KLOOP": "16"}, 60),         # the eight-wave 16x16x64 contraction kernel (round 5, opt-in)
    ({"XENG_SLAB_TABLES": "1"}, 60),    # packet slabs through offset tables / packet indices from the first launch (round 5)
    ({"XENG_SLAB_TABLES": "0"}, 60),    # ... and never (round 4's zero-fill + scatter of every irregular slab)
    ({"XENG_RAW": "0"}, 60),            # two-pass X-engine (corner turn + xcorr_mfma_kernel)
])
def test_results_do_not_depend_on_concurrency(env, rounds):
    for k, v in env.items():
        os.environ[k] = v
    try:
        _soak_once(rounds)
    finally:
        for k in env:
            del os.environ[k]


def _soak_once(rounds):
    sk = _load_soak()
    rng = np.random.default_rng(77)
    vin = rng.integers(0, 256, (sk.NTIME_GULP, sk.NCHAN, sk.NSTAND, 2), dtype=np.uint8)
    seq0 = 10 ** 12 + 11
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=0)
    order = rng.permutation(len(pk))
    slab = b"".join(pk[i] for i in order)
    lines = []
    res = sk.soak(N=rounds, packets=(slab, len(pk), len(pk[0]), seq0, vin.reshape(-1)), log=lines.append)
    print("\n".join(lines))
    assert len(res) >= (9 if os.environ.get("XENG_RAW") == "0" else 10)      # (the two-pass X-engine has no fused long accumulation)
    for name, n, bad in res:
        assert n > 0 and bad == 0, "%s: %d differing words / drops over %d results\n%s" % (name, bad, n, "\n".join(lines))


def test_full_size_blocks_match_standalone_calls():
    """The Python blocks on in-repo rings at BASELINE config-2 / config-4 size, all concurrent on one GPU
    (gpu-input read in place by Corr and by Beamform; Corr -> CorrAcc; Beamform -> BeamformSumBeams): every span that
    leaves a block equals, bit for bit, what the same C-ABI call returns alone on an idle GPU for the same gulps."""
    import threading
    import time

    import caltech_bifrost_dsp_amd  # noqa: F401
    from caltech_bifrost_dsp_amd import ffi
    from caltech_bifrost_dsp_amd.blocks import Beamform, BeamformSumBeams, Corr, CorrAcc
    from caltech_bifrost_dsp_amd.ring import Ring
    from tests.gpu_util import Xgpu
    from tests.pipeline_util import LOG, Sink, Source, source_header

    C, S, g, acc, nbeam, ns, nrep = 96, 352, 480, 2400, 32, 24, 4
    ninput, G = 2 * S, acc // g
    gulp_bytes = g * C * ninput
    rng = np.random.default_rng(4242)
    vin = rng.integers(0, 256, (G * g, C, S, 2), dtype=np.uint8)          # one integration; the stream repeats it nrep times
    W = (rng.uniform(-17, 17, (C, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (C, nbeam, ninput))).astype(np.complex64)
    x = Xgpu(S, C, g, max_gulps=G)
    ref_vis = x.run(vin, use_async=True)
    x.close()

    r_in = Ring("gpu-input", space="cuda")
    r_vis, r_slow = Ring("corr-output", space="cuda"), Ring("corr-slow-output", space="cuda_host")
    r_bf, r_pow = Ring("bf-output", space="cuda"), Ring("bf-pow-output", space="cuda_host")
    r_in.resize(gulp_bytes, total_span=2 * G * gulp_bytes)
    corr = Corr(LOG, r_in, r_vis, ntime_gulp=g, nchan=C, npol=2, nstand=S, acc_len=acc, autostartat=0, gpu=0)
    cacc = CorrAcc(LOG, r_vis, r_slow, nchan=C, npol=2, nstand=S, acc_len=2 * acc, autostartat=0, gpu=0)
    bf = Beamform(LOG, r_in, r_bf, nchan=C, nbeam=nbeam, ninput=ninput, ntime_gulp=g, gpu=0)
    sb = BeamformSumBeams(LOG, r_bf, r_pow, nchan=C, ntime_gulp=g, ntime_sum=ns, gpu=0)
    bf.gains_cpu[...] = W                                                 # (uploaded at the start of the sequence)

    class CheckSink(threading.Thread):
        """compares every span with `expect` as it arrives and keeps only the verdicts"""
        def __init__(self, ring, gulp, expect):
            super().__init__(daemon=True)
            self.gulp, self.expect, self.verdicts = gulp, expect, []
            self._gen = ring.read(guarantee=True)

        def run(self):
            for iseq in self._gen:
                for ispan in iseq.read(self.gulp):
                    if ispan.size == self.gulp:
                        self.verdicts.append(bool(np.array_equal(ispan.data.numpy().view(np.int32).reshape(-1), self.expect)))

    fast = CheckSink(r_vis, corr.ogulp_size, ref_vis.reshape(-1))
    slow = CheckSink(r_slow, cacc.ogulp_size, (2 * ref_vis.astype(np.int64)).astype(np.int32).reshape(-1))
    beams, power = Sink(r_bf, g * C * nbeam * 8), Sink(r_pow, (nbeam // 2) * (g // ns) * C * 16)
    data = np.tile(vin.reshape(-1), nrep)
    src = Source(r_in, [(source_header(C, S, 2), data, gulp_bytes)], wait_readers=2)
    ths = [threading.Thread(target=b.main, daemon=True) for b in (corr, cacc, bf, sb)]
    for t in [fast, slow, beams, power] + ths:
        t.start()
    t0 = time.time()
    while len(r_in._readers) < 2 and time.time() - t0 < 20:
        time.sleep(0.01)
    src.start()
    for t in [src] + ths + [fast, slow, beams, power]:
        t.join(180)
        assert not t.is_alive(), "pipeline thread did not finish: %r" % (t,)
    assert fast.verdicts == [True] * nrep, fast.verdicts
    assert slow.verdicts == [True] * (nrep // 2), slow.verdicts
    # the long accumulation was done group by group, each group's spans summed in one pass (the block's default), beside everything
    # else -- or, when the suite runs under XENG_CORRACC=fused (and the fused contraction kernel is in use), by the dumps themselves
    if os.environ.get("XENG_CORRACC") == "fused" and os.environ.get("XENG_RAW") != "0":
        assert corr.stats['fused_corracc'] is True and cacc.stats['fused'] is True and cacc.fused_dumps == nrep
    elif os.environ.get("XENG_CORRACC") == "map":
        assert corr.stats['fused_corracc'] is False and cacc.stats['grouped'] is False and cacc.fused_dumps == 0
    else:
        assert corr.stats['fused_corracc'] is False and cacc.stats['grouped'] is True and cacc.fused_dumps == 0

    # the beamformer calls alone, same weights, same gulps
    ffi.call("xengBeamformInitialize", 0, ninput, C, g, nbeam, 0)
    dw = ffi.DeviceBuffer(W.nbytes).upload(bf.gains_cpu)
    din = ffi.DeviceBuffer(G * gulp_bytes).upload(vin.reshape(-1))
    dbeam, dpow = ffi.DeviceBuffer(g * C * nbeam * 8), ffi.DeviceBuffer((nbeam // 2) * (g // ns) * C * 16)
    refs = []
    for k in range(G):
        ffi.call("xengBeamformRun", din.ptr + k * gulp_bytes, dbeam.ptr, dw.ptr)
        ffi.call("xengBeamformIntegrate", dbeam.ptr, dpow.ptr, ns)
        ffi.call("xengBeamformSync")
        refs.append((dbeam.download(np.uint8), dpow.download(np.uint8)))
    ffi.call("xengBeamformDestroy")
    (_, _, bspans), = beams.sequences
    (_, _, pspans), = power.sequences
    assert len(bspans) == nrep * G and len(pspans) == nrep * G
    for k in range(nrep * G):
        assert np.array_equal(bspans[k].reshape(-1), refs[k % G][0]), "voltage beams of gulp %d differ from the stand-alone call" % k
        assert np.array_equal(pspans[k].reshape(-1), refs[k % G][1]), "power beams of gulp %d differ from the stand-alone call" % k
