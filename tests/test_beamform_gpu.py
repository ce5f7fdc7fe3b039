"""GPU parity tests of CorrAcc's map kernels and the beamformer path, through the C ABI.
Integer work is bit-exact; the fp32 beamformer is held to north_star's tolerance:
max |err| <= 1e-5 of the output RMS against the float64-accumulating oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402

BEAM_RTOL = 1e-5   # BASELINE.json north_star: "beamformer fp32 within 1e-5 rel"


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util
    assert gpu_util.ffi.device_count() >= 1
    return gpu_util


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 4 * 1024 * 1024 + 3])
def test_map_assign_and_add(gpu, n):
    """corr_acc_block.py:304,306: 'a = b' then 'a += b' (int32, wraps)."""
    rng = np.random.default_rng(n)
    a = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    b = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    da = gpu.ffi.DeviceBuffer(a.nbytes + 16).upload(a)
    db = gpu.ffi.DeviceBuffer(b.nbytes + 16).upload(b)
    gpu.ffi.call("xengMapAddI32", da.ptr, db.ptr, n)
    gpu.ffi.call("xengStreamSynchronize")
    exp = orc.map_i32(a.copy(), b, add=True)
    assert np.array_equal(da.download(np.int32, n), exp)
    gpu.ffi.call("xengMapAssignI32", da.ptr, db.ptr, n)
    gpu.ffi.call("xengStreamSynchronize")
    assert np.array_equal(da.download(np.int32, n), b)


@pytest.mark.parametrize("n,nsrc", [(1, 1), (3, 2), (1000, 5), (4 * 1024 * 1024 + 3, 10), (1 << 20, 16), (47849472, 10)])
def test_map_sum_of_a_group_of_dumps(gpu, n, nsrc):
    """xengMapSumI32 (round 5): a = (add ? a : 0) + b_0 + ... + b_{K-1} in one pass == K applications of the reference's
    'a = b' / 'a += b' (corr_acc_block.py:298-306; int32 wraps, so the grouping does not matter).  Last case: config-2 planes."""
    import ctypes
    rng = np.random.default_rng(n + nsrc)
    a0 = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    da = gpu.ffi.DeviceBuffer(4 * n + 16).upload(a0)
    bs, dbs = [], []
    for j in range(nsrc):
        b = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
        bs.append(b)
        dbs.append(gpu.ffi.DeviceBuffer(4 * n + 16).upload(b))
    srcs = (ctypes.c_void_p * nsrc)(*[d.ptr for d in dbs])
    exp = a0.copy()
    for b in bs:
        exp = orc.map_i32(exp, b, add=True)
    gpu.ffi.call("xengMapSumI32", da.ptr, srcs, nsrc, n, 1)
    gpu.ffi.call("xengMapSync")
    assert np.array_equal(da.download(np.int32, n), exp)
    exp = bs[0].copy()
    for b in bs[1:]:
        exp = orc.map_i32(exp, b, add=True)
    gpu.ffi.call("xengMapSumI32", da.ptr, srcs, nsrc, n, 0)
    gpu.ffi.call("xengMapSync")
    assert np.array_equal(da.download(np.int32, n), exp)
    for d in dbs + [da]:
        d.free()


def test_map_sum_rejects_bad_arguments(gpu):
    import ctypes
    d = gpu.ffi.DeviceBuffer(64)
    srcs = (ctypes.c_void_p * 17)(*[d.ptr] * 17)
    L = gpu.ffi.lib()
    assert L.xengMapSumI32(d.ptr, srcs, 17, 4, 0) != 0          # more than XENG_MAP_SUM_MAX sources
    assert L.xengMapSumI32(d.ptr, srcs, 0, 4, 0) != 0
    assert L.xengMapSumI32(None, srcs, 1, 4, 0) != 0
    bad = (ctypes.c_void_p * 1)(d.ptr + 4)
    assert L.xengMapSumI32(d.ptr, bad, 1, 4, 0) != 0             # misaligned source
    d.free()


def run_beamform(gpu, vin, w, ntime, nchan, ninput, nbeam):
    gpu.ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, 0)
    di = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    dw = gpu.ffi.DeviceBuffer(w.nbytes).upload(np.ascontiguousarray(w, dtype=np.complex64))
    do = gpu.ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
    gpu.ffi.call("xengBeamformRun", di.ptr, do.ptr, dw.ptr)
    gpu.ffi.call("xengBeamformSync")
    out = do.download(np.complex64).reshape(nchan, nbeam, ntime)
    return out, do


def check_beams(got, exp):
    scale = np.sqrt(np.mean(np.abs(exp.astype(np.complex128)) ** 2))
    err = np.max(np.abs(got.astype(np.complex128) - exp.astype(np.complex128))) / scale
    assert err <= BEAM_RTOL, err
    return err


@pytest.mark.parametrize("tag", ["small", "tile"])
def test_beamform_golden(gpu, golden_dir, tag):
    """Against outputs of the reference's own SoftwareBf.cpu_beamform / cpu_sum_power (which accumulate
    in complex64 -- the reference's tolerance rtol=atol=1e-4 applies, beamformer_test.py:109) and
    against the float64 oracle at 1e-5."""
    z = np.load(os.path.join(golden_dir, "beamform_%s.npz" % tag))
    vin, w, beams, power = z["vin"], z["weights"], z["beams"], z["power"]
    ntime, nchan, ninput = vin.shape
    nbeam = w.shape[1]
    got, dout = run_beamform(gpu, vin, w, ntime, nchan, ninput, nbeam)
    scale = np.sqrt(np.mean(np.abs(beams) ** 2))
    assert np.all(np.isclose(got, beams, rtol=1e-4, atol=1e-4 * scale))
    check_beams(got, orc.beamform(vin, w, ntime, nchan, ninput, nbeam))
    ns = int(z["ntime_sum"])
    dp = gpu.ffi.DeviceBuffer(power.nbytes)
    gpu.ffi.call("xengBeamformIntegrate", dout.ptr, dp.ptr, ns)
    gpu.ffi.call("xengBeamformSync")
    gp = dp.download(np.float32).reshape(power.shape)
    exp = orc.beamform_integrate(got, ns)          # same input the GPU integrated
    assert np.all(np.isclose(gp, exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max()))
    assert np.all(np.isclose(gp, power, rtol=1e-4, atol=1e-4 * np.abs(power).max()))
    ds = gpu.ffi.DeviceBuffer(power[0].nbytes)
    for b in range(nbeam // 2):
        gpu.ffi.call("xengBeamformIntegrateSingleBeam", dout.ptr, ds.ptr, ns, b)
        gpu.ffi.call("xengBeamformSync")
        assert np.array_equal(ds.download(np.float32).reshape(power[0].shape), gp[b])
    gpu.ffi.call("xengBeamformDestroy")


def block_weights(nchan, nbeam, ninput, sfreq=50e6, chan_bw=23925.78125, seed=0xaabbccdd):
    """Weights as the Beamform block builds them (beamform_block.py:343-350) from the test's
    random delays / amps / cal gains (beamformer_test.py:131-139)."""
    rng = np.random.default_rng(seed)
    freqs = sfreq + np.arange(nchan) * chan_bw
    w = np.zeros((nchan, nbeam, ninput), np.complex64)
    for b in range(nbeam):
        delays_ns = rng.uniform(0, 12, ninput)
        amps = rng.uniform(10, 17, ninput)
        cal = (rng.uniform(-1, 1, (nchan, ninput)) + 1j * rng.uniform(-1, 1, (nchan, ninput))).astype(np.complex64)
        w[:, b, :] = amps * np.exp(1j * 2 * np.pi * freqs[:, None] * delays_ns * 1e-9) * cal
    return w


@pytest.mark.parametrize("ntime,nchan,ninput,nbeam", [
    (32, 2, 64, 32),
    (100, 3, 40, 5),        # ragged: time not /32, inputs not /64, beams not /32
    (960, 4, 704, 32),      # config 4 shapes at 4 channels
    (130, 1, 704, 34),      # more than one beam tile
])
def test_beamform_vs_oracle(gpu, ntime, nchan, ninput, nbeam):
    rng = np.random.default_rng(ntime + ninput)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput)
    got, _ = run_beamform(gpu, vin, w, ntime, nchan, ninput, nbeam)
    check_beams(got, orc.beamform(vin, w, ntime, nchan, ninput, nbeam))
    gpu.ffi.call("xengBeamformDestroy")


def test_config4_full_size(gpu):
    """BASELINE config 4: 704 inputs, 96 chan, 32 beams, 960 samples, then 16 dual-pol power beams
    (ntime_sum 24).  Full-size checks: the float64 oracle on all 96 channels + linearity in the weights."""
    ntime, nchan, ninput, nbeam, ns = 960, 96, 704, 32, 24
    rng = np.random.default_rng(44)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput)
    got, dout = run_beamform(gpu, vin, w, ntime, nchan, ninput, nbeam)
    check_beams(got, orc.beamform(vin, w, ntime, nchan, ninput, nbeam))      # every channel, float64 oracle
    # linearity: beams(2w) == 2*beams(w) exactly in fp32 (power-of-two scaling)
    got2, _ = run_beamform(gpu, vin, (2 * w).astype(np.complex64), ntime, nchan, ninput, nbeam)
    assert np.array_equal(got2, 2 * got)
    dp = gpu.ffi.DeviceBuffer((nbeam // 2) * (ntime // ns) * nchan * 16)
    gpu.ffi.call("xengBeamformIntegrate", dout.ptr, dp.ptr, ns)   # dout holds beams(2w) now? no: separate buffers
    gpu.ffi.call("xengBeamformSync")
    gp = dp.download(np.float32).reshape(nbeam // 2, ntime // ns, nchan, 4)
    exp = orc.beamform_integrate(got, ns)
    assert np.all(np.isclose(gp, exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max()))
    gpu.ffi.call("xengBeamformDestroy")


def test_integrated_mode(gpu):
    """ntime_blocks > 0 ('experimental' in the reference, beamform_block.py:108-110; parity unpinned):
    Run == Run(voltage) followed by Integrate over ntime/ntime_blocks samples."""
    ntime, nchan, ninput, nbeam, nblk = 96, 2, 64, 4, 4
    rng = np.random.default_rng(3)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput)
    gpu.ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, nblk)
    di = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    dw = gpu.ffi.DeviceBuffer(w.nbytes).upload(w)
    do = gpu.ffi.DeviceBuffer((nbeam // 2) * nblk * nchan * 16)
    gpu.ffi.call("xengBeamformRun", di.ptr, do.ptr, dw.ptr)
    gpu.ffi.call("xengBeamformSync")
    got = do.download(np.float32).reshape(nbeam // 2, nblk, nchan, 4)
    exp = orc.beamform_integrate(orc.beamform(vin, w, ntime, nchan, ninput, nbeam), ntime // nblk)
    assert np.all(np.isclose(got, exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max()))
    gpu.ffi.call("xengBeamformDestroy")


@pytest.mark.parametrize("ntime,nchan,ninput,nbeam,nblk", [
    (960, 4, 704, 32, 40),      # config-4 shapes: 24-sample blocks straddle the 128-sample work-group tiles
    (384, 2, 64, 6, 3),         # 128-sample blocks = one work-group each; 3 beam pairs
    (200, 3, 48, 4, 8),         # ragged last work-group (200 = 128 + 72), 25-sample blocks
])
def test_integrated_mode_fused_epilogue(gpu, ntime, nchan, ninput, nbeam, nblk):
    """ntime_blocks > 0: the power sums are formed in the beamformer kernel's epilogue (no voltage beams in memory, no
    Integrate launch) whenever the weights' routing allows it -- from the FIRST call on: the call that uploads new weights
    waits for the routing answer instead of composing Run + Integrate "until the answer is in" (the two paths sum in
    different orders, so which one ran would show in the last bits and depend on timing).  Against the oracle, and
    bit-identical between all calls."""
    import ctypes
    rng = np.random.default_rng(ntime)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput)
    gpu.ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, nblk)
    di = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    dw = gpu.ffi.DeviceBuffer(w.nbytes).upload(w)
    do = gpu.ffi.DeviceBuffer((nbeam // 2) * nblk * nchan * 16)
    exp = orc.beamform_integrate(orc.beamform(vin, w, ntime, nchan, ninput, nbeam), ntime // nblk)
    tm, cn = (ctypes.c_double * 2)(), (ctypes.c_int * 2)()
    gpu.ffi.call("xengBeamformSetProfiling", 1)
    gpu.ffi.call("xengBeamformGetTimes", tm, cn)
    outs = []
    for k in range(3):
        gpu.ffi.call("xengMemset", do.ptr, 0x7F, do.nbytes)
        gpu.ffi.call("xengBeamformRunVersioned", di.ptr, do.ptr, dw.ptr, 9)
        gpu.ffi.call("xengBeamformSync")
        outs.append(do.download(np.float32).reshape(nbeam // 2, nblk, nchan, 4))
        assert np.all(np.isclose(outs[-1], exp, rtol=1e-5, atol=1e-5 * np.abs(exp).max())), k
    gpu.ffi.call("xengBeamformGetTimes", tm, cn)
    gpu.ffi.call("xengBeamformSetProfiling", 0)
    if os.environ.get("XENG_BEAM", "int8x3") == "int8x3" and not os.environ.get("XENG_BEAM_F32"):
        assert cn[0] == 3 and cn[1] == 0                 # three fused calls, Integrate never launched
    else:
        assert cn[0] == 3 and cn[1] == 3                 # (the other kernels have no fused epilogue: Run -> Integrate)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    gpu.ffi.call("xengBeamformDestroy")


def test_versioned_weights_are_resplit_only_on_change(gpu):
    """xengBeamformRunVersioned: same (pointer, version) reuses the bf16-split weights; a new version
    (or version 0) re-splits, so results always follow the current weights."""
    ntime, nchan, ninput, nbeam = 128, 2, 64, 32
    rng = np.random.default_rng(9)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w1, w2 = block_weights(nchan, nbeam, ninput, seed=1), block_weights(nchan, nbeam, ninput, seed=2)
    gpu.ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, 0)
    di = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    dw = gpu.ffi.DeviceBuffer(w1.nbytes).upload(w1)
    do = gpu.ffi.DeviceBuffer(nchan * nbeam * ntime * 8)

    def run(version):
        gpu.ffi.call("xengBeamformRunVersioned", di.ptr, do.ptr, dw.ptr, version)
        gpu.ffi.call("xengBeamformSync")
        return do.download(np.complex64).reshape(nchan, nbeam, ntime)
    e1 = orc.beamform(vin, w1, ntime, nchan, ninput, nbeam)
    e2 = orc.beamform(vin, w2, ntime, nchan, ninput, nbeam)
    check_beams(run(7), e1)
    dw.upload(w2)
    direct = os.environ.get("XENG_BEAM") == "f32" or bool(os.environ.get("XENG_BEAM_F32"))
    check_beams(run(7), e2 if direct else e1)     # caller said "unchanged": the prepared copy of w1 is still in use
                                                  # (the fp32 kernel reads the caller's weights themselves: nothing is prepared)
    check_beams(run(8), e2)          # new version: re-split
    dw.upload(w1)
    check_beams(run(0), e1)          # version 0: always re-split (the reference call shape)
    gpu.ffi.call("xengBeamformDestroy")


def test_completion_tickets_and_their_query(gpu):
    """xengBeamformMark / Wait / TicketDone: a ticket whose kernels have completed reads done = 1 (and its output is
    there); unknown tickets are errors; the query never blocks (it is what the blocks call before a blocking Wait)."""
    import ctypes
    ntime, nchan, ninput, nbeam = 256, 4, 64, 32
    rng = np.random.default_rng(11)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput, seed=3)
    gpu.ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, 0)
    di = gpu.ffi.DeviceBuffer(vin.size).upload(vin)
    dw = gpu.ffi.DeviceBuffer(w.nbytes).upload(w)
    outs = [gpu.ffi.DeviceBuffer(nchan * nbeam * ntime * 8) for _ in range(6)]
    tickets, done = [], ctypes.c_int(-1)
    for o in outs:
        gpu.ffi.call("xengBeamformRunVersioned", di.ptr, o.ptr, dw.ptr, 1)
        t = ctypes.c_ulonglong()
        gpu.ffi.call("xengBeamformMark", ctypes.byref(t))
        tickets.append(t.value)
    assert tickets == sorted(set(tickets)) and tickets[0] >= 1
    gpu.ffi.call("xengBeamformTicketDone", tickets[-1], ctypes.byref(done))      # returns at once, whatever the answer
    assert done.value in (0, 1)
    gpu.ffi.call("xengBeamformWait", tickets[2])
    for t in tickets[:3]:                                                          # stream order: everything before it too
        gpu.ffi.call("xengBeamformTicketDone", t, ctypes.byref(done))
        assert done.value == 1
    expect = orc.beamform(vin, w, ntime, nchan, ninput, nbeam)
    check_beams(outs[2].download(np.complex64).reshape(nchan, nbeam, ntime), expect)
    gpu.ffi.call("xengBeamformSync")
    gpu.ffi.call("xengBeamformTicketDone", tickets[-1], ctypes.byref(done))
    assert done.value == 1
    check_beams(outs[-1].download(np.complex64).reshape(nchan, nbeam, ntime), expect)
    for bad in (0, tickets[-1] + 1):
        with pytest.raises(gpu.ffi.XengError):
            gpu.ffi.call("xengBeamformTicketDone", bad, ctypes.byref(done))
        with pytest.raises(gpu.ffi.XengError):
            gpu.ffi.call("xengBeamformWait", bad)
    with pytest.raises(gpu.ffi.XengError):
        gpu.ffi.call("xengBeamformTicketDone", tickets[0], None)
    # the backend's beam_wait (query first, blocking call only when needed) through both branches
    from caltech_bifrost_dsp_amd.backend import HipBackend
    be = HipBackend()
    gpu.ffi.call("xengBeamformRunVersioned", di.ptr, outs[0].ptr, dw.ptr, 1)
    tk = be.beam_mark()
    be.beam_wait(tk)
    be.beam_wait(tk)                 # already complete: answered by the query
    check_beams(outs[0].download(np.complex64).reshape(nchan, nbeam, ntime), expect)
    gpu.ffi.call("xengBeamformDestroy")


@pytest.mark.parametrize("ntime,nchan,ninput,nbeam,kind", [
    (32, 2, 64, 32, "block"),
    (100, 3, 48, 5, "block"),          # ragged: time not /32, inputs not /32, beams not /32
    (960, 4, 704, 32, "block"),        # config 4 shapes at 4 channels
    (130, 1, 704, 34, "block"),        # more than one beam tile
    (96, 2, 704, 8, "dynamic"),        # 60 dB of dynamic range inside a beam's weights, some inputs flagged (zero)
    (64, 1, 704, 4, "coherent"),       # every sample -8-8j and weights of one phase: quantisation errors add coherently
    # dominant weights (a huge calibration gain) on DEAD inputs (all-zero voltages): the output is made by the ordinary
    # weights alone, so a row scale taken from the row maximum would cost them their significant bits
    (96, 2, 704, 8, "dead1e2"),
    (96, 2, 704, 8, "dead1e4"),
    (96, 3, 704, 34, "dead1e6"),       # three dead inputs, two beam tiles
    (96, 2, 704, 8, "live1e4"),        # the same on a live input (the output is then dominated by that input)
    (96, 2, 704, 8, "dead_many"),      # 40 dead inputs with x1e3 gains: more than a tile's outlier list holds
    (96, 2, 704, 8, "dead_tail"),      # 12 dominant weights per row: more than a row's outlier list holds
    (96, 3, 704, 34, "dead_mixed"),    # only channel 1 / the second beam tile are affected: per-tile routing
    (96, 1, 64, 4, "sparse"),          # rows with 3 non-zero weights of very different size
])
@pytest.mark.parametrize("mode", ["int8x3", "bf16x3", "f32"])
def test_beamform_kernel_routes(gpu, ntime, nchan, ninput, nbeam, kind, mode):
    """The three Run kernels against the float64 oracle at the same 1e-5-of-RMS bar: int8x3 (default: weights as
    three balanced base-255 int8 digits per (channel, beam) row on the int8 MFMA, exact integer sums, fp32
    recombination), bf16x3 (exact three-term bf16 split, fp32 accumulation) and f32 (fp32 MFMA chain)."""
    rng = np.random.default_rng(ntime + ninput + nbeam)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    w = block_weights(nchan, nbeam, ninput)
    if kind == "dynamic":
        w = (w * (10.0 ** rng.uniform(-3, 0, (nchan, nbeam, ninput)))).astype(np.complex64)
        w[:, :, rng.integers(0, ninput, 40)] = 0
    if kind == "coherent":
        vin[...] = 0x88
        w = (np.abs(w) * np.exp(1j * 0.7)).astype(np.complex64)
    if kind.startswith("dead") or kind == "live1e4":
        ndead = {"dead1e2": 1, "dead1e4": 1, "dead1e6": 3, "live1e4": 1, "dead_many": 40, "dead_tail": 12, "dead_mixed": 40}[kind]
        gain = {"dead1e2": 1e2, "dead1e4": 1e4, "dead1e6": 1e6, "live1e4": 1e4, "dead_many": 1e3, "dead_tail": 3e3, "dead_mixed": 1e3}[kind]
        dead = rng.choice(ninput, ndead, replace=False)
        if kind != "live1e4":
            vin[:, :, dead] = 0
        if kind == "dead_mixed":
            w[1, 32:, dead] *= gain
        else:
            w[:, :, dead] *= gain
    if kind == "sparse":
        w[...] = 0
        for b in range(nbeam):
            w[0, b, rng.choice(ninput, 3, replace=False)] = np.array([1e-3, 1.0 + 2.0j, 3e4j], np.complex64) * (b + 1)
    os.environ["XENG_BEAM"] = mode
    try:
        got, _ = run_beamform(gpu, vin, w, ntime, nchan, ninput, nbeam)
        err = check_beams(got, orc.beamform(vin, w, ntime, nchan, ninput, nbeam))
        if mode == "int8x3":
            # the precision control took the expected route (beamform_kernels.h): a few dominant weights are added in
            # fp32 by the int8x3 kernel itself; too many of them send the (channel, beam tile) to the bf16x3 kernel
            import ctypes
            tot, nbf, nout = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            gpu.ffi.call("xengBeamformGetRouteInfo", ctypes.byref(tot), ctypes.byref(nbf), ctypes.byref(nout))
            nbt = (nbeam + 31) // 32
            assert tot.value == nchan * nbt
            expect = {"block": (0, 0), "coherent": (0, 0), "dead1e2": (0, nchan * nbt), "dead1e4": (0, nchan * nbt),
                      "dead1e6": (0, 3 * nchan * nbt), "live1e4": (0, nchan * nbt), "dead_many": (nchan * nbt, 0),
                      "dead_tail": (nchan * nbt, 0), "dead_mixed": (1, 0)}
            if kind in expect:
                assert (nbf.value, nout.value) == expect[kind], (kind, nbf.value, nout.value)
        got2, _ = run_beamform(gpu, vin, (2 * w).astype(np.complex64), ntime, nchan, ninput, nbeam)
        assert np.array_equal(got2, 2 * got)            # power-of-two scaling is exact in all three
        print("%s %s max err / rms = %.2e" % (mode, kind, err))
    finally:
        del os.environ["XENG_BEAM"]
        gpu.ffi.call("xengBeamformDestroy")


@pytest.mark.parametrize("mode", ["", "bf16x3", "f32"])
@pytest.mark.parametrize("ninput,nchan,ntime,nbeam,ntime0", [(704, 96, 960, 32, 480), (704, 8, 960, 32, 100), (80, 16, 120, 32, 64), (192, 5, 100, 6, 37)])
def test_two_part_gulp_equals_the_contiguous_gulp(gpu, mode, ninput, nchan, ntime, nbeam, ntime0):
    """xengBeamformRunParts (round 4): one beamformer gulp out of two separate spans -- the first ntime0 samples at one
    address, the rest at another -- is BIT-IDENTICAL to xengBeamformRun on the same samples laid out contiguously (same
    kernels, same order of operations; only the row addresses differ), on all three kernel routes, for a split on and off the
    128-sample work-group boundary; heavy-tailed weights exercise the outlier and routed-tile paths."""
    ffi = gpu.ffi
    rng = np.random.default_rng(ninput + ntime0)
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    w[:, :, 3] *= 4096.0
    w[0] *= np.exp(rng.uniform(-12, 12, (nbeam, ninput))).astype(np.float32)
    vin = rng.integers(0, 256, (ntime, nchan, ninput), dtype=np.uint8)
    old = os.environ.get("XENG_BEAM")
    if mode:
        os.environ["XENG_BEAM"] = mode
    try:
        ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, 0)
    finally:
        if mode:
            os.environ.pop("XENG_BEAM")
            if old is not None:
                os.environ["XENG_BEAM"] = old
    row = nchan * ninput
    dfull = ffi.DeviceBuffer(vin.size).upload(vin)
    # the two parts far apart, in the "wrong" order in memory, with junk around them
    dparts = ffi.DeviceBuffer(2 * vin.size + 4096)
    ffi.call("xengMemset", dparts.ptr, 0x77, dparts.nbytes)
    p1_off, p0_off = 1024, vin.size + 2048
    dparts.upload(vin[:ntime0], offset=p0_off)
    dparts.upload(vin[ntime0:], offset=p1_off)
    dw = ffi.DeviceBuffer(w.nbytes).upload(w)
    o1, o2 = ffi.DeviceBuffer(nchan * nbeam * ntime * 8), ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
    ffi.call("xengBeamformRunVersioned", dfull.ptr, o1.ptr, dw.ptr, 1)
    ffi.call("xengBeamformRunParts", dparts.ptr + p0_off, ntime0, dparts.ptr + p1_off, o2.ptr, dw.ptr, 1)
    ffi.call("xengBeamformSync")
    a, b = o1.download(np.uint32), o2.download(np.uint32)
    assert np.array_equal(a, b)
    # (values: against the oracle on two channels with ordinary weight rows -- channel 0's rows span ten decades, which is a
    # routing test case, test_beamform_kernel_routes, not an accuracy one)
    exp = orc.beamform(np.ascontiguousarray(vin[:, 1:3]), np.ascontiguousarray(w[1:3]), ntime, 2, ninput, nbeam)
    check_beams(o2.download(np.complex64).reshape(nchan, nbeam, ntime)[1:3], exp)
    with pytest.raises(ffi.XengError):
        ffi.call("xengBeamformRunParts", dparts.ptr + p0_off, ntime, dparts.ptr + p1_off, o2.ptr, dw.ptr, 1)     # first part = the whole gulp
    assert row * ntime0 % 4 == 0 or True
    ffi.call("xengBeamformDestroy")
    for d in (dfull, dparts, dw, o1, o2):
        d.free()
