"""Packet slabs as gulps (round 4, SURVEY 8(f) row 3 + the north star's "throughput on synthetic F-engine packets"): through the C ABI.

`xengXgpuKernelAsyncSlab` takes a gulp as the slab of SNAP2 packets it arrived in (format pinned by the reference's
transmitters: test_tx_vectors.py:38-48,103-108).  A regular slab is read by the contraction where it lies; anything else goes
through a scatter into the library's staging area.  Either way the visibilities are, bit for bit, those of the oracle's
unpack (`snap2_unpack`: missing samples read as zero, foreign / out-of-window packets dropped) followed by its correlation."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util
    assert gpu_util.ffi.device_count() >= 1
    return gpu_util


SEQ0, CHAN0 = 10 ** 12 + 7, 1000


def _need_fused(ffi):
    fused, fp6 = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(fused), ctypes.byref(fp6))
    if fused.value != 1:
        with pytest.raises(ffi.XengError):          # (the two-pass X-engine refuses slabs)
            ffi.call("xengXgpuKernelAsyncSlab", 1 << 20, 1, 64, 0, 0, 1 << 21, 0, None, 0)
        ffi.call("xengXgpuDestroy")
        pytest.skip("packet slabs need the fused contraction kernel")


def _slab(pkts):
    stride = len(pkts[0])
    assert all(len(p) == stride for p in pkts)
    return np.frombuffer(b"".join(pkts), dtype=np.uint8), stride


def _expected(pkt_lists, ntime, nchan, nstand):
    acc = None
    for g, pk in enumerate(pkt_lists):
        gulp, _, _ = orc.snap2_unpack(pk, SEQ0 + g * ntime, ntime, CHAN0, nchan, nstand * 2)
        acc = orc.xgpu_correlate(gulp.reshape(ntime, nchan, nstand, 2), nstand, nchan, acc)
    return acc


def _run(gpu, pkt_lists, nstand, nchan, ntime, acc_mode=0):
    """one integration of len(pkt_lists) slabs; returns (visibilities, accumulator or None, fallbacks)"""
    ffi = gpu.ffi
    ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime, len(pkt_lists))
    ffi.call("xengXgpuInitialize", 0)
    fused, fp6 = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(fused), ctypes.byref(fp6))
    if fused.value != 1:        # (XENG_RAW=0: the two-pass X-engine has no descriptor kernel -- the call says UNSUPPORTED, unpack instead)
        ffi.call("xengXgpuDestroy")
        pytest.skip("packet slabs need the fused contraction kernel")
    matlen = orc.per_chan(nstand) * nchan
    out = ffi.DeviceBuffer(matlen * 8)
    acc = ffi.DeviceBuffer(matlen * 8) if acc_mode else None
    ffi.call("xengMemset", out.ptr, 0x5A, out.nbytes)
    if acc:
        ffi.call("xengMemset", acc.ptr, 0, acc.nbytes)
    bufs = []
    for g, pk in enumerate(pkt_lists):
        raw, stride = _slab(pk)
        d = ffi.DeviceBuffer(raw.size).upload(raw)
        bufs.append(d)
        ffi.call("xengXgpuKernelAsyncSlab", d.ptr, len(pk), stride, SEQ0 + g * ntime, CHAN0, out.ptr, int(g == len(pkt_lists) - 1),
                 acc.ptr if acc else None, acc_mode)
    ffi.call("xengXgpuSync")
    nfb = ctypes.c_int(-1)
    ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(nfb))
    vis = out.download(np.int32)
    accv = acc.download(np.int32) if acc else None
    ffi.call("xengXgpuDestroy")
    for b in bufs + [out] + ([acc] if acc else []):
        b.free()
    return vis, accv, nfb.value


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp", [(352, 96, 480, 5), (96, 8, 96, 2), (64, 4, 192, 3), (32, 16, 96, 1)])
def test_regular_slabs_are_read_in_place(gpu, nstand, nchan, ntime, ngulp):
    vin = gpu.synth_voltages(ngulp * ntime, nchan, nstand, "full", seed=nstand + ntime)
    pk = [orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
          for g in range(ngulp)]
    vis, acc, nfb = _run(gpu, pk, nstand, nchan, ntime, acc_mode=1)
    want = orc.xgpu_correlate(vin, nstand, nchan)
    assert nfb == 0, "a regular slab took the scatter path"
    assert np.array_equal(vis, want) and np.array_equal(acc, want)


def test_irregular_slabs_take_the_scatter_and_give_the_unpacked_result(gpu):
    """lost, reordered, duplicated, foreign and out-of-window packets, one kind per gulp, beside a regular gulp in the same
    integration: every irregular gulp is counted, and the integration equals unpack + correlate of what was received"""
    nstand, nchan, ntime = 96, 8, 96
    rng = np.random.default_rng(5)
    vin = gpu.synth_voltages(6 * ntime, nchan, nstand, "full", seed=11)
    mk = lambda g, **kw: orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32,
                                           chan0_pipeline=CHAN0, **kw)
    regular = mk(0)
    shuffled = [mk(1)[i] for i in rng.permutation(len(mk(1)))]
    lost = mk(2)
    lost = lost[:17] + lost[18:40] + lost[41:] + [lost[5], lost[5]]                       # two packets lost; the slab padded with a duplicate
    dup = mk(3)
    dup[7] = dup[8]                                                                      # one lost, one duplicated: count stays regular
    foreign = mk(4)
    foreign[3] = orc.snap2_packets(vin[:1], seq0=SEQ0 + 4 * ntime + 2, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0 + 500)[0]   # another pipeline's channels
    late = mk(5)
    late[10] = orc.snap2_packets(vin[:1], seq0=SEQ0 + 99 * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)[0]              # outside the window
    lists = [regular, shuffled, lost, dup, foreign, late]
    vis, _, nfb = _run(gpu, lists, nstand, nchan, ntime)
    assert nfb == 5
    assert np.array_equal(vis, _expected(lists, ntime, nchan, nstand))


def test_other_packet_geometries_take_the_scatter(gpu):
    """two channel blocks per sample, or 16 stands per packet: complete and in order, but not the layout the contraction
    reads in place -- scattered, same visibilities"""
    nstand, nchan, ntime = 64, 8, 96
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=2)
    want = orc.xgpu_correlate(vin, nstand, nchan)
    for kw in (dict(nchan_blocks=2, nstand_per_pkt=32), dict(nchan_blocks=1, nstand_per_pkt=16)):
        pk = orc.snap2_packets(vin, seq0=SEQ0, sync_time=3, chan0_pipeline=CHAN0, **kw)
        vis, _, nfb = _run(gpu, [pk], nstand, nchan, ntime)
        assert nfb == 1 and np.array_equal(vis, want), kw


def test_slabs_and_plain_gulps_do_not_mix_inside_an_integration(gpu):
    ffi = gpu.ffi
    nstand, nchan, ntime = 64, 4, 96
    vin = gpu.synth_voltages(2 * ntime, nchan, nstand, "full", seed=3)
    raw, stride = _slab(orc.snap2_packets(vin[:ntime], seq0=SEQ0, sync_time=0, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0))
    ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime, 4)
    ffi.call("xengXgpuInitialize", 0)
    _need_fused(ffi)
    dslab, dgulp = ffi.DeviceBuffer(raw.size).upload(raw), ffi.DeviceBuffer(vin[ntime:].size).upload(vin[ntime:])
    out = ffi.DeviceBuffer(orc.per_chan(nstand) * nchan * 8)
    ffi.call("xengXgpuKernelAsyncSlab", dslab.ptr, raw.size // stride, stride, SEQ0, CHAN0, out.ptr, 0, None, 0)
    with pytest.raises(ffi.XengError):
        ffi.call("xengXgpuKernelAsync", dgulp.ptr, out.ptr, 1)
    ffi.call("xengXgpuReset")
    ffi.call("xengXgpuKernelAsync", dgulp.ptr, out.ptr, 0)
    with pytest.raises(ffi.XengError):
        ffi.call("xengXgpuKernelAsyncSlab", dslab.ptr, raw.size // stride, stride, SEQ0, CHAN0, out.ptr, 1, None, 0)
    ffi.call("xengXgpuReset")
    # ... and after the reset a whole integration of either kind is fine
    ffi.call("xengXgpuKernelAsync", dgulp.ptr, out.ptr, 1)
    ffi.call("xengXgpuSync")
    assert np.array_equal(out.download(np.int32), orc.xgpu_correlate(vin[ntime:], nstand, nchan))
    ffi.call("xengXgpuDestroy")
    for b in (dslab, dgulp, out):
        b.free()


def test_streaming_slabs_with_alternating_outputs(gpu):
    """the streaming pattern of bench.py (enqueue integration n, wait for n-1) on slabs, regular and irregular alternating:
    the descriptors and scratch gulps of one staging area are not reused while the launch that reads them is in flight"""
    ffi = gpu.ffi
    nstand, nchan, ntime, ng = 96, 8, 96, 2
    ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime, ng)
    ffi.call("xengXgpuInitialize", 0)
    _need_fused(ffi)
    matlen = orc.per_chan(nstand) * nchan
    outs = [ffi.DeviceBuffer(matlen * 8) for _ in range(2)]
    rng = np.random.default_rng(9)
    slabs, wants = [], []
    for n in range(6):
        vin = gpu.synth_voltages(ng * ntime, nchan, nstand, "full", seed=100 + n)
        lists = []
        for g in range(ng):
            pk = orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=0, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
            if (n + g) % 3 == 1:
                pk = [pk[i] for i in rng.permutation(len(pk))]
            lists.append(pk)
        wants.append(_expected(lists, ntime, nchan, nstand))
        slabs.append([(ffi.DeviceBuffer(_slab(pk)[0].size).upload(_slab(pk)[0]), len(pk), _slab(pk)[1]) for pk in lists])
    got = []
    for n in range(6):
        for g, (d, npk, stride) in enumerate(slabs[n]):
            ffi.call("xengXgpuKernelAsyncSlab", d.ptr, npk, stride, SEQ0 + g * ntime, CHAN0, outs[n & 1].ptr, int(g == ng - 1), None, 0)
        ffi.call("xengXgpuSyncLag", 1)
        if n >= 1:
            got.append(outs[(n - 1) & 1].download(np.int32))
    ffi.call("xengXgpuSync")
    got.append(outs[5 & 1].download(np.int32))
    for n in range(6):
        assert np.array_equal(got[n], wants[n]), n
    ffi.call("xengXgpuDestroy")
    for s in slabs:
        for d, _, _ in s:
            d.free()
    for o in outs:
        o.free()


# ---- the beamformer on packet slabs (xengBeamformRunSlabs) ----

def _beam_init(ffi, mode, ninput, nchan, ntime, nbeam, ntime_blocks=0):
    import os
    old = os.environ.get("XENG_BEAM")
    if mode:
        os.environ["XENG_BEAM"] = mode
    try:
        ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, ntime_blocks)
    finally:
        if mode:
            os.environ.pop("XENG_BEAM")
            if old is not None:
                os.environ["XENG_BEAM"] = old


def _beam_weights(rng, nchan, nbeam, ninput):
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    w[:, :, 3] *= 4096.0                                                       # an outlier input on every row
    w[0] *= np.exp(rng.uniform(-12, 12, (nbeam, ninput))).astype(np.float32)    # channel 0: routed to the bf16x3 kernel
    return w


@pytest.mark.parametrize("mode", ["", "bf16x3", "f32"])
@pytest.mark.parametrize("nstand,nchan,ntime,nbeam,ntime0", [(352, 96, 960, 32, 480), (96, 8, 256, 32, 64), (64, 5, 192, 6, 0)])
def test_beamformer_reads_regular_slabs_in_place(gpu, mode, nstand, nchan, ntime, nbeam, ntime0):
    """one or two regular slabs as a beamformer gulp: no part takes the scatter, and the beams are BIT-IDENTICAL to
    xengBeamformRun on the unpacked gulp (the same kernels do the arithmetic; only the addresses differ) -- on all three
    kernel routes, with outlier inputs and routed tiles in play"""
    ffi = gpu.ffi
    ninput = nstand * 2
    rng = np.random.default_rng(nstand + ntime0)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=ntime + nchan)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    _beam_init(ffi, mode, ninput, nchan, ntime, nbeam)
    mk = lambda lo, hi: _slab(orc.snap2_packets(vin[lo:hi], seq0=SEQ0 + lo, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0))
    parts = [mk(0, ntime0), mk(ntime0, ntime)] if ntime0 else [mk(0, ntime)]
    stride = parts[0][1]
    bufs = [ffi.DeviceBuffer(raw.size).upload(raw) for raw, _ in parts]
    dfull = ffi.DeviceBuffer(vin.size).upload(vin.reshape(-1))
    dw = ffi.DeviceBuffer(w.nbytes).upload(w)
    o1, o2 = ffi.DeviceBuffer(nchan * nbeam * ntime * 8), ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
    ffi.call("xengMemset", o2.ptr, 0x5A, o2.nbytes)
    ffi.call("xengBeamformRunVersioned", dfull.ptr, o1.ptr, dw.ptr, 1)
    if ntime0:
        ffi.call("xengBeamformRunSlabs", bufs[0].ptr, parts[0][0].size // stride, ntime0, bufs[1].ptr, parts[1][0].size // stride, stride, SEQ0, CHAN0,
                 o2.ptr, dw.ptr, 1)
    else:
        ffi.call("xengBeamformRunSlabs", bufs[0].ptr, parts[0][0].size // stride, ntime, None, 0, stride, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
    ffi.call("xengBeamformSync")
    nfb = ctypes.c_int(-1)
    ffi.call("xengBeamformGetSlabFallbacks", ctypes.byref(nfb))
    assert nfb.value == 0
    assert np.array_equal(o1.download(np.uint32), o2.download(np.uint32))
    ffi.call("xengBeamformDestroy")
    for d in bufs + [dfull, dw, o1, o2]:
        d.free()


@pytest.mark.parametrize("mode", ["", "f32"])
def test_beamformer_irregular_slabs_give_the_unpacked_result(gpu, mode):
    """a regular first part and a second part with lost, reordered and foreign packets; then both parts irregular: the beams
    equal xengBeamformRun on what snap2_unpack makes of the packets (missing samples zero), bit for bit"""
    ffi = gpu.ffi
    nstand, nchan, ntime, nbeam, ntime0 = 96, 8, 256, 32, 128
    ninput = nstand * 2
    rng = np.random.default_rng(77)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=5)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    _beam_init(ffi, mode, ninput, nchan, ntime, nbeam)
    mk = lambda lo, hi: orc.snap2_packets(vin[lo:hi], seq0=SEQ0 + lo, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
    p0, p1 = mk(0, ntime0), mk(ntime0, ntime)
    bad1 = [p1[i] for i in rng.permutation(len(p1))]
    bad1 = bad1[:50] + bad1[53:]                                                   # three lost
    bad1[7] = orc.snap2_packets(vin[:1], seq0=SEQ0 + 5000, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)[0]   # outside the window
    bad1 += [bad1[0]] * 3                                                          # (same count again: duplicates)
    bad0 = list(p0)
    bad0[11] = bad0[12]
    dw = ffi.DeviceBuffer(w.nbytes).upload(w)
    o1, o2 = ffi.DeviceBuffer(nchan * nbeam * ntime * 8), ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
    for first, second, nexp in ((p0, bad1, 1), (bad0, bad1, 2), (p0, p1, 0)):       # (the last: the scratch gulp of earlier calls is not read)
        g0, _, _ = orc.snap2_unpack(first, SEQ0, ntime0, CHAN0, nchan, ninput)
        g1, _, _ = orc.snap2_unpack(second, SEQ0 + ntime0, ntime - ntime0, CHAN0, nchan, ninput)
        unpacked = np.concatenate([g0, g1])
        dfull = ffi.DeviceBuffer(unpacked.size).upload(unpacked.reshape(-1))
        (r0, stride), (r1, _) = _slab(first), _slab(second)
        d0, d1 = ffi.DeviceBuffer(r0.size).upload(r0), ffi.DeviceBuffer(r1.size).upload(r1)
        ffi.call("xengBeamformRunVersioned", dfull.ptr, o1.ptr, dw.ptr, 1)
        ffi.call("xengBeamformRunSlabs", d0.ptr, r0.size // stride, ntime0, d1.ptr, r1.size // stride, stride, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
        ffi.call("xengBeamformSync")
        nfb = ctypes.c_int(-1)
        ffi.call("xengBeamformGetSlabFallbacks", ctypes.byref(nfb))
        assert nfb.value == nexp
        assert np.array_equal(o1.download(np.uint32), o2.download(np.uint32)), nexp
        for d in (dfull, d0, d1):
            d.free()
    with pytest.raises(ffi.XengError):       # parts must sit on 16-sample boundaries
        ffi.call("xengBeamformRunSlabs", o1.ptr, 10, 100, o1.ptr, 10, 6176, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
    ffi.call("xengBeamformDestroy")
    for d in (dw, o1, o2):
        d.free()


def test_beamformer_slabs_in_the_integrated_power_mode(gpu):
    """ntime_blocks > 0 (power sums formed in the kernel's epilogue): slabs give the same words as the unpacked gulp"""
    ffi = gpu.ffi
    nstand, nchan, ntime, nbeam, ntime0, nblocks = 96, 8, 256, 32, 128, 8
    ninput = nstand * 2
    rng = np.random.default_rng(3)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=9)
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    _beam_init(ffi, "", ninput, nchan, ntime, nbeam, nblocks)
    mk = lambda lo, hi: _slab(orc.snap2_packets(vin[lo:hi], seq0=SEQ0 + lo, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0))
    (r0, stride), (r1, _) = mk(0, ntime0), mk(ntime0, ntime)
    d0, d1 = ffi.DeviceBuffer(r0.size).upload(r0), ffi.DeviceBuffer(r1.size).upload(r1)
    dfull = ffi.DeviceBuffer(vin.size).upload(vin.reshape(-1))
    dw = ffi.DeviceBuffer(w.nbytes).upload(w)
    nout = (nbeam // 2) * nblocks * nchan * 4 * 4
    o1, o2 = ffi.DeviceBuffer(nout), ffi.DeviceBuffer(nout)
    for _ in range(2):        # (the second pass: the routing answer is known, the fused epilogue forms the sums)
        ffi.call("xengBeamformRunVersioned", dfull.ptr, o1.ptr, dw.ptr, 1)
        ffi.call("xengBeamformRunSlabs", d0.ptr, r0.size // stride, ntime0, d1.ptr, r1.size // stride, stride, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
        ffi.call("xengBeamformSync")
        a, b = o1.download(np.float32), o2.download(np.float32)
        assert np.allclose(a, b, rtol=1e-6, atol=0) and np.abs(a).max() > 0      # (sums added atomically across work-groups: order may differ in the last bit)
    ffi.call("xengBeamformDestroy")
    for d in (d0, d1, dfull, dw, o1, o2):
        d.free()
