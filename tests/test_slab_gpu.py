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
    assert fused.value == 1
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
