"""Packet slabs as gulps (round 4, SURVEY 8(f) row 3 + the north star's "throughput on synthetic F-engine packets"): through the C ABI.

`xengXgpuKernelAsyncSlab` takes a gulp as the slab of SNAP2 packets it arrived in (format pinned by the reference's
transmitters: test_tx_vectors.py:38-48,103-108).  Round 5: every slab of the deployed packet geometry is read by the contraction
where it lies, through a table of packet offsets built on the device -- in order, with lost, shifted, reordered, duplicated,
foreign or out-of-window packets alike; only packets of another geometry send a gulp through a scatter into the library's
staging area.  Either way the visibilities are, bit for bit, those of the oracle's unpack (`snap2_unpack`: missing samples read as
zero, foreign / out-of-window packets dropped, the last packet that carries a sample wins) followed by its correlation."""
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


def _init(ffi, tables, *cfg):
    """Configure + Initialize under XENG_SLAB_TABLES = tables ("1": the contraction follows the offset tables from the first launch
    on; "0": never; None: the shipped default -- by strides until a gulp was not regular)"""
    import os
    old = os.environ.pop("XENG_SLAB_TABLES", None)
    if tables is not None:
        os.environ["XENG_SLAB_TABLES"] = tables
    try:
        ffi.call("xengXgpuConfigure", *cfg)
        ffi.call("xengXgpuInitialize", 0)
    finally:
        os.environ.pop("XENG_SLAB_TABLES", None)
        if old is not None:
            os.environ["XENG_SLAB_TABLES"] = old


def _run(gpu, pkt_lists, nstand, nchan, ntime, acc_mode=0, tables=None):
    """one integration of len(pkt_lists) slabs; returns (visibilities, accumulator or None, gulps scattered); _run.irregular: gulps
    read in place through a table that was not regular"""
    ffi = gpu.ffi
    _init(ffi, tables, nstand, 2, nchan, ntime, len(pkt_lists))
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
    nfb, nirr = ctypes.c_int(-1), ctypes.c_int(-1)
    ffi.call("xengXgpuGetSlabStats", ctypes.byref(nfb), ctypes.byref(nirr))
    again = ctypes.c_int(-1)
    ffi.call("xengXgpuGetSlabFallbacks", ctypes.byref(again))
    assert again.value == 0                     # (reading the counters resets them)
    vis = out.download(np.int32)
    accv = acc.download(np.int32) if acc else None
    ffi.call("xengXgpuDestroy")
    for b in bufs + [out] + ([acc] if acc else []):
        b.free()
    _run.irregular = nirr.value
    return vis, accv, nfb.value


@pytest.mark.parametrize("tables", [None, "1"])
@pytest.mark.parametrize("nstand,nchan,ntime,ngulp", [(352, 96, 480, 5), (96, 8, 96, 2), (64, 4, 192, 3), (32, 16, 96, 1)])
def test_regular_slabs_are_read_in_place(gpu, nstand, nchan, ntime, ngulp, tables):
    vin = gpu.synth_voltages(ngulp * ntime, nchan, nstand, "full", seed=nstand + ntime)
    pk = [orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
          for g in range(ngulp)]
    vis, acc, nfb = _run(gpu, pk, nstand, nchan, ntime, acc_mode=1, tables=tables)
    want = orc.xgpu_correlate(vin, nstand, nchan)
    assert nfb == 0 and _run.irregular == 0, "a regular slab took the scatter path / was counted irregular"
    assert np.array_equal(vis, want) and np.array_equal(acc, want)


@pytest.mark.parametrize("tables", [None, "1"])
def test_irregular_slabs_give_the_unpacked_result(gpu, tables):
    """lost, reordered, duplicated, foreign and out-of-window packets, one kind per gulp, beside a regular gulp in the same
    integration: following the tables no gulp is scattered, every irregular gulp is counted; by strides (a link that has been clean so
    far) the five take the round-4 scatter; the integration equals unpack + correlate of what was received either way"""
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
    vis, _, nfb = _run(gpu, lists, nstand, nchan, ntime, tables=tables)
    assert (nfb, _run.irregular) == ((0, 5) if tables else (5, 0))
    assert np.array_equal(vis, _expected(lists, ntime, nchan, nstand))


def _mk(gpu, vin, g, ntime, **kw):
    return orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0, **kw)


@pytest.mark.parametrize("tables", ["1", None])
@pytest.mark.parametrize("nstand,nchan,ntime", [(96, 8, 96), (352, 96, 480)])
def test_lossy_links_shifted_slabs_and_last_duplicate_wins(gpu, nstand, nchan, ntime, tables):
    """what receivers really leave behind, one kind per gulp (the reference's transmitter has a deliberate-loss switch,
    test_tx_mt.c:22,108-118):
      0  arrival order with 1 % of the packets lost: everything behind a loss sits one slot early, the slab is SHORTER
      1  slots by index, a lost packet's slot keeps a stale packet of an older window
      2  a lost packet's slot holds a copy of the next packet (bench.py's model), two of them adjacent: the first then holds the
         only copy of a packet whose own slot is taken
      3  two packets carry the same (sample, block) with DIFFERENT payloads: the later one wins, as in the oracle's scatter
      4  a whole sample and a whole 64-input block missing
      5  nothing but foreign packets: the gulp reads as zeros
    Following the tables all are read in place, none is scattered; by strides all six take the scatter (two different payloads for one
    sample make that path's result depend on the order of its waves: that gulp is left out there)."""
    rng = np.random.default_rng(nstand)
    vin = gpu.synth_voltages(6 * ntime, nchan, nstand, "full", seed=21)
    nblk = nstand * 2 // 64
    lists = []
    full = _mk(gpu, vin, 0, ntime)
    lost = set(rng.choice(len(full), size=max(2, len(full) // 100), replace=False).tolist())
    lists.append([pk for i, pk in enumerate(full) if i not in lost])
    full = _mk(gpu, vin, 1, ntime)
    stale = orc.snap2_packets(vin[:1], seq0=SEQ0 - 7 * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
    for i in rng.choice(len(full), size=5, replace=False):
        full[i] = stale[i % nblk]
    lists.append(full)
    full = _mk(gpu, vin, 2, ntime)
    orig = list(full)
    for i in (9, 10, 40):
        full[i] = orig[i + 1]
    lists.append(full)
    full = _mk(gpu, vin, 3, ntime)
    other = _mk(gpu, vin[::-1], 3, ntime)                      # same headers, other payloads
    full[5] = other[20]                                       # an early copy of (sample, block) 20 that loses against slot 20 ...
    full[len(full) - 3] = other[33]                           # ... and a late one of 33 that wins
    lists.append(full)
    full = _mk(gpu, vin, 4, ntime)
    lists.append([pk for i, pk in enumerate(full) if i // nblk != 17 and i % nblk != 1])
    lists.append(orc.snap2_packets(vin[:ntime], seq0=SEQ0 + 5 * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0 + 5000))
    if not tables:
        lists[3] = _mk(gpu, vin, 3, ntime)[::-1]              # (reversed instead: the scatter has no order among duplicates)
    vis, acc, nfb = _run(gpu, lists, nstand, nchan, ntime, acc_mode=1, tables=tables)
    want = _expected(lists, ntime, nchan, nstand)
    assert (nfb, _run.irregular) == ((0, 6) if tables else (6, 0))
    assert np.array_equal(vis, want) and np.array_equal(acc, want)


def test_the_contraction_follows_the_tables_on_a_lossy_link(gpu):
    """the shipped default: slabs are addressed by their strides while the link is clean or nearly so -- a gulp that is not regular
    takes the round-4 scatter and is counted in pinned memory --; once more than a quarter of the recent gulps were not regular the
    host's launches read every gulp through its table; when the link is clean again they go back.  Exact at every step."""
    ffi = gpu.ffi
    nstand, nchan, ntime = 96, 8, 96
    vin = gpu.synth_voltages(3 * ntime, nchan, nstand, "full", seed=31)
    full = [_mk(gpu, vin, g, ntime) for g in range(3)]
    lossy = [full[0][:7] + full[0][8:], full[1][:30] + full[1][31:], full[2][1:]]
    want_full, want_lossy = _expected(full, ntime, nchan, nstand), _expected(lossy, ntime, nchan, nstand)
    _init(ffi, None, nstand, 2, nchan, ntime, 3)
    _need_fused(ffi)
    out = ffi.DeviceBuffer(orc.per_chan(nstand) * nchan * 8)
    bufs = {}

    def integrate(lists):
        for g, pk in enumerate(lists):
            raw, stride = _slab(pk)
            d = bufs.setdefault(id(pk), ffi.DeviceBuffer(raw.size).upload(raw))
            ffi.call("xengXgpuKernelAsyncSlab", d.ptr, len(pk), stride, SEQ0 + g * ntime, CHAN0, out.ptr, int(g == 2), None, 0)
        ffi.call("xengXgpuSync")
        ns, ni = ctypes.c_int(-1), ctypes.c_int(-1)
        ffi.call("xengXgpuGetSlabStats", ctypes.byref(ns), ctypes.byref(ni))
        return out.download(np.int32), ns.value, ni.value

    vis, ns, ni = integrate(full)
    assert (ns, ni) == (0, 0) and np.array_equal(vis, want_full)
    seen = []
    for _ in range(4):                                       # every gulp lossy: scattered at first, then read through the tables
        vis, ns, ni = integrate(lossy)
        assert ns + ni == 3 and np.array_equal(vis, want_lossy), (ns, ni)
        seen.append((ns, ni))
    assert seen[0] == (3, 0) and seen[-1] == (0, 3), seen
    for k in range(30):                                      # clean again: regular tables at first, then back to the strides
        vis, ns, ni = integrate(full)
        assert (ns, ni) == (0, 0) and np.array_equal(vis, want_full), k
    vis, ns, ni = integrate(lossy)
    assert (ns, ni) == (3, 0) and np.array_equal(vis, want_lossy), (ns, ni)
    ffi.call("xengXgpuDestroy")
    for b in list(bufs.values()) + [out]:
        b.free()


def test_other_packet_geometries_take_the_scatter(gpu):
    """two channel blocks per sample, or 16 stands per packet: complete and in order, but not the layout the contraction
    reads in place -- scattered, same visibilities"""
    nstand, nchan, ntime = 64, 8, 96
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=2)
    want = orc.xgpu_correlate(vin, nstand, nchan)
    for tables in (None, "1"):
        for kw in (dict(nchan_blocks=2, nstand_per_pkt=32), dict(nchan_blocks=1, nstand_per_pkt=16)):
            pk = orc.snap2_packets(vin, seq0=SEQ0, sync_time=3, chan0_pipeline=CHAN0, **kw)
            vis, _, nfb = _run(gpu, [pk], nstand, nchan, ntime, tables=tables)
            assert nfb == 1 and np.array_equal(vis, want), (kw, tables)


@pytest.mark.parametrize("tables", [None, "1"])
def test_an_integration_longer_than_the_staging_depth_on_lossy_slabs(gpu, tables):
    """five slabs per integration with room for two in the library: the contraction runs after the second and the fourth slab (partial
    sums read-modify-written in `out`) and at the dump -- every launch with its own tables / descriptors, regular and lossy slabs mixed"""
    ffi = gpu.ffi
    nstand, nchan, ntime, ng = 96, 8, 96, 5
    vin = gpu.synth_voltages(ng * ntime, nchan, nstand, "full", seed=41)
    lists = [_mk(gpu, vin, g, ntime) for g in range(ng)]
    lists[1] = lists[1][:11] + lists[1][12:]
    lists[2] = lists[2][::-1]
    lists[4] = lists[4][:50] + lists[4][53:] + [lists[4][0]]
    want = _expected(lists, ntime, nchan, nstand)
    _init(ffi, tables, nstand, 2, nchan, ntime, 2)
    _need_fused(ffi)
    out = ffi.DeviceBuffer(orc.per_chan(nstand) * nchan * 8)
    ffi.call("xengMemset", out.ptr, 0x5A, out.nbytes)
    bufs = []
    for g, pk in enumerate(lists):
        raw, stride = _slab(pk)
        d = ffi.DeviceBuffer(raw.size).upload(raw)
        bufs.append(d)
        ffi.call("xengXgpuKernelAsyncSlab", d.ptr, len(pk), stride, SEQ0 + g * ntime, CHAN0, out.ptr, int(g == ng - 1), None, 0)
    ffi.call("xengXgpuSync")
    ns, ni = ctypes.c_int(-1), ctypes.c_int(-1)
    ffi.call("xengXgpuGetSlabStats", ctypes.byref(ns), ctypes.byref(ni))
    assert ns.value + ni.value == 3 and (ns.value == 0 if tables else ns.value >= 1)
    assert np.array_equal(out.download(np.int32), want)
    ffi.call("xengXgpuDestroy")
    for b in bufs + [out]:
        b.free()


def test_input_counts_that_are_not_whole_64_input_blocks_are_scattered(gpu):
    """80 inputs (16 inputs per packet): the table kernel reads whole 64-input blocks, so every slab is scattered and the plain kernel
    contracts the copies -- same visibilities"""
    nstand, nchan, ntime = 40, 8, 96
    vin = gpu.synth_voltages(2 * ntime, nchan, nstand, "full", seed=4)
    want = orc.xgpu_correlate(vin, nstand, nchan)
    for tables in (None, "1"):
        pk = [orc.snap2_packets(vin[g * ntime:(g + 1) * ntime], seq0=SEQ0 + g * ntime, sync_time=3, nchan_blocks=1, nstand_per_pkt=8, chan0_pipeline=CHAN0)
              for g in range(2)]
        pk[1] = pk[1][::-1]
        vis, _, nfb = _run(gpu, pk, nstand, nchan, ntime, tables=tables)
        assert nfb == 2 and _run.irregular == 0 and np.array_equal(vis, want), tables


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

def _beam_init(ffi, mode, ninput, nchan, ntime, nbeam, ntime_blocks=0, tables=None):
    """tables: XENG_SLAB_TABLES for this context ("1": the parts are read through their packet indices from the first call on)"""
    import os
    old = {k: os.environ.pop(k, None) for k in ("XENG_BEAM", "XENG_SLAB_TABLES")}
    if mode:
        os.environ["XENG_BEAM"] = mode
    if tables is not None:
        os.environ["XENG_SLAB_TABLES"] = tables
    try:
        ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, ntime_blocks)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _beam_weights(rng, nchan, nbeam, ninput):
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    w[:, :, 3] *= 4096.0                                                       # an outlier input on every row
    w[0] *= np.exp(rng.uniform(-12, 12, (nbeam, ninput))).astype(np.float32)    # channel 0: routed to the bf16x3 kernel
    return w


@pytest.mark.parametrize("tables", [None, "1"])
@pytest.mark.parametrize("mode", ["", "bf16x3", "f32"])
@pytest.mark.parametrize("nstand,nchan,ntime,nbeam,ntime0", [(352, 96, 960, 32, 480), (96, 8, 256, 32, 64), (64, 5, 192, 6, 0)])
def test_beamformer_reads_regular_slabs_in_place(gpu, mode, nstand, nchan, ntime, nbeam, ntime0, tables):
    """one or two regular slabs as a beamformer gulp: no part takes the scatter, and the beams are BIT-IDENTICAL to
    xengBeamformRun on the unpacked gulp (the same kernels do the arithmetic; only the addresses differ) -- on all three
    kernel routes, with outlier inputs and routed tiles in play"""
    ffi = gpu.ffi
    ninput = nstand * 2
    rng = np.random.default_rng(nstand + ntime0)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=ntime + nchan)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    _beam_init(ffi, mode, ninput, nchan, ntime, nbeam, tables=tables)
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
    nfb, nir = ctypes.c_int(-1), ctypes.c_int(-1)
    ffi.call("xengBeamformGetSlabStats", ctypes.byref(nfb), ctypes.byref(nir))
    assert nfb.value == 0 and nir.value == 0
    assert np.array_equal(o1.download(np.uint32), o2.download(np.uint32))
    ffi.call("xengBeamformDestroy")
    for d in bufs + [dfull, dw, o1, o2]:
        d.free()


@pytest.mark.parametrize("mode,tables", [("", None), ("f32", None), ("", "1"), ("bf16x3", "1"), ("f32", "1")])
def test_beamformer_irregular_slabs_give_the_unpacked_result(gpu, mode, tables):
    """a regular first part and a second part with lost, reordered and foreign packets; then both parts irregular: the beams
    equal xengBeamformRun on what snap2_unpack makes of the packets (missing samples zero), bit for bit -- scattered first (a link that
    has been clean; always with the fp32 kernel) or, through the packet indices (round 5), read where they lie"""
    ffi = gpu.ffi
    nstand, nchan, ntime, nbeam, ntime0 = 96, 8, 256, 32, 128
    ninput = nstand * 2
    rng = np.random.default_rng(77)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=5)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    _beam_init(ffi, mode, ninput, nchan, ntime, nbeam, tables=tables)
    by_index = tables == "1" and mode != "f32"
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
    ncall = 0
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
        nfb, nir = ctypes.c_int(-1), ctypes.c_int(-1)
        ffi.call("xengBeamformGetSlabStats", ctypes.byref(nfb), ctypes.byref(nir))
        # (the shipped default goes over to the indices once a part was not regular: the first call scatters, the later ones do not)
        index_now = by_index or (tables is None and mode != "f32" and ncall > 0)
        assert (nfb.value, nir.value) == ((0, nexp) if index_now else (nexp, 0)), (ncall, nfb.value, nir.value)
        ncall += 1
        assert np.array_equal(o1.download(np.uint32), o2.download(np.uint32)), nexp
        for d in (dfull, d0, d1):
            d.free()
    with pytest.raises(ffi.XengError):       # parts must sit on 16-sample boundaries
        ffi.call("xengBeamformRunSlabs", o1.ptr, 10, 100, o1.ptr, 10, 6176, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
    ffi.call("xengBeamformDestroy")
    for d in (dw, o1, o2):
        d.free()


def test_beamformer_shifted_and_lossy_parts_through_the_indices(gpu):
    """what a lossy link leaves (round 5): part 0 in arrival order with 2 % of its packets lost (everything behind a loss one slot
    early, the slab shorter), part 1 with a whole sample, a whole 64-input block and an outlier input's packets missing and one sample
    carried twice with different payloads (the later wins) -- read where they lie through the packet indices, bit-identical to
    xengBeamformRun on what snap2_unpack makes of them; then the default context: scattered at first, by index from the second call on"""
    ffi = gpu.ffi
    nstand, nchan, ntime, nbeam, ntime0 = 96, 8, 256, 32, 128
    ninput, nblk = nstand * 2, nstand * 2 // 64
    rng = np.random.default_rng(8)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=15)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    mk = lambda lo, hi, v=vin: orc.snap2_packets(v[lo:hi], seq0=SEQ0 + lo, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
    p0, p1 = mk(0, ntime0), mk(ntime0, ntime)
    lost = set(rng.choice(len(p0), size=len(p0) // 50, replace=False).tolist())
    q0 = [pk for i, pk in enumerate(p0) if i not in lost]
    other = mk(ntime0, ntime, vin[::-1])
    q1 = [pk for i, pk in enumerate(p1) if i // nblk != 5 and i % nblk != 0]
    q1[3] = other[40 * nblk + 1]                             # an early copy of (sample 40, block 1) that loses ...
    q1.append(other[50 * nblk + 2])                          # ... and a late one of (sample 50, block 2) that wins
    g0, _, _ = orc.snap2_unpack(q0, SEQ0, ntime0, CHAN0, nchan, ninput)
    g1, _, _ = orc.snap2_unpack(q1, SEQ0 + ntime0, ntime - ntime0, CHAN0, nchan, ninput)
    unpacked = np.concatenate([g0, g1])
    (r0, stride), (r1, _) = _slab(q0), _slab(q1)
    for mode, tables in (("", "1"), ("bf16x3", "1"), ("", None)):
        _beam_init(ffi, mode, ninput, nchan, ntime, nbeam, tables=tables)
        dfull = ffi.DeviceBuffer(unpacked.size).upload(unpacked.reshape(-1))
        d0, d1 = ffi.DeviceBuffer(r0.size).upload(r0), ffi.DeviceBuffer(r1.size).upload(r1)
        dw = ffi.DeviceBuffer(w.nbytes).upload(w)
        o1, o2 = ffi.DeviceBuffer(nchan * nbeam * ntime * 8), ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
        ffi.call("xengBeamformRunVersioned", dfull.ptr, o1.ptr, dw.ptr, 1)
        seen = []
        for call in range(3):
            ffi.call("xengMemset", o2.ptr, 0x5A, o2.nbytes)
            ffi.call("xengBeamformRunSlabs", d0.ptr, r0.size // stride, ntime0, d1.ptr, r1.size // stride, stride, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
            ffi.call("xengBeamformSync")
            nfb, nir = ctypes.c_int(-1), ctypes.c_int(-1)
            ffi.call("xengBeamformGetSlabStats", ctypes.byref(nfb), ctypes.byref(nir))
            seen.append((nfb.value, nir.value))
            assert np.array_equal(o1.download(np.uint32), o2.download(np.uint32)), (mode, tables, call)
        assert seen == ([(0, 2)] * 3 if tables else [(2, 0), (0, 2), (0, 2)]), (mode, tables, seen)
        ffi.call("xengBeamformDestroy")
        for d in (dfull, d0, d1, dw, o1, o2):
            d.free()


def test_beamformer_index_generations_wrap(gpu):
    """the packet indices are never cleared per call: an entry is valid only if it carries the call's 12-bit generation.  At the
    wrap the indices are cleared, so that an entry written 4095 calls ago does not come back to life.  Started three calls before
    the wrap (test hook XENG_SLAB_GEN0), alternating a part with (sample 9, block 1) present and the same part without it:
    the lost row must read as zero on every call, also on the calls right after the wrap."""
    import os
    ffi = gpu.ffi
    nstand, nchan, ntime, nbeam = 96, 8, 128, 32
    ninput, nblk = nstand * 2, nstand * 2 // 64
    rng = np.random.default_rng(12)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=19)
    w = _beam_weights(rng, nchan, nbeam, ninput)
    full = orc.snap2_packets(vin, seq0=SEQ0, sync_time=1, nchan_blocks=1, nstand_per_pkt=32, chan0_pipeline=CHAN0)
    holed = [pk for i, pk in enumerate(full) if i != 9 * nblk + 1]
    os.environ["XENG_SLAB_GEN0"] = str(4095 - 3)
    try:
        _beam_init(ffi, "", ninput, nchan, ntime, nbeam, tables="1")
        dw = ffi.DeviceBuffer(w.nbytes).upload(w)
        o1, o2 = ffi.DeviceBuffer(nchan * nbeam * ntime * 8), ffi.DeviceBuffer(nchan * nbeam * ntime * 8)
        wants, bufs = {}, {}
        for name, pk in (("full", full), ("holed", holed)):
            g, _, _ = orc.snap2_unpack(pk, SEQ0, ntime, CHAN0, nchan, ninput)
            d = ffi.DeviceBuffer(g.size).upload(g.reshape(-1))
            ffi.call("xengBeamformRunVersioned", d.ptr, o1.ptr, dw.ptr, 1)
            ffi.call("xengBeamformSync")
            wants[name] = o1.download(np.uint32)
            raw, stride = _slab(pk)
            bufs[name] = (ffi.DeviceBuffer(raw.size).upload(raw), len(pk), stride)
            d.free()
        assert not np.array_equal(wants["full"], wants["holed"])
        for call in range(8):                                 # generations 4093, 4094, 4095, (wrap) 1, 2, ...
            name = "full" if call % 2 == 0 else "holed"
            d, npk, stride = bufs[name]
            ffi.call("xengBeamformRunSlabs", d.ptr, npk, ntime, None, 0, stride, SEQ0, CHAN0, o2.ptr, dw.ptr, 1)
            ffi.call("xengBeamformSync")
            assert np.array_equal(o2.download(np.uint32), wants[name]), (call, name)
    finally:
        os.environ.pop("XENG_SLAB_GEN0", None)
    ffi.call("xengBeamformDestroy")
    for b in [dw, o1, o2] + [v[0] for v in bufs.values()]:
        b.free()


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
