"""No kernel of the library writes outside the buffer it was given (GPU; through the C ABI).

Every parity test compares the bytes INSIDE an output buffer.  A write just outside one lands, in the running pipeline, in a
neighbouring allocation that the rings keep alive -- silent -- and becomes a GPU memory fault the day that neighbour is
really freed (round 3: one fault when span memory was freed at cycle-collector time).  Here every output --
at the production shapes of BASELINE configs 2, 4 and 5 and at ragged small ones -- sits between two 2 MiB guard bands of
a known pattern, and the bands must come back untouched."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402

G = 2 << 20
PATTERN = 0xA5


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util
    assert gpu_util.ffi.device_count() >= 1
    return gpu_util


class Guarded:
    """payload of `nbytes` (at .ptr, 2 MiB-aligned like a fresh allocation) between two guard bands"""

    def __init__(self, ffi, nbytes, space=None, fill=0x5A):
        self.ffi, self.nbytes = ffi, int(nbytes)
        self.space = ffi.SPACE_CUDA if space is None else space
        self.buf = ffi.DeviceBuffer(2 * G + self.nbytes + 64, self.space)
        ffi.call("xengMemset", self.buf.ptr, PATTERN, self.buf.nbytes)
        self.ptr = self.buf.ptr + G
        if self.nbytes:
            ffi.call("xengMemset", self.ptr, fill, self.nbytes)

    def check(self, what):
        lo = self.buf.download(np.uint8, G, 0)
        hi = self.buf.download(np.uint8, G + 64, G + self.nbytes)
        bad_lo, bad_hi = np.flatnonzero(lo != PATTERN), np.flatnonzero(hi != PATTERN)
        assert bad_lo.size == 0, "%s: %d bytes written BEFORE the buffer (nearest at -%d)" % (what, bad_lo.size, G - bad_lo[-1])
        assert bad_hi.size == 0, "%s: %d bytes written PAST the end of the buffer (first at +%d)" % (what, bad_hi.size, bad_hi[0])

    def payload(self, dtype):
        return self.buf.download(dtype, self.nbytes // np.dtype(dtype).itemsize, G)

    def free(self):
        self.buf.free()


def _env(**kv):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kv}
            for k, v in kv.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

        def __exit__(self, *exc):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


_WANT = {}
XCORR_SHAPES = [
    # nstand, nchan, ntime, ngulp, env
    (352, 96, 480, 5, {}),                       # config 2 / 5: the shipped fused kernel
    (352, 96, 480, 5, {"XENG_RAW": "0"}),        # two-pass path (corner turn + xcorr_mfma_kernel)
    (352, 96, 480, 5, {"XENG_TILING": "64"}),    # the 64x64 tiling
    (352, 96, 480, 5, {"XENG_KLOOP": "16"}),     # the eight-wave 16x16x64 kernel (round 5, opt-in)
    (344, 8, 96, 2, {}),                         # 688 inputs: last block three quarters full
    (96, 8, 192, 2, {}),                         # 3 blocks
    (16, 4, 8, 2, {}),                           # config 1 shape: two-pass path (ntime not a multiple of 96)
    (36, 5, 40, 3, {}),                          # ragged: 72 inputs
]


@pytest.mark.parametrize("nstand,nchan,ntime,ngulp,env", XCORR_SHAPES)
def test_contraction_stays_inside_its_output_and_accumulator(gpu, nstand, nchan, ntime, ngulp, env):
    ffi = gpu.ffi
    with _env(**env):
        ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime, ngulp)
        ffi.call("xengXgpuInitialize", 0)
    matlen = orc.per_chan(nstand) * nchan
    gulp = ntime * nchan * nstand * 2
    vin = gpu.synth_voltages(ngulp * ntime, nchan, nstand, "full", seed=nstand + ntime)
    din = Guarded(ffi, ngulp * gulp)                 # (inputs are only read; the bands around them must survive too)
    ffi.call("xengMemcpy", din.ptr, vin.ctypes.data, vin.nbytes)
    out, acc = Guarded(ffi, matlen * 8), Guarded(ffi, matlen * 8)
    fused, _ = ctypes.c_int(), ctypes.c_int()
    ffi.call("xengXgpuGetPath", ctypes.byref(fused), ctypes.byref(_))
    for g in range(ngulp):
        ffi.call("xengXgpuKernel", din.ptr + g * gulp, out.ptr, int(g == ngulp - 1))
    key = (nstand, nchan, ntime, ngulp)
    if key not in _WANT:
        _WANT.clear()                                # (one 191 MB result at a time)
        _WANT[key] = orc.xgpu_correlate(vin, nstand, nchan)
    want = _WANT[key]
    assert np.array_equal(out.payload(np.int32), want)
    out.check("synchronous dump")
    if fused.value:
        for mode in (1, 2):
            for g in range(ngulp):
                ffi.call("xengXgpuKernelAsyncAcc", din.ptr + g * gulp, out.ptr, int(g == ngulp - 1), acc.ptr, mode)
            ffi.call("xengXgpuSync")
        assert np.array_equal(acc.payload(np.int32), 2 * want)
        out.check("dump with the long accumulation fused in")
        acc.check("long accumulator of the fused dump")
    din.check("input gulps")
    # the consumers of the span
    bl, cj = orc.xgpu_get_order(np.arange(nstand * 2, dtype=np.int32).reshape(nstand, 2))
    if nchan % 4 == 0:
        nvis = 4704 if nstand == 352 else 37
        rng = np.random.default_rng(1)
        s0, s1, p0, p1 = rng.integers(0, nstand, nvis), rng.integers(0, nstand, nvis), rng.integers(0, 2, nvis), rng.integers(0, 2, nvis)
        vm, cjv = bl[s0, s1, p0, p1].astype(np.int32), cj[s0, s1, p0, p1].astype(np.int32)
        dv, dc = ffi.DeviceBuffer(vm.nbytes).upload(vm), ffi.DeviceBuffer(cjv.nbytes).upload(cjv)
        sub = Guarded(ffi, (nchan // 4) * nvis * 8)
        ffi.call("xengXgpuSubSelect", out.ptr, sub.ptr, dv.ptr, dc.ptr, nvis, 4)
        assert np.array_equal(sub.payload(np.int32).reshape(nchan // 4, nvis, 2), orc.xgpu_subselect(want, vm, cjv, nchan, 4, nstand))
        sub.check("SubSelect")
        for b in (dv, dc):
            b.free()
        sub.free()
    dbl, dcj = ffi.DeviceBuffer(bl.nbytes).upload(np.ascontiguousarray(bl)), ffi.DeviceBuffer(cj.nbytes).upload(np.ascontiguousarray(cj))
    pk = Guarded(ffi, (nstand * (nstand + 1) // 2) * 4 * nchan * 8)
    for fmt in (0, 1):
        ffi.call("xengXgpuPacketize", out.ptr, pk.ptr, dbl.ptr, dcj.ptr, fmt)
        pk.check("Packetize fmt %d" % fmt)
    ffi.call("xengXgpuDestroy")
    for b in (din, out, acc, pk):
        b.free()
    for b in (dbl, dcj):
        b.free()


@pytest.mark.parametrize("n", [1, 3, 1000, 47849472, 4 * 1024 * 1024 + 3])
def test_map_stays_inside_its_accumulator(gpu, n):
    ffi = gpu.ffi
    b = np.random.default_rng(n).integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    a, db = Guarded(ffi, 4 * n, fill=0), Guarded(ffi, 4 * n)
    ffi.call("xengMemcpy", db.ptr, b.ctypes.data, b.nbytes)
    ffi.call("xengMapAssignI32", a.ptr, db.ptr, n)
    ffi.call("xengMapAddI32", a.ptr, db.ptr, n)
    ffi.call("xengMapSync")
    assert np.array_equal(a.payload(np.int32), (2 * b.astype(np.int64)).astype(np.int32))
    a.check("map a = b / a += b")
    db.check("map source")
    a.free()
    db.free()


BEAM_SHAPES = [
    # ninput, nchan, ntime, nbeam, ntime_sum, env
    (704, 96, 480, 32, 24, {}),                          # config 5: 480-sample gulps (3.75 work-groups of 128 samples)
    (704, 96, 960, 32, 24, {}),                          # config 4
    (704, 96, 480, 32, 24, {"XENG_BEAM": "bf16x3"}),
    (704, 16, 480, 32, 24, {"XENG_BEAM": "f32"}),
    (80, 16, 120, 32, 24, {}),                           # the reference's beamformer test shape (beamformer_test.py:121-139)
    (192, 5, 100, 6, 10, {}),                            # ragged: 6 beams, 100 samples
    (36, 3, 48, 2, 8, {}),                               # ninput % 16 != 0: the fp32 kernel
]


@pytest.mark.parametrize("ninput,nchan,ntime,nbeam,ntime_sum,env", BEAM_SHAPES)
@pytest.mark.parametrize("heavy_tail", [False, True])
def test_beamformer_stays_inside_its_outputs(gpu, ninput, nchan, ntime, nbeam, ntime_sum, env, heavy_tail):
    ffi = gpu.ffi
    rng = np.random.default_rng(ninput + ntime)
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    if heavy_tail:                                       # outlier inputs + tiles routed to the bf16x3 kernel
        w[:, :, 3] *= 4096.0
        w[0] *= np.exp(rng.uniform(-12, 12, (nbeam, ninput))).astype(np.float32)
    vin = rng.integers(0, 256, ntime * nchan * ninput, dtype=np.uint8)
    din, dw = Guarded(ffi, vin.nbytes), Guarded(ffi, w.nbytes)
    ffi.call("xengMemcpy", din.ptr, vin.ctypes.data, vin.nbytes)
    ffi.call("xengMemcpy", dw.ptr, w.ctypes.data, w.nbytes)
    nblk = ntime // ntime_sum
    for mode_blocks in (0, nblk):                        # voltage mode, then the integrated-power mode
        with _env(**env):
            ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, mode_blocks)
        if mode_blocks == 0:
            out = Guarded(ffi, nchan * nbeam * ntime * 8)
            for ver in (0, 1, 1):                        # re-split, versioned first call, versioned reuse
                ffi.call("xengBeamformRunVersioned", din.ptr, out.ptr, dw.ptr, ver)
            ffi.call("xengBeamformSync")
            out.check("BeamformRun")
            beams = out.payload(np.complex64).reshape(nchan, nbeam, ntime)
            for c in sorted({0, nchan - 1}):             # (values: the parity tests' job; two channels here keep the run short)
                exp = orc.beamform(np.ascontiguousarray(vin.reshape(ntime, nchan, ninput)[:, c:c + 1]), np.ascontiguousarray(w[c:c + 1]), ntime, 1, ninput, nbeam)
                assert np.max(np.abs(beams[c:c + 1] - exp)) / np.sqrt(np.mean(np.abs(exp) ** 2)) < 1e-5, c
            for space in (ffi.SPACE_CUDA, ffi.SPACE_CUDA_HOST):      # BeamformSumBeams writes into pinned-host spans
                pw = Guarded(ffi, (nbeam // 2) * nblk * nchan * 16, space=space)
                ffi.call("xengBeamformIntegrate", out.ptr, pw.ptr, ntime_sum)
                ffi.call("xengBeamformSync")
                pw.check("BeamformIntegrate (%s)" % ("device" if space == ffi.SPACE_CUDA else "pinned host"))
                pw.free()
            one = Guarded(ffi, nblk * nchan * 16)
            ffi.call("xengBeamformIntegrateSingleBeam", out.ptr, one.ptr, ntime_sum, nbeam // 2 - 1)
            ffi.call("xengBeamformSync")
            one.check("BeamformIntegrateSingleBeam")
            one.free()
            out.free()
        else:
            pw = Guarded(ffi, (nbeam // 2) * nblk * nchan * 16)
            for ver in (1, 1, 1):                        # (the fused power epilogue starts once the routing answer is back)
                ffi.call("xengBeamformRunVersioned", din.ptr, pw.ptr, dw.ptr, ver)
                ffi.call("xengBeamformSync")
            pw.check("BeamformRun, integrated-power mode")
            pw.free()
    din.check("beamformer input")
    dw.check("beamformer weights")
    ffi.call("xengBeamformDestroy")
    din.free()
    dw.free()


@pytest.mark.parametrize("T,C,S,nchan_blocks,nstand_per_pkt,lose", [
    (480, 96, 352, 1, 32, 0), (480, 96, 352, 1, 32, 7), (6, 8, 64, 2, 32, 0), (5, 6, 12, 3, 3, 2)])
def test_ingest_scatter_stays_inside_the_gulp(gpu, T, C, S, nchan_blocks, nstand_per_pkt, lose):
    ffi = gpu.ffi
    rng = np.random.default_rng(T + S + lose)
    vin = rng.integers(0, 256, (T, C, S, 2), dtype=np.uint8)
    seq0, chan0 = 10 ** 12 + 5, 1000
    pk = orc.snap2_packets(vin, seq0=seq0, sync_time=9, nchan_blocks=nchan_blocks, nstand_per_pkt=nstand_per_pkt, chan0_pipeline=chan0)
    if lose:
        pk = pk[lose:]
    # packets outside the window / geometry must be dropped, not written somewhere
    stray = orc.snap2_packets(vin[:1], seq0=seq0 + T + 3, sync_time=9, nchan_blocks=nchan_blocks, nstand_per_pkt=nstand_per_pkt, chan0_pipeline=chan0)
    before = orc.snap2_packets(vin[:1], seq0=seq0 - 1, sync_time=9, nchan_blocks=nchan_blocks, nstand_per_pkt=nstand_per_pkt, chan0_pipeline=chan0)
    pk = list(pk) + list(stray) + list(before)
    stride = len(pk[0])
    raw = np.frombuffer(b"".join(pk), dtype=np.uint8)
    dpk = ffi.DeviceBuffer(raw.size).upload(raw)
    out = Guarded(ffi, T * C * S * 2)
    placed, dropped = ctypes.c_int(), ctypes.c_int()
    for clear in (1, 0):
        ffi.call("xengSnap2Unpack", dpk.ptr, len(pk), stride, out.ptr, seq0, T, chan0, C, S * 2, clear, ctypes.byref(placed), ctypes.byref(dropped))
        out.check("Snap2Unpack clear=%d" % clear)
        assert dropped.value == len(stray) + len(before)
        ffi.call("xengSnap2UnpackAsync", dpk.ptr, len(pk), stride, out.ptr, seq0, T, chan0, C, S * 2, clear)
        ffi.call("xengStreamSynchronize")
        out.check("Snap2UnpackAsync clear=%d" % clear)
    dpk.free()
    out.free()
