"""The calls the blocks make with the interpreter lock kept never wait (GPU; round-3 ADVICE, review item 6).

`xengXgpuTryKernelAsyncAcc` returns XENG_STATUS_WOULD_BLOCK instead of waiting when the caller is 256 launches ahead of the
GPU; `_xfast.xgpu_kernel_async` (what the Corr block calls) gives the lock up for that wait, so the other block threads keep
running; the results are the ones of the waiting calls."""
import ctypes
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import xeng_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util
    assert gpu_util.ffi.device_count() >= 1
    return gpu_util


def test_try_enqueue_reports_would_block_instead_of_waiting(gpu):
    ffi = gpu.ffi
    nstand, nchan, ntime = 352, 96, 480            # one launch per call, 40 us of GPU work each: 300 calls outrun the GPU
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=1)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=3)
    din = ffi.DeviceBuffer(vin.size).upload(vin)
    L = ffi.lib()
    # No wall-clock bound (the host is shared): what is asserted is the protocol.  A Try call either enqueues (0) or says
    # WOULD_BLOCK -- and it says so exactly while the GPU is far behind: at that moment the dump three before the latest has not
    # completed (256 launches are in flight), which a call that had WAITED for a slot could never observe.
    blocked, behind = 0, 0
    for k in range(600):
        rc = L.xengXgpuTryKernelAsyncAcc(din.ptr, x.out.ptr, 1, None, 0)
        if rc == ffi.STATUS_WOULD_BLOCK:
            blocked += 1
            done = ctypes.c_int(-1)
            ffi.call("xengXgpuDumpDone", 3, ctypes.byref(done))
            behind += int(done.value == 0)
            ffi.call("xengXgpuWaitLaunchSlot")
            ffi.call("xengXgpuTryKernelAsyncAcc", din.ptr, x.out.ptr, 1, None, 0)
        else:
            assert rc == 0
    ffi.call("xengXgpuSync")
    assert blocked > 0, "600 enqueue-only launches never got 256 ahead of the GPU"
    assert behind == blocked, "WOULD_BLOCK was reported %d times, %d of them with the GPU less than 4 launches behind" % (blocked, blocked - behind)
    assert np.array_equal(x.out.download(np.int32), orc.xgpu_correlate(vin, nstand, nchan))
    din.free()
    x.close()


def test_a_caller_far_ahead_of_the_gpu_does_not_keep_the_interpreter_lock(gpu):
    """While one Python thread enqueues 800 launches through _xfast (lock kept per call, given up to wait), another keeps
    running: its longest stall stays far below the time the enqueuer spends waiting for the GPU."""
    from caltech_bifrost_dsp_amd import _xfast
    ffi = gpu.ffi
    nstand, nchan, ntime = 352, 96, 480
    x = gpu.Xgpu(nstand, nchan, ntime, max_gulps=1)
    vin = gpu.synth_voltages(ntime, nchan, nstand, "full", seed=4)
    din = ffi.DeviceBuffer(vin.size).upload(vin)
    stop, ticks = threading.Event(), [0]

    def ticker():
        while not stop.is_set():
            time.sleep(0.0002)
            ticks[0] += 1

    th = threading.Thread(target=ticker, daemon=True)
    th.start()
    while ticks[0] < 5:                       # (the ticker is running)
        time.sleep(0.001)
    # 2000 launches of ~40 us: the enqueuer spends ~70 ms waiting for launch slots.  A count, not a clock: had it kept the lock
    # while it waited, the ticker would not have moved at all during those waits (it needs the lock for every tick); with the lock
    # given up it ticks a few hundred times.  The bound is a tenth of that.
    before = ticks[0]
    for k in range(2000):
        assert _xfast.xgpu_kernel_async(din.ptr, x.out.ptr, 1) == 0
    during = ticks[0] - before
    ffi.call("xengXgpuSync")
    stop.set()
    th.join(5)
    assert np.array_equal(x.out.download(np.int32), orc.xgpu_correlate(vin, nstand, nchan))
    assert during >= 30, "the other thread ticked %d times while 2000 launches were enqueued: the enqueuer kept the interpreter lock while it waited" % during
    din.free()
    x.close()


def test_try_run_in_integrated_power_mode_never_waits(gpu):
    ffi = gpu.ffi
    ninput, nchan, ntime, nbeam, ns = 704, 96, 480, 32, 24
    rng = np.random.default_rng(2)
    w = (rng.uniform(-17, 17, (nchan, nbeam, ninput)) + 1j * rng.uniform(-17, 17, (nchan, nbeam, ninput))).astype(np.complex64)
    vin = rng.integers(0, 256, ntime * nchan * ninput, dtype=np.uint8)
    ffi.call("xengBeamformInitialize", 0, ninput, nchan, ntime, nbeam, ntime // ns)
    di, dw = ffi.DeviceBuffer(vin.size).upload(vin), ffi.DeviceBuffer(w.nbytes).upload(w)
    do = ffi.DeviceBuffer((nbeam // 2) * (ntime // ns) * nchan * 16)
    L = ffi.lib()
    rc = L.xengBeamformTryRunVersioned(di.ptr, do.ptr, dw.ptr, 7)      # a fresh weight upload: the routing answer is not back yet
    assert rc in (0, ffi.STATUS_WOULD_BLOCK)
    ffi.call("xengBeamformRunVersioned", di.ptr, do.ptr, dw.ptr, 7)    # the waiting form: runs
    ffi.call("xengBeamformSync")
    assert L.xengBeamformTryRunVersioned(di.ptr, do.ptr, dw.ptr, 7) == 0      # same weights again: nothing to wait for
    ffi.call("xengBeamformSync")
    got = do.download(np.float32).reshape(nbeam // 2, ntime // ns, nchan, 4)
    exp = orc.beamform_integrate(orc.beamform(vin.reshape(ntime, nchan, ninput)[:, :2].copy(), w[:2].copy(), ntime, 2, ninput, nbeam), ns)
    assert np.allclose(got[:, :, :2], exp, rtol=2e-5, atol=2e-5 * np.abs(exp).max())
    ffi.call("xengBeamformDestroy")
    for b in (di, dw, do):
        b.free()
