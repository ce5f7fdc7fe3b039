"""Helpers shared by the GPU parity tests (they go through the C ABI via ffi.py)."""
import ctypes

import numpy as np

import caltech_bifrost_dsp_amd  # noqa: F401  (import shim)
from caltech_bifrost_dsp_amd import ffi
from oracle import xeng_oracle as orc


def synth_voltages(ntime, nchan, nstand, kind="random", seed=0xdeadbeef):
    """Synthetic F-engine voltages uint8[T,C,S,2]: 'random' = the generator of
    make_golden_inputs.py:57,137; '88' = every nibble -8; 'chanramp' = make_golden_inputs.py:112-116."""
    if kind == "random":
        return np.random.RandomState(seed).randint(0, 255, size=(ntime, nchan, nstand, 2), dtype=np.uint8)
    if kind == "88":
        return np.full((ntime, nchan, nstand, 2), 0x88, dtype=np.uint8)
    if kind == "chanramp":
        d = np.zeros((ntime, nchan, nstand, 2), dtype=np.uint8)
        d[...] = (np.arange(nchan, dtype=np.uint32) & 0xFF).astype(np.uint8)[None, :, None, None]
        return d
    if kind == "full":   # all 256 byte values, including 0xFF which randint(0,255) never emits
        return np.random.RandomState(seed).randint(0, 256, size=(ntime, nchan, nstand, 2), dtype=np.uint8)
    raise ValueError(kind)


class Xgpu:
    """Thin test driver around xengXgpu* (one process-global context)."""

    def __init__(self, nstand, nchan, ntime_gulp, max_gulps=0, gpu=0):
        self.nstand, self.nchan, self.ntime = nstand, nchan, ntime_gulp
        ffi.call("xengXgpuConfigure", nstand, 2, nchan, ntime_gulp, max_gulps)
        ffi.call("xengXgpuInitialize", gpu)
        self.matlen = orc.per_chan(nstand) * nchan
        self.gulp_bytes = ntime_gulp * nchan * nstand * 2
        self.out = ffi.DeviceBuffer(self.matlen * 8)
        self.inbuf = None

    def run(self, vin, dumps_every=None, use_async=False, poison=True):
        """Feed vin (uint8[G*ntime, C, S, 2]) gulp by gulp, dump on the last; returns planar int32."""
        vin = np.ascontiguousarray(vin, dtype=np.uint8).reshape(-1)
        ngulp = vin.size // self.gulp_bytes
        assert ngulp * self.gulp_bytes == vin.size
        if self.inbuf is None or self.inbuf.nbytes < vin.size:
            self.inbuf = ffi.DeviceBuffer(vin.size)
        self.inbuf.upload(vin)
        if poison:
            ffi.call("xengMemset", self.out.ptr, 0x5A, self.out.nbytes)
        fn = "xengXgpuKernelAsync" if use_async else "xengXgpuKernel"
        for g in range(ngulp):
            ffi.call(fn, self.inbuf.ptr + g * self.gulp_bytes, self.out.ptr, int(g == ngulp - 1))
        if use_async:
            ffi.call("xengXgpuSync")
        return self.out.download(np.int32)

    def path(self):
        """(fused_corner_turn, fp6) of the live context."""
        a, b = ctypes.c_int(), ctypes.c_int()
        ffi.call("xengXgpuGetPath", ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def close(self):
        ffi.call("xengXgpuDestroy")
        self.out.free()
        if self.inbuf is not None:
            self.inbuf.free()
