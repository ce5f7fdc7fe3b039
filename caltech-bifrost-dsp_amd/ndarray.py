"""Arrays in the three memory spaces the hot-path blocks use.

Stands in for `bifrost.ndarray` (BFArray) as the reference blocks use it:
`BFArray(np_array | shape=, dtype=, space=)`, `.as_BFarray()`, `.copy(space=)`,
slice-assignment across spaces (beamform_block.py:433, beamform_sum_beams_block.py:246),
`copy_array(dst, src)` (corr_acc_block.py:315), `span.data_view(dtype)`.

Spaces keep bifrost's names: 'system' (numpy memory, needs no GPU), 'cuda' (HIP device
memory) and 'cuda_host' (pinned host memory); the latter two go through libxeng.
"""
import ctypes

import math

import numpy as np

from . import ffi

_SPACE_ID = {"system": ffi.SPACE_SYSTEM, "cuda": ffi.SPACE_CUDA, "cuda_host": ffi.SPACE_CUDA_HOST}

# bifrost dtype strings used by the blocks -> numpy
_DTYPES = {"i8": np.int8, "u8": np.uint8, "i32": np.int32, "i64": np.int64, "f32": np.float32,
           "cf32": np.complex64, "ci32": np.dtype([("re", np.int32), ("im", np.int32)]),
           "ci4": np.uint8}


_DT_CACHE = {}
_U8 = np.dtype(np.uint8)


def to_dtype(dt):
    try:
        return _DT_CACHE[dt]              # (per-gulp path: data_view('i8'), view(np.float32) ...)
    except (KeyError, TypeError):
        pass
    d = np.dtype(_DTYPES[dt]) if isinstance(dt, str) and dt in _DTYPES else np.dtype(dt)
    try:
        _DT_CACHE[dt] = d
    except TypeError:                     # unhashable spec (a list of fields)
        pass
    return d


class XArray:
    """A typed, shaped window on memory in one space.  `base` keeps the owner alive."""

    def __init__(self, data=None, shape=None, dtype=None, space="system", _ptr=None, _base=None):
        if space not in _SPACE_ID:
            raise ValueError("unknown space %r" % (space,))
        self.space = space
        if _ptr is not None:                       # window on existing memory (the per-gulp path of every block: kept short)
            self.dtype = dtype if isinstance(dtype, np.dtype) else to_dtype(dtype)
            self.shape = shape = tuple(int(s) for s in shape)
            self.size = math.prod(shape) if shape else 1
            self.nbytes = self.size * self.dtype.itemsize
            self.ptr = int(_ptr)
            self.base = _base
            return
        if data is not None:
            src = np.ascontiguousarray(data, dtype=None if dtype is None else to_dtype(dtype))
            shape, dtype = src.shape, src.dtype
        else:
            src = None
            if isinstance(shape, int):
                shape = (shape,)
        self.dtype = to_dtype(dtype)
        self.shape = tuple(int(s) for s in shape)
        self.size = math.prod(self.shape) if self.shape else 1      # (shape and dtype are fixed for the life of the array)
        self.nbytes = nbytes = self.size * self.dtype.itemsize
        if space == "system":
            self.base = np.zeros(max(nbytes, 1), dtype=np.uint8)
            self.ptr = self.base.ctypes.data
        else:
            self.base = ffi.DeviceBuffer(max(nbytes, 1), _SPACE_ID[space])
            self.ptr = self.base.ptr
            if src is None:
                ffi.call("xengMemset", self.ptr, 0, max(nbytes, 1))
        if src is not None:
            self[...] = src

    @classmethod
    def window(cls, ptr, nbytes, space, base):
        """A byte window on existing memory -- what a ring hands out per gulp (the general constructor spends as long on its
        argument handling as the native ring call spends in all)."""
        a = cls.__new__(cls)
        a.space, a.dtype, a.shape, a.size, a.nbytes, a.ptr, a.base = space, _U8, (nbytes,), nbytes, nbytes, ptr, base
        return a

    # ------------------------------------------------------------------ geometry (size, nbytes: set once in __init__)
    def view(self, dtype):
        dtype = to_dtype(dtype)
        assert self.nbytes % dtype.itemsize == 0
        return XArray(shape=(self.nbytes // dtype.itemsize,), dtype=dtype, space=self.space, _ptr=self.ptr, _base=self)

    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            k = shape.index(-1)
            rest = math.prod(int(s) for s in shape if s != -1)
            shape[k] = self.size // rest
        assert math.prod(int(s) for s in shape) == self.size, (shape, self.shape)
        return XArray(shape=shape, dtype=self.dtype, space=self.space, _ptr=self.ptr, _base=self)

    def byte_slice(self, offset, nbytes):
        assert 0 <= offset and offset + nbytes <= self.nbytes
        return XArray(shape=(nbytes,), dtype=np.uint8, space=self.space, _ptr=self.ptr + offset, _base=self)

    # ------------------------------------------------------------------ access
    def numpy(self):
        """Zero-copy numpy view for host spaces; a downloaded copy for 'cuda'."""
        if self.space == "cuda":
            out = np.empty(self.shape, dtype=self.dtype)
            if self.nbytes:
                ffi.call("xengMemcpy", out.ctypes.data, self.ptr, self.nbytes)
            return out
        buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        buf._owner = self       # the numpy view keeps `buf` alive, and `buf` keeps the allocation alive
        a = np.frombuffer(buf, dtype=self.dtype, count=self.size).reshape(self.shape)
        return a

    def __setitem__(self, key, value):
        if key is not Ellipsis and key != slice(None):
            if self.space == "cuda":
                raise NotImplementedError("partial assignment into device memory")
            self.numpy()[key] = value.numpy() if isinstance(value, XArray) else value
            return
        copy_array(self, value)

    def copy(self, space=None):
        out = XArray(shape=self.shape, dtype=self.dtype, space=space or self.space)
        copy_array(out, self)
        return out

    def as_BFarray(self):
        """What the bf*-named entry points take (include/xeng.h XENGarray = bifrost BFarray): a reference that turns into
        the struct pointer when ctypes passes it to a bf* function, and answers `.contents.data` -- all the raw xeng* calls
        of the backend need -- without building the struct (two fresh views per gulp in every block)."""
        # (a fresh reference every time, never stored on the array: array -> reference -> array would be a cycle, and span
        # memory must return to its ring the moment the last user lets go of it, not when the cycle collector next runs)
        return _BFRef(self)

    def _xeng_array(self):
        a = ffi.XENGarray()
        a.data = self.ptr
        a.space = _SPACE_ID[self.space]
        a.dtype = 0
        a.ndim = len(self.shape)
        stride = self.dtype.itemsize
        for k in range(len(self.shape) - 1, -1, -1):
            a.shape[k] = self.shape[k]
            a.strides[k] = stride
            stride *= self.shape[k]
        return a


class _BFRef:
    """`XArray.as_BFarray()`: quacks like ctypes.pointer(XENGarray) for the uses the backends make of it."""
    __slots__ = ("arr", "_struct", "_ptr")

    def __init__(self, arr):
        self.arr = arr
        self._struct = None
        self._ptr = None

    @property
    def contents(self):                   # .contents.data / .contents.shape[...] as on a ctypes pointer
        if self._struct is None:
            self._struct = self.arr._xeng_array()
        return self._struct

    @property
    def data(self):
        return self.arr.ptr

    @property
    def _as_parameter_(self):             # ctypes: passed where POINTER(XENGarray) is expected
        if self._ptr is None:
            self._ptr = ctypes.pointer(self.contents)
        return self._ptr


def copy_array(dst, src):
    """bifrost.ndarray.copy_array(dst, src): whole-array copy between any two spaces."""
    if not isinstance(src, XArray):
        src_np = np.ascontiguousarray(src, dtype=dst.dtype)
        assert src_np.nbytes == dst.nbytes, (src_np.shape, dst.shape)
        if dst.space == "system":
            dst.numpy()[...] = src_np.reshape(dst.shape)
        else:
            ffi.call("xengMemcpy", dst.ptr, src_np.ctypes.data, dst.nbytes)
        return dst
    assert src.nbytes == dst.nbytes, (src.shape, src.dtype, dst.shape, dst.dtype)
    if dst.space == "system" and src.space == "system":
        ctypes.memmove(dst.ptr, src.ptr, dst.nbytes)
    elif dst.nbytes:
        ffi.call("xengMemcpy", dst.ptr, src.ptr, dst.nbytes)
    return dst
