"""ctypes binding of libohhost.so -- the C++ host adapter (control plane: Jiffies, Ramp algebra, message
model).  Product code; tests and bench use it to build descriptors the way the pipeline's elements would."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libohhost.so")

RAMP_NONE, RAMP_UP, RAMP_DOWN, RAMP_MUTE = 0, 1, 2, 3


class Ramp(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("direction", C.c_uint32), ("enabled", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python ohpipeline_amd/build.py` first")
        from . import capi
        capi.lib()                      # libohhost.so links against libohgpu.so (the C ABI)
        L = C.CDLL(LIB_PATH)
        L.ohhost_jiffies_per_sample.restype = C.c_int
        L.ohhost_jiffies_per_sample.argtypes = [C.c_uint32]
        L.ohhost_ramp_set.restype = C.c_int
        L.ohhost_ramp_set.argtypes = [C.POINTER(Ramp), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(Ramp), C.POINTER(Ramp), C.POINTER(C.c_uint32)]
        L.ohhost_stream_ramp_schedule.restype = C.c_int
        L.ohhost_stream_ramp_schedule.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def stream_ramp_schedule(sizes_jiffies, up_jiffies, down_jiffies):
    """[(enabled, start, end)] per message: ramp up over the first up_jiffies, down over the last down_jiffies."""
    sizes = np.ascontiguousarray(sizes_jiffies, dtype=np.uint32)
    n = sizes.size
    flags = np.zeros(n, dtype=np.uint8)
    starts = np.zeros(n, dtype=np.uint16)
    ends = np.zeros(n, dtype=np.uint16)
    rc = lib().ohhost_stream_ramp_schedule(sizes.ctypes.data_as(C.c_void_p), n, up_jiffies, down_jiffies,
                                           flags.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p),
                                           ends.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"ohhost_stream_ramp_schedule failed: {rc}")
    return list(zip(flags.tolist(), starts.tolist(), ends.tolist()))
