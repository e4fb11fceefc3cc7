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
        L.ohhost_live_create.restype = C.c_int
        L.ohhost_live_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_void_p)]
        L.ohhost_live_tick.restype = C.c_int
        L.ohhost_live_tick.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.ohhost_live_stats.restype = C.c_int
        L.ohhost_live_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4 + [C.POINTER(C.c_uint32)]
        L.ohhost_live_destroy.restype = C.c_int
        L.ohhost_live_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class LiveDriver:
    """`lanes` chains of SampleRateConverter -> CreatePlayable behind one MsgFactory (one GPU context), read with ONE
    PlayableBatch::Run per tick: the C++ adapter's live path, as a driver thread would use it (host/ohhost_c.cpp)."""

    def __init__(self, device, lanes, rate_in, rate_out, channels, bits, little_endian=True, out_bits=24):
        self._h = C.c_void_p()
        rc = lib().ohhost_live_create(device, lanes, rate_in, rate_out, channels, bits, 1 if little_endian else 0, out_bits, C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"ohhost_live_create failed: {rc}")
        self.lanes = lanes
        self.frame_bytes = channels * (bits // 8)
        self._out_bytes = np.zeros(lanes, dtype=np.uint32)

    def tick(self, input_u8, in_lane_stride, frames, output_u8, out_lane_stride):
        """Feeds every lane `frames` input frames (lane l from input_u8[l * in_lane_stride:]), reads all lanes' audio in one
        PlayableBatch::Run; lane l's output lands at output_u8[l * out_lane_stride:].  Returns the bytes per lane (uint32 array)."""
        rc = lib().ohhost_live_tick(self._h, input_u8.ctypes.data_as(C.c_void_p), in_lane_stride, frames,
                                    output_u8.ctypes.data_as(C.c_void_p), out_lane_stride, self._out_bytes.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"ohhost_live_tick failed: {rc}")
        return self._out_bytes

    def stats(self):
        v = [C.c_uint64(0) for _ in range(4)]
        f = C.c_uint32(0)
        rc = lib().ohhost_live_stats(self._h, *[C.byref(x) for x in v], C.byref(f))
        if rc != 0:
            raise RuntimeError(f"ohhost_live_stats failed: {rc}")
        return dict(src_calls=int(v[0].value), h2d_bytes=int(v[1].value), d2h_bytes=int(v[2].value), device_allocs=int(v[3].value),
                    filters=int(f.value))

    def close(self):
        if self._h:
            lib().ohhost_live_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


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
