"""ctypes binding of libohgpu.so -- exactly the entry points include/ohgpu.h declares.

There is no CPU fallback: if the library is missing this module raises, and if there is no GPU
ohgpu_init() returns OHGPU_ERR_NO_DEVICE which `Context` turns into an exception.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libohgpu.so")

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_NO_DEVICE, ERR_NOMEM, ERR_BOUNDS, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
ENDIAN_LITTLE, ENDIAN_BIG = 1, 2
RAMP_MAX = 16384
UNITY_ATTENUATION = 256
FLAG_RAMP, FLAG_SILENCE, FLAG_ZERO_LSB32, FLAG_SRC_PLANAR32 = 1, 2, 4, 8

# numpy views of ohgpu_msg_desc (32 B) and ohgpu_src_msg_desc (64 B)
MSG_DESC = np.dtype([
    ("src_offset", "<u8"), ("dst_offset", "<u8"), ("n_frames", "<u4"),
    ("ramp_start", "<u2"), ("ramp_end", "<u2"), ("attenuation", "<u2"),
    ("channels", "u1"), ("src_bits", "u1"), ("src_endian", "u1"),
    ("dst_bits", "u1"), ("dst_endian", "u1"), ("flags", "u1")], align=False)
SRC_MSG_DESC = np.dtype([
    ("src_offset", "<u8"), ("src_frame0", "<u8"), ("src_frames", "<u8"), ("out_frame0", "<u8"),
    ("dst_offset", "<u8"), ("n_frames", "<u4"),
    ("ramp_start", "<u2"), ("ramp_end", "<u2"), ("attenuation", "<u2"),
    ("channels", "u1"), ("src_bits", "u1"), ("src_endian", "u1"),
    ("dst_bits", "u1"), ("dst_endian", "u1"), ("flags", "u1"), ("src_plane_stride", "<u8")], align=False)

FMT_UNPACK_PLANAR, FMT_SENDER_PACK, FMT_FLAC_PACK = 1, 2, 3
FMT_DESC = np.dtype([
    ("src_offset", "<u8"), ("dst_offset", "<u8"), ("src_plane_stride", "<u8"), ("dst_plane_stride", "<u8"),
    ("n_frames", "<u4"), ("kind", "u1"), ("channels", "u1"), ("src_bits", "u1"), ("dst_bits", "u1"),
    ("reserved", "u1", (8,))], align=False)

FLYWHEEL_DESC = np.dtype([
    ("src_offset", "<u8"), ("channel_bytes", "<u8"), ("dst_offset", "<u8"), ("in_samples", "<u4"),
    ("out_frames", "<u4"), ("block_frames", "<u4"), ("sample_rate", "<u4"), ("channels", "<u4"),
    ("reserved", "<u4")], align=False)

OHM_FLAG_HALT, OHM_FLAG_LOSSLESS, OHM_FLAG_TIMESTAMPED, OHM_FLAG_RESENT = 1, 2, 4, 8
OHM_STREAM = np.dtype([
    ("samples_total", "<u8"), ("sample_rate", "<u4"), ("bit_rate", "<u4"), ("volume_offset", "<i2"),
    ("src_channels", "u1"), ("src_bits", "u1"), ("codec_bytes", "u1"), ("codec", "u1", (29,)),
    ("src_endian", "u1"), ("reserved", "u1", (13,))], align=False)
OHM_FRAGMENT = np.dtype([
    ("src_offset", "<u8"), ("n_frames", "<u4"), ("ramp_start", "<u2"), ("ramp_end", "<u2"), ("attenuation", "<u2"),
    ("flags", "u1"), ("reserved", "u1", (5,))], align=False)
OHM_FRAME_DESC = np.dtype([
    ("dst_offset", "<u8"), ("sample_start", "<u8"), ("stream", "<u4"), ("frame", "<u4"), ("network_timestamp", "<u4"),
    ("media_latency", "<u4"), ("media_timestamp", "<u4"), ("first_fragment", "<u4"), ("n_fragments", "<u2"),
    ("flags", "u1"), ("reserved", "u1", (5,))], align=False)
assert OHM_STREAM.itemsize == 64 and OHM_FRAGMENT.itemsize == 24 and OHM_FRAME_DESC.itemsize == 48

# every symbol of include/ohgpu.h: name -> (restype, argtypes)
_vp, _vpp = C.c_void_p, C.POINTER(C.c_void_p)
_u64p = C.POINTER(C.c_uint64)
SYMBOLS = {
    "ohgpu_abi_version": (C.c_int, []),
    "ohgpu_last_error": (C.c_char_p, []),
    "ohgpu_device_count": (C.c_int, []),
    "ohgpu_init": (C.c_int, [C.c_int, _vpp]),
    "ohgpu_shutdown": (C.c_int, [_vp]),
    "ohgpu_device_name": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "ohgpu_device_pci_bus_id": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "ohgpu_malloc": (C.c_int, [_vp, C.c_size_t, _vpp]),
    "ohgpu_free": (C.c_int, [_vp, _vp]),
    "ohgpu_malloc_host": (C.c_int, [_vp, C.c_size_t, _vpp]),
    "ohgpu_free_host": (C.c_int, [_vp, _vp]),
    "ohgpu_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "ohgpu_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "ohgpu_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp]),
    "ohgpu_stream_create": (C.c_int, [_vp, _vpp]),
    "ohgpu_stream_destroy": (C.c_int, [_vp, _vp]),
    "ohgpu_stream_sync": (C.c_int, [_vp, _vp]),
    "ohgpu_event_create": (C.c_int, [_vp, _vpp]),
    "ohgpu_event_destroy": (C.c_int, [_vp, _vp]),
    "ohgpu_event_record": (C.c_int, [_vp, _vp, _vp]),
    "ohgpu_stream_wait_event": (C.c_int, [_vp, _vp, _vp]),
    "ohgpu_event_elapsed_ms": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "ohgpu_ramp_table": (C.c_int, [C.POINTER(C.c_uint16)]),
    "ohgpu_pcm_batch_create": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vpp]),
    "ohgpu_pcm_batch_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ohgpu_batch_destroy": (C.c_int, [_vp, _vp]),
    "ohgpu_batch_info": (C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p, _u64p]),
    "ohgpu_pcm_process_host": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint64, _vp, C.c_uint64]),
    "ohgpu_fmt_batch_create": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vpp]),
    "ohgpu_fmt_batch_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ohgpu_flywheel_batch_create": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vpp]),
    "ohgpu_flywheel_batch_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ohgpu_flywheel_process_host": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint64, _vp, C.c_uint64]),
    "ohgpu_ohm_frame_layout": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ohgpu_ohm_batch_create": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vpp]),
    "ohgpu_ohm_batch_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ohgpu_ohm_process_host": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_uint64, _vp, C.c_uint64]),
    "ohgpu_src_design": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, _vp, C.c_size_t,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ohgpu_src_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vpp]),
    "ohgpu_src_destroy": (C.c_int, [_vp, _vp]),
    "ohgpu_src_out_frames": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint64]),
    "ohgpu_src_mfma_tables": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp, C.c_size_t, _vp, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "ohgpu_src_mfma_halfband_tables": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]),
    "ohgpu_src_batch_create": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint64, C.c_uint64, _vpp]),
    "ohgpu_src_batch_run": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ohgpu_src_batch_run_timed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ohgpu_src_batch_plan": (C.c_int, [_vp, _u64p, _u64p]),
    "ohgpu_src_batch_units": (C.c_int, [_vp, _u64p, _u64p]),
    "ohgpu_set_plan_threads": (C.c_int, [C.c_int]),
    "ohgpu_src_batch_block": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ohgpu_src_batch_advance": (C.c_int, [_vp, _vp, C.c_uint64]),
    "ohgpu_src_batch_set_ramps": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t]),
    "ohgpu_src_plan_digest": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int,
                                        _vp, C.c_int, _u64p, _u64p, _u64p, C.POINTER(C.c_int)]),
    "ohgpu_src_batch_kernel_name": (C.c_int, [_vp, _vp, C.c_char_p, C.c_size_t]),
    "ohgpu_src_batch_occupancy": (C.c_int, [_vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "ohgpu_measure_shader_clock": (C.c_int, [_vp, _vp, C.POINTER(C.c_double)]),
    "ohgpu_device_allocations": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "ohgpu_src_process_host": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_uint64, _vp, C.c_uint64]),
    "ohgpu_host_transfer_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ohgpu_set_kernel_variant": (C.c_int, [_vp, C.c_int]),
}


class OhGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ohgpu error {code}: {message}")
        self.code = code


_lib = None


def lib():
    """Loads libohgpu.so (build it first with `python -m ohpipeline_amd.build` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP extension first (python ohpipeline_amd/build.py); "
                          "there is no CPU fallback for the product path")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)          # AttributeError here = the library does not export what ohgpu.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error():
    return lib().ohgpu_last_error().decode("utf-8", "replace")


def check(code):
    if code != OK:
        raise OhGpuError(code, last_error())
    return code


def ramp_table():
    out = (C.c_uint16 * 512)()
    check(lib().ohgpu_ramp_table(out))
    return np.frombuffer(out, dtype=np.uint16).copy()


def device_count():
    """GPUs visible to the process (ohgpu_device_count); 0 without one."""
    n = lib().ohgpu_device_count()
    return max(int(n), 0)


MF_STEP = np.dtype([("aoff", "<u4", (16,)), ("b0", "<u4", (16,)), ("b1", "<u4", (16,)), ("b2", "<i4", (16,)), ("kc", "<u4"), ("pad", "<u4", (7,))])
assert MF_STEP.itemsize == 288


def set_plan_threads(threads):
    check(lib().ohgpu_set_plan_threads(int(threads)))


def src_plan_digest(L, M, T, descs, src_arena_bytes, dst_arena_bytes, kernel_variant=0, coef_q28=None, num_cus=0):
    """The plan ohgpu_src_batch_create would make, hashed on the host (no device): {digest, units, generic_pieces, kernel}.
    coef_q28: the filter's coefficients (the plan then is exactly a real batch's: half-band form, tables, gain); None = a plain
    polyphase filter of sane gain.  num_cus: the device's CU count (0 = 256)."""
    d = np.ascontiguousarray(descs)
    coef = None if coef_q28 is None else np.ascontiguousarray(coef_q28, dtype=np.int32)
    h, u, p, k = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    check(lib().ohgpu_src_plan_digest(L, M, T, d.ctypes.data_as(C.c_void_p), d.size, src_arena_bytes, dst_arena_bytes, kernel_variant,
                                      None if coef is None else coef.ctypes.data_as(C.c_void_p), num_cus,
                                      C.byref(h), C.byref(u), C.byref(p), C.byref(k)))
    return {"digest": int(h.value), "units": int(u.value), "generic_pieces": int(p.value), "kernel": int(k.value)}


def src_mfma_tables(L, M, T, coef_q28, max_blocks_per_row=8):
    """(digits [4][L][96] int8, steps MF_STEP[], outputs per block) -- the matrix-pipe resampler kernel's host tables (no device)."""
    coef = np.ascontiguousarray(coef_q28, dtype=np.int32)
    nb, ns, lb = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
    check(lib().ohgpu_src_mfma_tables(L, M, T, coef.ctypes.data, max_blocks_per_row, None, 0, None, 0, C.byref(nb), C.byref(ns), C.byref(lb)))
    dig = np.zeros(nb.value, dtype=np.int8)
    steps = np.zeros(ns.value // MF_STEP.itemsize, dtype=MF_STEP)
    check(lib().ohgpu_src_mfma_tables(L, M, T, coef.ctypes.data, max_blocks_per_row, dig.ctypes.data, dig.nbytes, steps.ctypes.data, steps.nbytes,
                                      C.byref(nb), C.byref(ns), C.byref(lb)))
    return dig.reshape(4, L, 96), steps, lb.value


def src_mfma_halfband_tables(coef_q28):
    """(image [4 digits][4 K groups][16 outputs][16] int8, bias, outputs per block) -- the half-band form's host tables (no device)."""
    coef = np.ascontiguousarray(coef_q28, dtype=np.int32)
    image = np.zeros(4096, dtype=np.int8)
    bias, lb = C.c_int64(0), C.c_uint32(0)
    check(lib().ohgpu_src_mfma_halfband_tables(coef.ctypes.data, image.ctypes.data, C.byref(bias), C.byref(lb)))
    return image.reshape(4, 4, 16, 16), int(bias.value), int(lb.value)


def src_design(rate_in, rate_out, taps_per_phase=32, beta=9.0, f_pass=20000.0):
    """Returns (L, M, coef_q28[L*T]) from the library's own host-side filter design."""
    L_, M_ = C.c_uint32(0), C.c_uint32(0)
    check(lib().ohgpu_src_design(rate_in, rate_out, taps_per_phase, beta, f_pass, None, 0, C.byref(L_), C.byref(M_)))
    coef = np.zeros(L_.value * taps_per_phase, dtype=np.int32)
    check(lib().ohgpu_src_design(rate_in, rate_out, taps_per_phase, beta, f_pass, coef.ctypes.data_as(C.c_void_p),
                                 coef.size, C.byref(L_), C.byref(M_)))
    return L_.value, M_.value, coef


class Context:
    """One GPU context (ohgpu_ctx).  Owns device allocations made through it."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().ohgpu_init(device, C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def name(self):
        buf = C.create_string_buffer(128)
        check(lib().ohgpu_device_name(self._h, buf, 128))
        return buf.value.decode()

    def close(self):
        if self._h:
            lib().ohgpu_shutdown(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- memory
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().ohgpu_malloc(self._h, nbytes, C.byref(p)))
        return p

    def free(self, dptr):
        check(lib().ohgpu_free(self._h, dptr))

    def stream_create(self):
        p = C.c_void_p()
        check(lib().ohgpu_stream_create(self._h, C.byref(p)))
        return p

    def stream_destroy(self, stream):
        check(lib().ohgpu_stream_destroy(self._h, stream))

    def malloc_host(self, nbytes):
        """Pinned host memory as a uint8 array (free with free_host(array))."""
        p = C.c_void_p()
        check(lib().ohgpu_malloc_host(self._h, max(nbytes, 1), C.byref(p)))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]
        return a

    def free_host(self, array):
        check(lib().ohgpu_free_host(self._h, C.c_void_p(array.ctypes.data)))

    def copy_h2d(self, dptr, array, stream=None):
        check(lib().ohgpu_memcpy_h2d(self._h, dptr, array.ctypes.data_as(C.c_void_p), array.nbytes, stream))

    def copy_d2h(self, array, dptr, stream=None):
        check(lib().ohgpu_memcpy_d2h(self._h, array.ctypes.data_as(C.c_void_p), dptr, array.nbytes, stream))

    def upload(self, array, stream=None):
        a = np.ascontiguousarray(array)
        p = self.malloc(max(a.nbytes, 1))
        check(lib().ohgpu_memcpy_h2d(self._h, p, a.ctypes.data_as(C.c_void_p), a.nbytes, stream))
        self.sync(stream)
        return p

    def download(self, dptr, nbytes, stream=None):
        out = np.empty(nbytes, dtype=np.uint8)
        check(lib().ohgpu_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), dptr, nbytes, stream))
        self.sync(stream)
        return out

    def memset(self, dptr, value, nbytes, stream=None):
        check(lib().ohgpu_memset(self._h, dptr, value, nbytes, stream))

    def sync(self, stream=None):
        check(lib().ohgpu_stream_sync(self._h, stream))

    def set_kernel_variant(self, v):
        check(lib().ohgpu_set_kernel_variant(self._h, v))

    # ---- events (HIP events on the launch stream)
    def event(self):
        e = C.c_void_p()
        check(lib().ohgpu_event_create(self._h, C.byref(e)))
        return e

    def event_destroy(self, event):
        check(lib().ohgpu_event_destroy(self._h, event))

    def record(self, event, stream=None):
        check(lib().ohgpu_event_record(self._h, event, stream))

    def wait_event(self, stream, event):
        check(lib().ohgpu_stream_wait_event(self._h, stream, event))

    def elapsed_ms(self, start, stop):
        ms = C.c_float(0)
        check(lib().ohgpu_event_elapsed_ms(self._h, start, stop, C.byref(ms)))
        return ms.value

    # ---- batches
    def pcm_batch(self, descs, src_arena_bytes, dst_arena_bytes):
        d = np.ascontiguousarray(descs)
        assert d.dtype == MSG_DESC
        b = C.c_void_p()
        check(lib().ohgpu_pcm_batch_create(self._h, d.ctypes.data_as(C.c_void_p), d.size, src_arena_bytes,
                                           dst_arena_bytes, C.byref(b)))
        return b

    def pcm_run(self, batch, d_src, d_dst, stream=None):
        check(lib().ohgpu_pcm_batch_run(self._h, batch, d_src, d_dst, stream))

    def batch_destroy(self, batch):
        check(lib().ohgpu_batch_destroy(self._h, batch))

    def batch_info(self, batch):
        v = [C.c_uint64(0) for _ in range(5)]
        check(lib().ohgpu_batch_info(batch, *[C.byref(x) for x in v]))
        keys = ("n_msgs", "in_frames", "out_frames", "src_bytes_touched", "dst_bytes_written")
        return dict(zip(keys, (int(x.value) for x in v)))

    def pcm_process_host(self, descs, src, dst):
        d = np.ascontiguousarray(descs)
        assert d.dtype == MSG_DESC
        check(lib().ohgpu_pcm_process_host(self._h, d.ctypes.data_as(C.c_void_p), d.size,
                                           src.ctypes.data_as(C.c_void_p), src.nbytes,
                                           dst.ctypes.data_as(C.c_void_p), dst.nbytes))
        return dst

    def fmt_batch(self, descs, src_arena_bytes, dst_arena_bytes):
        d = np.ascontiguousarray(descs)
        assert d.dtype == FMT_DESC
        b = C.c_void_p()
        check(lib().ohgpu_fmt_batch_create(self._h, d.ctypes.data_as(C.c_void_p), d.size, src_arena_bytes,
                                           dst_arena_bytes, C.byref(b)))
        return b

    def fmt_run(self, batch, d_src, d_dst, stream=None):
        check(lib().ohgpu_fmt_batch_run(self._h, batch, d_src, d_dst, stream))

    def flywheel_batch(self, descs, src_arena_bytes, dst_arena_bytes):
        d = np.ascontiguousarray(descs)
        assert d.dtype == FLYWHEEL_DESC
        b = C.c_void_p()
        check(lib().ohgpu_flywheel_batch_create(self._h, d.ctypes.data_as(C.c_void_p), d.size, src_arena_bytes,
                                                dst_arena_bytes, C.byref(b)))
        return b

    def flywheel_run(self, batch, d_src, d_dst, stream=None):
        check(lib().ohgpu_flywheel_batch_run(self._h, batch, d_src, d_dst, stream))

    def ohm_batch(self, streams, frames, fragments, src_arena_bytes, dst_arena_bytes):
        st, fr, fg = np.ascontiguousarray(streams), np.ascontiguousarray(frames), np.ascontiguousarray(fragments)
        assert st.dtype == OHM_STREAM and fr.dtype == OHM_FRAME_DESC and fg.dtype == OHM_FRAGMENT
        b = C.c_void_p()
        check(lib().ohgpu_ohm_batch_create(self._h, st.ctypes.data_as(C.c_void_p), st.size, fr.ctypes.data_as(C.c_void_p), fr.size,
                                           fg.ctypes.data_as(C.c_void_p), fg.size, src_arena_bytes, dst_arena_bytes, C.byref(b)))
        return b

    def ohm_run(self, batch, d_src, d_dst, stream=None):
        check(lib().ohgpu_ohm_batch_run(self._h, batch, d_src, d_dst, stream))

    def src_create(self, L, M, T, coef_q28):
        c = np.ascontiguousarray(coef_q28, dtype=np.int32)
        s = C.c_void_p()
        check(lib().ohgpu_src_create(self._h, L, M, T, c.ctypes.data_as(C.c_void_p), C.byref(s)))
        return s

    def src_destroy(self, src):
        check(lib().ohgpu_src_destroy(self._h, src))

    def src_batch(self, src, descs, src_arena_bytes, dst_arena_bytes):
        d = np.ascontiguousarray(descs)
        assert d.dtype == SRC_MSG_DESC
        b = C.c_void_p()
        check(lib().ohgpu_src_batch_create(self._h, src, d.ctypes.data_as(C.c_void_p), d.size, src_arena_bytes,
                                           dst_arena_bytes, C.byref(b)))
        return b

    def src_batch_block(self, batch):
        lo, mi = C.c_uint32(0), C.c_uint32(0)
        check(lib().ohgpu_src_batch_block(batch, C.byref(lo), C.byref(mi)))
        return int(lo.value), int(mi.value)

    def src_batch_advance(self, batch, blocks):
        check(lib().ohgpu_src_batch_advance(self._h, batch, blocks))

    def src_batch_set_ramps(self, batch, ramp_start, ramp_end):
        a = np.ascontiguousarray(ramp_start, dtype=np.uint16)
        e = np.ascontiguousarray(ramp_end, dtype=np.uint16)
        assert a.size == e.size
        check(lib().ohgpu_src_batch_set_ramps(self._h, batch, a.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), a.size))

    def src_process_host(self, src, descs, src_bytes, dst_bytes_array):
        """ohgpu_src_process_host: host buffers in, host buffers out (validation, upload, launch, download, sync in one call)."""
        d = np.ascontiguousarray(descs)
        assert d.dtype == SRC_MSG_DESC
        check(lib().ohgpu_src_process_host(self._h, src, d.ctypes.data_as(C.c_void_p), d.size,
                                           src_bytes.ctypes.data_as(C.c_void_p), src_bytes.nbytes,
                                           dst_bytes_array.ctypes.data_as(C.c_void_p), dst_bytes_array.nbytes))
        return dst_bytes_array

    def src_plan(self, batch):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().ohgpu_src_batch_plan(batch, C.byref(a), C.byref(b)))
        return {"block_kernel_out_frames": int(a.value), "generic_pieces": int(b.value)}

    def src_units(self, batch):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().ohgpu_src_batch_units(batch, C.byref(a), C.byref(b)))
        return {"units": int(a.value), "long_units": int(b.value)}

    def src_kernel_name(self, batch):
        buf = C.create_string_buffer(256)
        check(lib().ohgpu_src_batch_kernel_name(self._h, batch, buf, 256))
        return buf.value.decode()

    def src_occupancy(self, batch):
        g, want, lds = C.c_int(0), C.c_int(0), C.c_uint32(0)
        check(lib().ohgpu_src_batch_occupancy(self._h, batch, C.byref(g), C.byref(want), C.byref(lds)))
        return {"workgroups_per_cu": int(g.value), "designed_for": int(want.value), "lds_bytes": int(lds.value)}

    def device_allocations(self):
        n = C.c_uint64(0)
        check(lib().ohgpu_device_allocations(self._h, C.byref(n)))
        return int(n.value)

    def pci_bus_id(self):
        buf = C.create_string_buffer(64)
        check(lib().ohgpu_device_pci_bus_id(self._h, buf, 64))
        return buf.value.decode()

    def host_transfer_stats(self):
        """What the *_process_host calls of this context moved so far (ohgpu_host_transfer_stats)."""
        v = [C.c_uint64(0) for _ in range(4)]
        check(lib().ohgpu_host_transfer_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("calls", "src_calls", "h2d_bytes", "d2h_bytes"), (int(x.value) for x in v)))

    def shader_clock_mhz(self, stream=None):
        mhz = C.c_double(0)
        check(lib().ohgpu_measure_shader_clock(self._h, stream, C.byref(mhz)))
        return float(mhz.value)

    def src_run(self, batch, d_src, d_dst, stream=None, events=None):
        """`events` = (start, stop) of ctx.event(): they bracket the batch's device work (on the dispatch itself where the batch is one launch)."""
        if events is not None:
            check(lib().ohgpu_src_batch_run_timed(self._h, batch, d_src, d_dst, stream, events[0], events[1]))
        else:
            check(lib().ohgpu_src_batch_run(self._h, batch, d_src, d_dst, stream))
