"""ctypes binding of oracle/libohp_oracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (ohpipeline_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libohp_oracle.so")

OK, ERR_ASSERT, ERR_SAMPLE_RATE, ERR_UNSUPPORTED = 0, -1, -2, -3
ENDIAN_LITTLE, ENDIAN_BIG = 1, 2
RAMP_NONE, RAMP_UP, RAMP_DOWN, RAMP_MUTE = 0, 1, 2, 3
RAMP_MAX, RAMP_MIN = 1 << 14, 0
JIFFIES_PER_SEC = 56448000
JIFFIES_PER_MS = 56448
MAX_BYTES = 9216
UNITY_ATTENUATION = 256
FLAG_RAMP, FLAG_SILENCE, FLAG_ZERO_LSB32 = 1, 2, 4

# Same layout as include/ohgpu.h's ohgpu_msg_desc / ohgpu_src_msg_desc (tests/test_capi_loads.py checks).
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
assert MSG_DESC.itemsize == 32 and SRC_MSG_DESC.itemsize == 64


class Ramp(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("direction", C.c_uint32), ("enabled", C.c_uint32)]

    def __repr__(self):
        return f"Ramp({self.start}..{self.end}, dir={self.direction}, enabled={self.enabled})"


class MsgAudio(C.Structure):
    _fields_ = [("size_jiffies", C.c_uint32), ("offset_jiffies", C.c_uint32), ("sample_rate", C.c_uint32),
                ("bit_depth", C.c_uint32), ("channels", C.c_uint32), ("attenuation", C.c_uint32),
                ("is_silence", C.c_uint32), ("ramp", Ramp)]


class Playable(C.Structure):
    _fields_ = [("offset_bytes", C.c_uint32), ("size_bytes", C.c_uint32), ("jiffies", C.c_uint32),
                ("sample_rate", C.c_uint32), ("bit_depth", C.c_uint32), ("channels", C.c_uint32),
                ("attenuation", C.c_uint32), ("is_silence", C.c_uint32), ("ramp", Ramp)]


def build(force=False):
    """Compile the oracle with its own Makefile (gcc -O2)."""
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


class FeedbackModel(C.Structure):
    _fields_ = [("coeffs", C.c_void_p), ("samples", C.c_void_p), ("state_count", C.c_uint32),
                ("data_descale_bits", C.c_uint32), ("coeff_format", C.c_uint32), ("scale_shift_for_output", C.c_int32)]


class OhmAudio(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("msg_type", "msg_bytes", "halt", "lossless", "timestamped", "timestamped2", "resent",
                                          "samples", "frame", "network_timestamp", "media_latency", "media_timestamp")] + \
               [("sample_start", C.c_uint64), ("samples_total", C.c_uint64), ("sample_rate", C.c_uint32), ("bit_rate", C.c_uint32),
                ("volume_offset", C.c_int32), ("bit_depth", C.c_uint32), ("channels", C.c_uint32), ("codec_bytes", C.c_uint32),
                ("codec", C.c_uint8 * 32), ("audio_offset", C.c_uint32), ("audio_bytes", C.c_uint32)]


class SenderFragment(C.Structure):
    _fields_ = [("msg", C.c_uint32), ("playable", Playable)]


class SenderPacket(C.Structure):
    _fields_ = [("first_fragment", C.c_uint32), ("n_fragments", C.c_uint32)]


class OhmDriver(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("bytes_per_sample", C.c_uint32), ("lossless", C.c_uint32), ("latency_ms", C.c_uint32),
                ("latency_ohm", C.c_uint32), ("timestamp_multiplier", C.c_uint32), ("sample_start", C.c_uint64),
                ("samples_total", C.c_uint64), ("frame", C.c_uint32), ("first_frame", C.c_uint32), ("send", C.c_uint32),
                ("stream_header", C.c_uint8 * 88), ("stream_header_bytes", C.c_uint32)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, i32p, f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    vp = C.c_void_p
    sig = {
        "ohp_ramp_table": (C.POINTER(C.c_uint16), []),
        "ohp_construct_pcm": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, vp]),
        "ohp_jiffies_per_sample": (C.c_int, [C.c_uint32]),
        "ohp_jiffies_to_bytes": (C.c_uint32, [u32p, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ohp_jiffies_to_bytes_sample_block": (C.c_uint32, [u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ohp_jiffies_round_down": (C.c_int, [u32p, C.c_uint32]),
        "ohp_jiffies_round_up": (C.c_int, [u32p, C.c_uint32]),
        "ohp_jiffies_round_down_nonzero_sample_block": (None, [u32p, C.c_uint32]),
        "ohp_jiffies_to_songcast_time": (C.c_int, [C.c_uint32, C.c_uint32, u32p]),
        "ohp_ramp_reset": (None, [C.POINTER(Ramp)]),
        "ohp_ramp_set": (C.c_int, [C.POINTER(Ramp), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Ramp), u32p]),
        "ohp_ramp_set_muted": (None, [C.POINTER(Ramp)]),
        "ohp_ramp_validate": (C.c_int, [C.POINTER(Ramp)]),
        "ohp_ramp_split": (C.c_int, [C.POINTER(Ramp), C.c_uint32, C.c_uint32, C.POINTER(Ramp)]),
        "ohp_ramp_median_multiplier": (C.c_uint32, [C.POINTER(Ramp)]),
        "ohp_msg_audio_init_pcm": (C.c_int, [C.POINTER(MsgAudio), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ohp_msg_audio_init_silence": (C.c_int, [C.POINTER(MsgAudio), u32p, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ohp_msg_audio_split": (C.c_int, [C.POINTER(MsgAudio), C.c_uint32, C.POINTER(MsgAudio)]),
        "ohp_msg_audio_set_ramp": (C.c_int, [C.POINTER(MsgAudio), C.c_uint32, u32p, C.c_uint32, C.POINTER(MsgAudio),
                                             C.POINTER(C.c_int), u32p]),
        "ohp_create_playable": (C.c_int, [C.POINTER(MsgAudio), C.POINTER(Playable)]),
        "ohp_playable_split": (C.c_int, [C.POINTER(Playable), C.c_uint32, C.POINTER(Playable), C.POINTER(C.c_int)]),
        "ohp_apply_attenuation": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32]),
        "ohp_ramp_apply": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
        "ohp_playable_read": (C.c_int, [C.POINTER(Playable), vp, vp, C.c_uint32, u32p, C.c_uint32, u32p, u32p]),
        "ohp_flywheel_unpack": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, u32p]),
        "ohp_rampgen_pack": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, u32p]),
        "ohp_sender_pack": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p]),
        "ohp_flac_pack": (C.c_int, [C.POINTER(i32p), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p]),
        "ohp_unpack_s24": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, vp]),
        "ohp_pack_from_s24": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, vp]),
        "ohp_convert_format": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, vp]),
        "ohp_msg_process": (C.c_int, [vp, vp, vp]),
        "ohp_msg_process_batch": (C.c_int, [vp, C.c_size_t, vp, vp]),
        "ohp_src_msg_process": (C.c_int, [vp, vp, vp, vp]),
        "ohp_src_msg_process_batch": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
        "ohp_src_msg_process_batch_steady": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
        "ohp_src_msg_process_f64": (C.c_int, [vp, vp, vp, vp]),
        "ohp_src_new": (vp, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double]),
        "ohp_src_delete": (None, [vp]),
        "ohp_src_L": (C.c_uint32, [vp]), "ohp_src_M": (C.c_uint32, [vp]), "ohp_src_T": (C.c_uint32, [vp]),
        "ohp_src_coef_q28": (i32p, [vp]), "ohp_src_coef_f64": (f64p, [vp]),
        "ohp_src_sum_abs_max": (C.c_int64, [vp]), "ohp_src_f_stop": (C.c_double, [vp]),
        "ohp_src_out_frames": (C.c_uint64, [vp, C.c_uint64]),
        "ohp_burgs_method": (None, [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]),
        "ohp_flywheel_decimation_factor": (C.c_uint32, [C.c_uint32]),
        "ohp_flywheel_coeff_overflow": (C.c_int16, [vp, C.c_uint32, C.c_uint32]),
        "ohp_feedback_init": (None, [C.POINTER(FeedbackModel), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]),
        "ohp_feedback_next_sample": (C.c_int32, [C.POINTER(FeedbackModel)]),
        "ohp_flywheel_ramp": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
        "ohp_ohm_stream_header": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32, vp, C.c_uint32]),
        "ohp_ohm_audio_frame": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                          vp, C.c_uint32, vp, C.c_uint32]),
        "ohp_ohm_audio_parse": (C.c_int, [vp, C.c_uint32, C.POINTER(OhmAudio)]),
        "ohp_sender_packetise": (C.c_int, [C.POINTER(MsgAudio), C.c_uint32, C.c_int, C.POINTER(SenderFragment), C.c_uint32, u32p,
                                           C.POINTER(SenderPacket), C.c_uint32, u32p]),
        "ohp_ohm_driver_init": (None, [C.POINTER(OhmDriver), C.c_uint32]),
        "ohp_ohm_driver_set_track_position": (None, [C.POINTER(OhmDriver), C.c_uint64, C.c_uint64]),
        "ohp_ohm_driver_set_audio_format": (C.c_int, [C.POINTER(OhmDriver), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      vp, C.c_uint32, C.c_uint64]),
        "ohp_ohm_driver_send_audio": (C.c_int, [C.POINTER(OhmDriver), vp, C.c_uint32, C.c_int, vp, C.c_uint32]),
        "ohp_ohm_driver_stream_interrupted": (None, [C.POINTER(OhmDriver)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# ------------------------------------------------------------------ numpy-friendly helpers
def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def ramp_table():
    return np.ctypeslib.as_array(lib().ohp_ramp_table(), shape=(512,)).copy()


def construct_pcm(data, bit_depth, endian):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    dst = np.empty_like(src)
    err = lib().ohp_construct_pcm(_ptr(src), src.size, bit_depth, endian, _ptr(dst))
    return err, dst


def apply_attenuation(data, bit_depth, attenuation):
    buf = np.array(np.frombuffer(bytes(data), dtype=np.uint8))
    err = lib().ohp_apply_attenuation(_ptr(buf), buf.size, bit_depth, attenuation)
    return err, buf


def ramp_apply(data, bit_depth, channels, start, end):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    dst = np.empty_like(src)
    err = lib().ohp_ramp_apply(_ptr(src), src.size, bit_depth, channels, start, end, _ptr(dst))
    return err, dst


def convert_format(data, src_bits, src_endian, dst_bits, dst_endian, zero_lsb32=0):
    src = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    n = src.size // (src_bits // 8)
    dst = np.empty(n * (dst_bits // 8), dtype=np.uint8)
    err = lib().ohp_convert_format(_ptr(src), n, src_bits, src_endian, dst_bits, dst_endian, zero_lsb32, _ptr(dst))
    return err, dst


def playable_read(playable, audio):
    """Returns (err, out_bytes ndarray, [fragment sizes]).  audio is attenuated in place like the reference."""
    out = np.zeros(max(int(playable.size_bytes), 1), dtype=np.uint8)
    frags = np.zeros(max(int(playable.size_bytes), 1) + 4, dtype=np.uint32)
    nf, ob = C.c_uint32(0), C.c_uint32(0)
    aptr = _ptr(audio) if audio is not None else None
    err = lib().ohp_playable_read(C.byref(playable), aptr, _ptr(out), out.size,
                                  frags.ctypes.data_as(C.POINTER(C.c_uint32)), frags.size, C.byref(nf), C.byref(ob))
    return err, out[:ob.value].copy(), [int(v) for v in frags[:nf.value]]


def msg_process_batch(descs, src, dst):
    descs = np.ascontiguousarray(descs)
    assert descs.dtype == MSG_DESC
    return lib().ohp_msg_process_batch(_ptr(descs), descs.size, _ptr(src), _ptr(dst))


class Src:
    """The oracle's resampler design + exact integer model (own specification, parity unpinned)."""

    def __init__(self, rate_in, rate_out, taps_per_phase=32, beta=9.0, f_pass=20000.0):
        self.h = lib().ohp_src_new(rate_in, rate_out, taps_per_phase, beta, f_pass)
        if not self.h:
            raise ValueError("ohp_src_design failed")
        L = lib()
        self.L, self.M, self.T = L.ohp_src_L(self.h), L.ohp_src_M(self.h), L.ohp_src_T(self.h)
        n = self.L * self.T
        self.coef_q28 = np.ctypeslib.as_array(L.ohp_src_coef_q28(self.h), shape=(n,)).copy()
        self.coef_f64 = np.ctypeslib.as_array(L.ohp_src_coef_f64(self.h), shape=(n,)).copy()
        self.sum_abs_max = L.ohp_src_sum_abs_max(self.h)
        self.f_stop = L.ohp_src_f_stop(self.h)

    def out_frames(self, in_frames):
        return int(lib().ohp_src_out_frames(self.h, in_frames))

    def process_batch(self, descs, src, dst):
        descs = np.ascontiguousarray(descs)
        assert descs.dtype == SRC_MSG_DESC
        return lib().ohp_src_msg_process_batch(self.h, _ptr(descs), descs.size, _ptr(src), _ptr(dst))

    def process_f64(self, desc, src):
        desc = np.ascontiguousarray(desc).reshape(1)
        y = np.zeros(int(desc["n_frames"][0]) * int(desc["channels"][0]), dtype=np.float64)
        err = lib().ohp_src_msg_process_f64(self.h, _ptr(desc), _ptr(src), _ptr(y))
        return err, y

    def __del__(self):
        try:
            if self.h:
                lib().ohp_src_delete(self.h)
                self.h = None
        except Exception:
            pass


# ------------------------------------------------------------------ Songcast sender (oracle/ohp_songcast.h)
def ohm_stream_header(samples_total, sample_rate, bit_rate, volume_offset, bit_depth, channels, codec=b""):
    buf = np.zeros(88, dtype=np.uint8)
    c = np.frombuffer(bytes(codec) or b"\0", dtype=np.uint8).copy()
    n = lib().ohp_ohm_stream_header(_ptr(buf), buf.size, samples_total, sample_rate, bit_rate, volume_offset, bit_depth,
                                    channels, _ptr(c), len(codec))
    return n, buf[:max(n, 0)].copy()


def ohm_audio_frame(flags, samples, frame, network_timestamp, media_latency, sample_start, stream_header, audio):
    out = np.zeros(8192, dtype=np.uint8)
    sh = np.ascontiguousarray(np.frombuffer(bytes(stream_header), dtype=np.uint8))
    au = np.ascontiguousarray(np.frombuffer(bytes(audio) or b"\0", dtype=np.uint8))
    n = lib().ohp_ohm_audio_frame(_ptr(out), out.size, flags, samples, frame, network_timestamp, media_latency, sample_start,
                                  _ptr(sh), sh.size, _ptr(au), len(bytes(audio)))
    return n, out[:max(n, 0)].copy()


def ohm_audio_parse(datagram):
    d = np.ascontiguousarray(np.frombuffer(bytes(datagram), dtype=np.uint8))
    out = OhmAudio()
    err = lib().ohp_ohm_audio_parse(_ptr(d), d.size, C.byref(out))
    return err, out


def sender_packetise(msgs, flush=True):
    """msgs: list of MsgAudio.  Returns (err, [SenderFragment], [SenderPacket])."""
    arr = (MsgAudio * max(len(msgs), 1))(*msgs)
    cap = 16 * len(msgs) + 64
    frags, packs = (SenderFragment * cap)(), (SenderPacket * cap)()
    nf, npk = C.c_uint32(0), C.c_uint32(0)
    err = lib().ohp_sender_packetise(arr, len(msgs), 1 if flush else 0, frags, cap, C.byref(nf), packs, cap, C.byref(npk))
    return err, list(frags[:nf.value]), list(packs[:npk.value])


def songcast_datagrams(driver, msgs, audio, flush=True, halt_last=False):
    """The reference's Sender + OhmSenderDriver over messages msgs[i] (MsgAudio) whose DecodedAudio is audio[i] (uint8 array,
    None for silence): packetise, read every fragment's playable (attenuation, ramp), Sender pack, frame.  Returns the list
    of datagrams (uint8 arrays) in send order."""
    err, frags, packs = sender_packetise(msgs, flush)
    assert err == 0, err
    out = []
    for k, pk in enumerate(packs):
        payload = []
        for f in frags[pk.first_fragment:pk.first_fragment + pk.n_fragments]:
            a = None if audio[f.msg] is None else np.array(audio[f.msg], dtype=np.uint8)    # Read attenuates in place: work on a copy
            e, pcm, _ = playable_read(f.playable, a)
            assert e == 0, e
            if pcm.size == 0:
                continue
            packed = np.zeros(pcm.size, dtype=np.uint8)
            nb = C.c_uint32(0)
            assert lib().ohp_sender_pack(_ptr(pcm), pcm.size, f.playable.channels, f.playable.bit_depth // 8, _ptr(packed), C.byref(nb)) == 0
            payload.append(packed[:nb.value])
        au = np.concatenate(payload) if payload else np.zeros(0, dtype=np.uint8)
        buf = np.zeros(8192, dtype=np.uint8)
        aptr = np.ascontiguousarray(au if au.size else np.zeros(1, dtype=np.uint8))
        n = lib().ohp_ohm_driver_send_audio(C.byref(driver), _ptr(aptr), au.size, 1 if (halt_last and k == len(packs) - 1) else 0,
                                            _ptr(buf), buf.size)
        assert n >= 0, n
        if n > 0:
            out.append(buf[:n].copy())
    return out
