"""ctypes binding of oracle/_ref/libflac_ref.so -- the reference's vendored libFLAC 1.2.1 built from its own sources
(oracle/Makefile, target `ref`).  TEST INFRASTRUCTURE ONLY: the decoder is what sits in front of the reference's FLAC
packer (OpenHome/Media/Codec/Flac.cpp hands it the same callbacks), the encoder is used to make test streams.

decode() returns the frames exactly as CodecFlac::CallbackWrite receives them: per frame (blocksize, channels,
bits_per_sample, sample_rate, planes) with planes = int32[channels][blocksize], host endian."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "oracle", "_ref", "libflac_ref.so")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libflac_ref.so is missing: `make -C oracle ref` builds it where /root/reference exists")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        for name, res, args in [
            ("FLAC__stream_encoder_new", vp, []), ("FLAC__stream_encoder_delete", None, [vp]),
            ("FLAC__stream_encoder_set_channels", C.c_int, [vp, C.c_uint]), ("FLAC__stream_encoder_set_bits_per_sample", C.c_int, [vp, C.c_uint]),
            ("FLAC__stream_encoder_set_sample_rate", C.c_int, [vp, C.c_uint]), ("FLAC__stream_encoder_set_compression_level", C.c_int, [vp, C.c_uint]),
            ("FLAC__stream_encoder_set_blocksize", C.c_int, [vp, C.c_uint]), ("FLAC__stream_encoder_set_verify", C.c_int, [vp, C.c_int]),
            ("FLAC__stream_encoder_set_total_samples_estimate", C.c_int, [vp, C.c_uint64]),
            ("FLAC__stream_encoder_init_stream", C.c_int, [vp, vp, vp, vp, vp, vp]),
            ("FLAC__stream_encoder_process_interleaved", C.c_int, [vp, vp, C.c_uint]), ("FLAC__stream_encoder_finish", C.c_int, [vp]),
            ("FLAC__stream_decoder_new", vp, []), ("FLAC__stream_decoder_delete", None, [vp]),
            ("FLAC__stream_decoder_set_md5_checking", C.c_int, [vp, C.c_int]),
            ("FLAC__stream_decoder_init_stream", C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
            ("FLAC__stream_decoder_process_until_end_of_stream", C.c_int, [vp]), ("FLAC__stream_decoder_finish", C.c_int, [vp]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


_ENC_WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint, C.c_uint, C.c_void_p)
_ENC_SEEK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
_ENC_TELL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
_DEC_READ = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t), C.c_void_p)
_DEC_WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
_DEC_META = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)
_DEC_ERROR = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


def encode(pcm, bits, rate, blocksize=4096, level=5):
    """pcm: int32 array [frames, channels] of right-justified samples.  Returns the FLAC stream (bytes), STREAMINFO complete
    (the encoder seeks back to fill in the MD5 of the audio and the sample count)."""
    L = lib()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    frames, ch = pcm.shape
    out = bytearray()
    pos = [0]

    def write(_e, buf, nbytes, _samples, _frame, _c):
        data = C.string_at(buf, nbytes)
        out[pos[0]:pos[0] + nbytes] = data
        pos[0] += nbytes
        return 0

    def seek(_e, offset, _c):
        pos[0] = int(offset)
        return 0

    def tell(_e, poff, _c):
        poff[0] = pos[0]
        return 0

    cbs = (_ENC_WRITE(write), _ENC_SEEK(seek), _ENC_TELL(tell))
    e = L.FLAC__stream_encoder_new()
    assert e
    try:
        ok = L.FLAC__stream_encoder_set_channels(e, ch) and L.FLAC__stream_encoder_set_bits_per_sample(e, bits) and \
            L.FLAC__stream_encoder_set_sample_rate(e, rate) and L.FLAC__stream_encoder_set_compression_level(e, level) and \
            L.FLAC__stream_encoder_set_blocksize(e, blocksize) and L.FLAC__stream_encoder_set_verify(e, 1) and \
            L.FLAC__stream_encoder_set_total_samples_estimate(e, frames)
        assert ok
        st = L.FLAC__stream_encoder_init_stream(e, C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), C.cast(cbs[2], C.c_void_p), None, None)
        assert st == 0, f"FLAC__stream_encoder_init_stream -> {st}"
        assert L.FLAC__stream_encoder_process_interleaved(e, pcm.ctypes.data_as(C.c_void_p), frames)
        assert L.FLAC__stream_encoder_finish(e)
    finally:
        L.FLAC__stream_encoder_delete(e)
    return bytes(out)


def decode(stream, read_chunk=4096):
    """Returns (frames, md5_ok): frames = [(blocksize, channels, bits, rate, planes int32[channels][blocksize])], md5_ok = the
    decoder's own check of the decoded audio against the MD5 in STREAMINFO (FLAC__stream_decoder_finish)."""
    L = lib()
    data = bytes(stream)
    pos = [0]
    frames, errors = [], []

    def read(_d, buf, pbytes, _c):
        want = min(int(pbytes[0]), read_chunk)
        n = min(want, len(data) - pos[0])
        if n <= 0:
            pbytes[0] = 0
            return 1                                         # FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM
        C.memmove(buf, data[pos[0]:pos[0] + n], n)
        pos[0] += n
        pbytes[0] = n
        return 0

    def write(_d, frame, buffers, _c):
        # FLAC__FrameHeader starts: unsigned blocksize, sample_rate, channels; enum channel_assignment; unsigned bits_per_sample
        blocksize, rate, ch, _assign, bits = frame[0], frame[1], frame[2], frame[3], frame[4]
        planes = np.empty((ch, blocksize), dtype=np.int32)
        for c in range(ch):
            planes[c] = np.ctypeslib.as_array(buffers[c], shape=(blocksize,))
        frames.append((int(blocksize), int(ch), int(bits), int(rate), planes))
        return 0

    def meta(_d, _m, _c):
        pass

    def error(_d, status, _c):
        errors.append(int(status))

    cbs = (_DEC_READ(read), _DEC_WRITE(write), _DEC_META(meta), _DEC_ERROR(error))
    d = L.FLAC__stream_decoder_new()
    assert d
    try:
        assert L.FLAC__stream_decoder_set_md5_checking(d, 1)
        st = L.FLAC__stream_decoder_init_stream(d, C.cast(cbs[0], C.c_void_p), None, None, None, None, C.cast(cbs[1], C.c_void_p),
                                                C.cast(cbs[2], C.c_void_p), C.cast(cbs[3], C.c_void_p), None)
        assert st == 0, f"FLAC__stream_decoder_init_stream -> {st}"
        ok = L.FLAC__stream_decoder_process_until_end_of_stream(d)
        md5_ok = bool(L.FLAC__stream_decoder_finish(d))
        assert ok and not errors, (ok, errors)
    finally:
        L.FLAC__stream_decoder_delete(d)
    return frames, md5_ok


def streaminfo(stream):
    """The STREAMINFO block of a FLAC stream (format spec: 'fLaC', block header, 34 bytes): a few fields and the MD5."""
    s = bytes(stream)
    assert s[:4] == b"fLaC" and (s[4] & 0x7f) == 0
    b = s[8:8 + 34]
    v = int.from_bytes(b[10:18], "big")
    return dict(min_block=int.from_bytes(b[0:2], "big"), max_block=int.from_bytes(b[2:4], "big"), rate=v >> 44,
                channels=((v >> 41) & 7) + 1, bits=((v >> 36) & 31) + 1, total_samples=v & ((1 << 36) - 1), md5=b[18:34])
