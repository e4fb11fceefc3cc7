"""oracle/_ref: the reference's vendored libFLAC 1.2.1, built from its own sources (oracle/Makefile `ref`), and the oracle's
FLAC packer (row a14) against it.  What pins what:
  * each fixture's STREAMINFO holds the MD5 of the audio it encodes (FLAC format); the reference decoder checks its own
    output against it, and so does hashlib here -- the decoder's planes are therefore known-good input for a14;
  * FLAC is lossless, so packing the decoded planes the way CodecFlac::CallbackWrite does (Flac.cpp:379-417) must give
    exactly the packed big-endian form of the PCM that was encoded: a known answer for ohp_flac_pack."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import flac_workload as FW
import oracle_lib as O

pytestmark = pytest.mark.skipif(not FW.ensure_ref(), reason="oracle/_ref/libflac_ref.so not built (needs /root/reference)")

NAMES = sorted(FW.index())


def interleaved(frames):
    return np.concatenate([f[4] for f in frames], axis=1).T


@pytest.mark.parametrize("name", NAMES)
def test_reference_decoder_reproduces_the_md5_in_streaminfo(name):
    import flac_ref as F
    info, stream, frames, md5_ok = FW.load(name)
    si = F.streaminfo(stream)
    assert md5_ok                                                     # FLAC__stream_decoder_finish: MD5 of the decoded audio matches
    assert (si["bits"], si["channels"], si["rate"], si["total_samples"]) == (info["bits"], info["channels"], info["rate"], info["frames"])
    pcm = interleaved(frames)
    assert pcm.shape == (info["frames"], info["channels"])
    bps = info["bits"] // 8
    raw = (pcm.reshape(-1).astype(np.int64) & ((1 << info["bits"]) - 1)).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :bps].tobytes()
    assert hashlib.md5(raw).digest() == si["md5"] == bytes.fromhex(info["md5"])       # little-endian samples at byte depth
    assert all(f[0] <= info["blocksize"] and f[1] == info["channels"] and f[2] == info["bits"] for f in frames)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_flac_pack_of_the_decoded_frames_is_the_original_audio(name):
    info, _stream, frames, _ = FW.load(name)
    ch, bits = info["channels"], info["bits"]
    want = FW.pack_be(interleaved(frames), bits)
    got = np.zeros(want.size, dtype=np.uint8)
    at = 0
    for k, first, n in FW.callback_write_chunks(frames):
        planes = frames[k][4]
        ptrs = (C.POINTER(C.c_int32) * ch)(*[planes[c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(ch)])
        nb = C.c_uint32(0)
        assert O.lib().ohp_flac_pack(ptrs, ch, first, n, bits, got[at:].ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
        assert nb.value == n * ch * bits // 8 and nb.value <= FW.MAX_BYTES
        at += nb.value
    assert at == want.size and np.array_equal(got, want)


def test_encoder_and_decoder_round_trip_fresh_audio():
    import flac_ref as F
    rng = np.random.default_rng(9)
    for bits, ch, block in [(16, 2, 4096), (24, 2, 1152), (24, 8, 2048)]:
        lim = 1 << (bits - 1)
        pcm = rng.integers(-lim, lim, size=(5000, ch), dtype=np.int64).astype(np.int32)      # noise: verbatim / poorly predicted subframes
        frames, md5_ok = F.decode(F.encode(pcm, bits, 48000, blocksize=block, level=4))
        assert md5_ok and np.array_equal(interleaved(frames), pcm)
