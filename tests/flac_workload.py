"""Helpers for BASELINE config 5 (FLAC decode -> pack -> resample -> ramp -> fmt): the committed FLAC fixtures decoded by the
reference's own libFLAC (tests/flac_ref.py) into the planar TInt32 frames CodecFlac::CallbackWrite receives, and that
callback's chunking (OpenHome/Media/Codec/Flac.cpp:355-417).  Test infrastructure only."""
import json
import os

import numpy as np

import flac_ref as F

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE_DIR = os.path.join(HERE, "golden", "flac")
MAX_BYTES = 9216                                             # sizeof(CodecFlac::iBuf) = DecodedAudio::kMaxBytes, Flac.cpp:51


def ensure_ref():
    """Builds oracle/_ref where the reference tree exists; returns False when the decoder is not available."""
    if not F.available() and os.path.isdir("/root/reference/thirdparty/flac-1.2.1"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "-s", "ref"])
    return F.available()


def index():
    return json.load(open(os.path.join(FIXTURE_DIR, "index.json")))


def load(name):
    """Returns (info, frames, md5_ok): frames as flac_ref.decode gives them."""
    info = index()[name]
    stream = open(os.path.join(FIXTURE_DIR, name + ".flac"), "rb").read()
    frames, md5_ok = F.decode(stream)
    return info, stream, frames, md5_ok


def callback_write_chunks(frames):
    """[(frame index, first sample, samples)] in the order CallbackWrite hands audio to OutputAudioPcm (Flac.cpp:379-417)."""
    chunks = []
    for k, (blocksize, ch, bits, _rate, _planes) in enumerate(frames):
        max_samples = MAX_BYTES // ((bits // 8) * ch)
        start, left = 0, blocksize
        while left > 0:
            n = min(left, max_samples)
            chunks.append((k, start, n))
            start += n
            left -= n
    return chunks


def pack_be(pcm, bits):
    """int32 [frames, channels] -> packed big-endian interleaved bytes at bits/8 bytes per subsample (numpy, independent of
    the oracle's and the device's packers)."""
    bps = bits // 8
    v = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1).astype(np.int64) & ((1 << bits) - 1)
    out = np.empty((v.size, bps), dtype=np.uint8)
    for b in range(bps):
        out[:, b] = (v >> (8 * (bps - 1 - b))) & 0xff
    return out.reshape(-1)
