#!/usr/bin/env python3
"""Makes tests/golden/flac/*.flac: short synthetic PCM encoded by the REFERENCE's own vendored libFLAC 1.2.1 encoder
(oracle/_ref/libflac_ref.so, built from /root/reference/thirdparty/flac-1.2.1 by `make -C oracle ref`).  The fixtures are
data: FLAC streams whose STREAMINFO carries the MD5 of the audio they hold, which is what pins the decoder's output.
Run from the repo root where /root/reference exists:  python tests/golden/make_flac_fixtures.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import flac_ref as F  # noqa: E402

# name, bits, channels, rate, frames, blocksize, compression level
CASES = [("s16_stereo_44k1_b1152_l5", 16, 2, 44100, 11025, 1152, 5),
         ("s24_stereo_44k1_b4096_l8", 24, 2, 44100, 11025, 4096, 8),
         ("s24_stereo_44k1_b576_l0", 24, 2, 44100, 6000, 576, 0),
         ("s24_6ch_48k_b4608_l3", 24, 6, 48000, 9000, 4608, 3),
         ("s8_mono_8k_b256_l2", 8, 1, 8000, 3000, 256, 2)]


def synth(name, bits, ch, rate, frames):
    """Deterministic: a few partials per channel plus noise, right-justified at the given depth."""
    rng = np.random.default_rng(sum(name.encode()))
    t = np.arange(frames)
    amp = (1 << (bits - 1)) * 0.55
    cols = []
    for c in range(ch):
        x = amp * (0.7 * np.sin(2 * np.pi * (220.0 * (c + 1)) * t / rate) + 0.3 * np.sin(2 * np.pi * (3100.0 + 17 * c) * t / rate))
        x += rng.normal(0, amp * 0.02, frames)
        cols.append(x)
    lim = (1 << (bits - 1)) - 1
    return np.clip(np.round(np.stack(cols, axis=1)), -lim - 1, lim).astype(np.int32)


def main():
    out_dir = os.path.join(HERE, "flac")
    os.makedirs(out_dir, exist_ok=True)
    index = {}
    for name, bits, ch, rate, frames, block, level in CASES:
        pcm = synth(name, bits, ch, rate, frames)
        stream = F.encode(pcm, bits, rate, blocksize=block, level=level)
        decoded, md5_ok = F.decode(stream)
        assert md5_ok and np.array_equal(np.concatenate([f[4] for f in decoded], axis=1).T, pcm)
        open(os.path.join(out_dir, name + ".flac"), "wb").write(stream)
        info = F.streaminfo(stream)
        index[name] = dict(bits=bits, channels=ch, rate=rate, frames=frames, blocksize=block, level=level, bytes=len(stream),
                           md5=info["md5"].hex())
        print(name, len(stream), "bytes")
    json.dump(index, open(os.path.join(out_dir, "index.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
