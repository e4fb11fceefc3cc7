"""bench_flac.py -- BASELINE configs[4] for bench.py: FLAC frames -> pack (CodecFlac::CallbackWrite, Flac.cpp:379-417) ->
44.1->48 kHz resample -> ramp -> S24 on the device, 256 streams per GPU, the pack fused into the resampler's load.

The entropy decode stays on the host, as in the reference (CodecFlac drives libFLAC on the codec thread): the synthetic
streams are encoded and decoded here with the reference's own vendored libFLAC 1.2.1, built from its sources into
oracle/_ref (tests/flac_ref.py binds it).  That is input preparation -- the decoder's planar TInt32 frames are what the
device path starts from -- and its time is reported beside the device's, never inside `value`.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
RATE = 44100
MAX_BYTES = 9216                    # sizeof(CodecFlac::iBuf) = DecodedAudio::kMaxBytes (Flac.cpp:51, Msg.h:117)
PROGRAMMES = 16                     # distinct encoded programmes; stream s plays programme s % PROGRAMMES


def programme(k, frames, bits):
    """Stereo test programme k: a few sines and a little noise, so that the encoder's predictors and residual coder have work."""
    rng = np.random.default_rng(1000 + k)
    t = np.arange(frames, dtype=np.float64) / RATE
    full = float((1 << (bits - 1)) - 1)
    x = np.zeros((frames, 2))
    for ch in range(2):
        for _ in range(4):
            f, a, ph = rng.uniform(80, 9000), rng.uniform(0.03, 0.2), rng.uniform(0, 6.28)
            x[:, ch] += a * np.sin(2 * np.pi * f * t + ph)
        x[:, ch] += 0.002 * rng.standard_normal(frames)
    return np.clip(np.round(x * full), -full - 1, full).astype(np.int32)


def pack_be(pcm, bits):
    """int32 [frames, channels] -> packed big-endian interleaved bytes (numpy; independent of the device's packer)."""
    bps = bits // 8
    v = np.ascontiguousarray(pcm, dtype=np.int32).reshape(-1).astype(np.int64) & ((1 << bits) - 1)
    out = np.empty((v.size, bps), dtype=np.uint8)
    for b in range(bps):
        out[:, b] = (v >> (8 * (bps - 1 - b))) & 0xff
    return out.reshape(-1)


class FlacFront:
    """What precedes the device in config 5: the host decode (timed, reported, never part of `value`).  Round 1 packed the
    decoder's planes on the device into a packed arena the resampler then read (a14 as a kernel of its own); now the
    resampler reads the planes themselves (OHGPU_FLAG_SRC_PLANAR32: a14 -> a1 -> a-R in one pass), so the step has no
    kernel of this object's left -- run() is where it was."""

    def __init__(self):
        self.decode_s = 0.0
        self.decode_threads = 1
        self.frames_decoded = 0
        self.stream_bytes = 0
        self.lossless = False

    def run(self, ctx, ev):
        if ev is not None:                 # (bench.py times "the pack": an empty interval now)
            ctx.record(ev[0])
            ctx.record(ev[1])

    def report(self, pack_ms):
        return {"programmes": PROGRAMMES, "encoded_bytes": self.stream_bytes, "decoder": "libFLAC 1.2.1 of the reference tree (oracle/_ref), host",
                "host_decode": {"frames_per_s_M": round(self.frames_decoded / self.decode_s / 1e6, 2), "threads": self.decode_threads,
                                "seconds": round(self.decode_s, 3), "what": "every stream decoded once, wall clock, not part of value"},
                "pack": "fused: the resampler reads the decoder's planar TInt32 output (OHGPU_FLAG_SRC_PLANAR32); no pack kernel, no packed arena"}

    def check(self, ctx):
        return "lossless: the decoded planes the device reads are the PCM that was encoded" if self.lossless else "MISMATCH"

    def close(self, ctx):
        pass


def build(capi, ctx, args, rank, world, Group):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor

    import flac_ref as F
    if not F.available():
        raise RuntimeError("config 5 needs oracle/_ref/libflac_ref.so (`make -C oracle ref`, built by __graft_entry__.build() where the reference tree exists)")
    frames = int(round(args.seconds * RATE))
    ids = list(range(rank * args.streams, (rank + 1) * args.streams))
    front = FlacFront()
    # encode the programmes (16-bit for even programme numbers, 24-bit for odd ones), then decode every stream's programme once
    encoded = {}
    for k in range(PROGRAMMES):
        bits = 16 if k % 2 == 0 else 24
        pcm = programme(k, frames, bits)
        encoded[k] = (bits, pcm, F.encode(pcm, bits, RATE))
    share = getattr(args, "host_share", None)                 # (a rank's share of the host: bench.main / bench.host_share)
    threads = max(1, min(len(os.sched_getaffinity(0)), 16, share["threads"] if share and world > 1 else 16))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        decoded = list(ex.map(lambda s: F.decode(encoded[s % PROGRAMMES][2]), ids))
    front.decode_s, front.decode_threads = time.perf_counter() - t0, threads
    front.frames_decoded = len(ids) * frames
    front.stream_bytes = sum(len(encoded[s % PROGRAMMES][2]) for s in ids)
    groups = []
    lossless = True
    for bits in (16, 24):
        mine = [i for i, s in enumerate(ids) if encoded[s % PROGRAMMES][0] == bits]
        if not mine:
            continue
        # The write callback's job (Flac.cpp:379-417) is now two copies per frame: libFLAC's buffer[c][0..blocksize) goes to
        # its place in the stream's plane c -- [stream][channel][frames] TInt32, what the device reads.
        planes = np.zeros((len(mine), 2, frames), dtype=np.int32)
        for n_local, i in enumerate(mine):
            frs, md5_ok = decoded[i]
            assert md5_ok
            done = 0
            for (blocksize, ch, fbits, _rate, pl) in frs:
                assert ch == 2 and fbits == bits
                planes[n_local, :, done:done + blocksize] = np.asarray(pl, dtype=np.int32).reshape(2, blocksize)
                done += blocksize
            assert done == frames
            lossless = lossless and np.array_equal(planes[n_local].T, encoded[ids[i] % PROGRAMMES][1])
        g = Group(capi, RATE, 2, [ids[i] for i in mine], frames, src_bits=bits, src_endian=capi.ENDIAN_BIG, planar=True)
        g.src = planes.reshape(-1).view(np.uint8)
        g.oracle_src = np.concatenate([pack_be(encoded[ids[i] % PROGRAMMES][1], bits) for i in mine])   # the oracle's composition: pack, then resample
        groups.append(g)
    front.lossless = lossless
    # (per GPU the device path is weak-scaling; the wall clock of a real step is the host decode's -- 84 M frames/s on sixteen CPUs
    # against the device's 320 G: every rank decodes its own streams on its share of the same CPUs, so the curve is flat in wall terms)
    return groups, "weak; host-decode-bound (config.flac.host_decode: the decode shares the host's CPUs among the ranks, the device path scales)", front
