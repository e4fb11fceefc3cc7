"""bench_flac.py -- BASELINE configs[4] for bench.py: FLAC frames -> pack (CodecFlac::CallbackWrite, Flac.cpp:379-417) ->
44.1->48 kHz resample -> ramp -> S24 on the device, 256 streams per GPU.

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
    def __init__(self):
        self.parts = []           # per bit depth: dict(batch, d_planes, d_packed, planes_bytes, packed_bytes, expect)
        self.decode_s = 0.0
        self.decode_threads = 1
        self.frames_decoded = 0
        self.stream_bytes = 0

    def run(self, ctx, ev):
        if ev is not None:
            ctx.record(ev[0])
        for p in self.parts:
            ctx.fmt_run(p["batch"], p["d_planes"], p["d_packed"])
        if ev is not None:
            ctx.record(ev[1])

    def report(self, pack_ms):
        pb = sum(p["planes_bytes"] + p["packed_bytes"] for p in self.parts)
        return {"programmes": PROGRAMMES, "encoded_bytes": self.stream_bytes, "decoder": "libFLAC 1.2.1 of the reference tree (oracle/_ref), host",
                "host_decode": {"frames_per_s_M": round(self.frames_decoded / self.decode_s / 1e6, 2), "threads": self.decode_threads,
                                "seconds": round(self.decode_s, 3), "what": "every stream decoded once, wall clock, not part of value"},
                "pack_kernel_ms": round(pack_ms, 4), "pack_gbps": round(pb / (pack_ms * 1e-3) / 1e9, 1)}

    def check(self, ctx):
        ok = all(np.array_equal(ctx.download(p["d_packed"], p["packed_bytes"]), p["expect"]) for p in self.parts)
        return "lossless: the device's packed PCM is the PCM that was encoded" if ok else "MISMATCH"

    def close(self, ctx):
        for p in self.parts:
            ctx.batch_destroy(p["batch"])
            ctx.free(p["d_planes"])
            ctx.free(p["d_packed"])


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
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        decoded = list(ex.map(lambda s: F.decode(encoded[s % PROGRAMMES][2]), ids))
    front.decode_s, front.decode_threads = time.perf_counter() - t0, threads
    front.frames_decoded = len(ids) * frames
    front.stream_bytes = sum(len(encoded[s % PROGRAMMES][2]) for s in ids)
    groups = []
    for bits in (16, 24):
        mine = [i for i, s in enumerate(ids) if encoded[s % PROGRAMMES][0] == bits]
        if not mine:
            continue
        bps = bits // 8
        planes, descs = [], []
        plane_off = 0
        for n_local, i in enumerate(mine):
            frs, md5_ok = decoded[i]
            assert md5_ok
            done = 0
            for (blocksize, ch, fbits, _rate, pl) in frs:
                assert ch == 2 and fbits == bits
                planes.append(np.ascontiguousarray(pl, dtype=np.int32).reshape(-1))        # [2][blocksize]
                max_samples = MAX_BYTES // (bps * ch)                                       # CallbackWrite's chunking (Flac.cpp:379-417)
                start = 0
                while start < blocksize:
                    n = min(blocksize - start, max_samples)
                    descs.append((plane_off + start * 4, blocksize * 4, (n_local * frames + done + start) * ch * bps, n))
                    start += n
                plane_off += 2 * blocksize * 4
                done += blocksize
            assert done == frames
        planes = np.concatenate(planes)
        d = np.zeros(len(descs), dtype=capi.FMT_DESC)
        arr = np.array(descs, dtype=np.int64)
        d["src_offset"], d["src_plane_stride"], d["dst_offset"], d["n_frames"] = arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]
        d["kind"], d["channels"], d["src_bits"], d["dst_bits"] = capi.FMT_FLAC_PACK, 2, 32, bits
        packed_bytes = len(mine) * frames * 2 * bps
        part = {"planes_bytes": planes.nbytes, "packed_bytes": packed_bytes}
        part["d_planes"] = ctx.upload(planes.view(np.uint8))
        part["d_packed"] = ctx.malloc(packed_bytes)
        ctx.memset(part["d_packed"], 0, packed_bytes)
        part["batch"] = ctx.fmt_batch(d, planes.nbytes, packed_bytes)
        part["expect"] = np.concatenate([pack_be(encoded[ids[i] % PROGRAMMES][1], bits) for i in mine])
        front.parts.append(part)
        g = Group(capi, RATE, 2, [ids[i] for i in mine], frames, src_bits=bits, src_endian=capi.ENDIAN_BIG)
        g.src = part["expect"]                                   # (what the pack must produce: the CPU oracle's input)
        g.d_src_external = part["d_packed"]
        groups.append(g)
    return groups, "weak", front
