#!/usr/bin/env python3
"""Throughput of the PCM message path (unpack -> attenuate -> ramp|silence -> pack, no
resampling) on one GPU: `streams` contiguous streams cut into 5 ms messages with a ramp-up / ramp-down schedule,
inputs resident in HBM.  Prints one JSON line: GB/s over the algorithmic bytes (source read once + destination written
once) against the 8 TB/s HBM peak.  Usage: python tools/bench_pcm.py [--variant 0|1] [--src-bits 24 --dst-bits 24 ...]
With --check the output is compared with the CPU oracle on a few streams (test infrastructure, not the measured path)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--frames", type=int, default=480000, help="frames per stream")
    ap.add_argument("--msg-frames", type=int, default=240)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--src-bits", type=int, default=24)
    ap.add_argument("--dst-bits", type=int, default=24)
    ap.add_argument("--src-endian", choices=["big", "little"], default="big")
    ap.add_argument("--dst-endian", choices=["big", "little"], default="big")
    ap.add_argument("--attenuation", type=int, default=256)
    ap.add_argument("--misalign", type=int, default=0, help="byte offset added to both arenas' stream starts")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of back-to-back launches before the timed ones (untimed), so that they see the clock the chip holds under this load")
    ap.add_argument("--all-ramped", action="store_true", help="every message carries a ramp (worst case for RampApplicator, a7)")
    ap.add_argument("--mix", action="store_true", help="streams cycle through six layouts (stereo S24, S16->S24, S32->S24, six-channel S24, S16, S24->S32): "
                                                       "one batch, one launch per layout")
    ap.add_argument("--mix-8bit", action="store_true", help="with --mix: every 64th stream is 8-bit (the general staged path, in the same batch)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (one thread) on a bounded sample of the streams")
    a = ap.parse_args()

    from ohpipeline_amd import capi, hostmodel
    ch, sb, db = a.channels, a.src_bits // 8, a.dst_bits // 8
    n_msgs = (a.frames + a.msg_frames - 1) // a.msg_frames
    first = np.arange(n_msgs, dtype=np.int64) * a.msg_frames
    count = np.minimum(a.msg_frames, a.frames - first)
    jps = 56448000 // 48000
    sched = np.array(hostmodel.stream_ramp_schedule([int(c) * jps for c in count], 50 * 56448, 500 * 56448), dtype=np.int64)
    d = np.zeros(a.streams * n_msgs, dtype=capi.MSG_DESC)
    layouts = [(ch, a.src_bits, a.dst_bits)]
    if a.mix:
        layouts = [(2, 24, 24), (2, 16, 24), (2, 32, 24), (6, 24, 24), (2, 16, 16), (2, 24, 32)]
    src_pos = dst_pos = 0
    algo = 0
    stream_dst_end = []
    for s in range(a.streams):
        lch, lsbits, ldbits = layouts[s % len(layouts)]
        if a.mix and a.mix_8bit and s % 64 == 63:
            lch, lsbits, ldbits = 2, 8, 8
        lsb, ldb = lsbits // 8, ldbits // 8
        sl = slice(s * n_msgs, (s + 1) * n_msgs)
        d["src_offset"][sl] = src_pos + a.misalign + first * lch * lsb
        d["dst_offset"][sl] = dst_pos + a.misalign + first * lch * ldb
        d["n_frames"][sl] = count
        d["flags"][sl] = sched[:, 0]
        d["ramp_start"][sl] = sched[:, 1]
        d["ramp_end"][sl] = sched[:, 2]
        d["channels"][sl], d["src_bits"][sl], d["dst_bits"][sl] = lch, lsbits, ldbits
        src_pos += a.frames * lch * lsb + a.misalign
        dst_pos += a.frames * lch * ldb + a.misalign
        algo += a.frames * lch * (lsb + ldb)
        stream_dst_end.append(dst_pos)
    if a.all_ramped:
        d["flags"] = capi.FLAG_RAMP
        d["ramp_start"], d["ramp_end"] = 16384, 3000
    d["attenuation"] = a.attenuation
    d["src_endian"] = capi.ENDIAN_BIG if a.src_endian == "big" else capi.ENDIAN_LITTLE
    d["dst_endian"] = capi.ENDIAN_BIG if a.dst_endian == "big" else capi.ENDIAN_LITTLE
    src_bytes, dst_bytes = src_pos, dst_pos
    rng = np.random.default_rng(1234)
    src = rng.integers(0, 256, size=src_bytes, dtype=np.uint8)

    ctx = capi.Context(0)
    ctx.set_kernel_variant(a.variant)
    d_src = ctx.upload(src)
    d_dst = ctx.malloc(dst_bytes)
    ctx.memset(d_dst, 0xEE, dst_bytes)
    batch = ctx.pcm_batch(d, src_bytes, dst_bytes)
    import time
    ctx.pcm_run(batch, d_src, d_dst)
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.sustain:                            # steady state first, by the clock (bench.py does the same)
        for _ in range(16):
            ctx.pcm_run(batch, d_src, d_dst)
        ctx.sync()
    for _ in range(a.warmup):
        ctx.pcm_run(batch, d_src, d_dst)
    ctx.sync()
    ev = [(ctx.event(), ctx.event()) for _ in range(a.steps)]
    for k in range(a.steps):
        ctx.record(ev[k][0])
        ctx.pcm_run(batch, d_src, d_dst)
        ctx.record(ev[k][1])
    ctx.sync()
    ms = sorted(ctx.elapsed_ms(e0, e1) for e0, e1 in ev)
    avg = sum(ms) / len(ms)
    out = dict(metric="PCM message path GB/s", ms_avg=round(avg, 4), ms_min=round(ms[0], 4),
               gbps=round(algo / avg / 1e6, 1), frac_of_8TBps=round(algo / avg / 1e6 / 8000.0, 4),
               msamples_per_s=round(a.streams * a.frames / avg / 1e3, 1), algorithmic_bytes=algo,
               config=dict(streams=a.streams, frames=a.frames, msg_frames=a.msg_frames, channels=ch, src_bits=a.src_bits,
                           dst_bits=a.dst_bits, src_endian=a.src_endian, dst_endian=a.dst_endian,
                           attenuation=a.attenuation, all_ramped=a.all_ramped, misalign=a.misalign, variant=a.variant, msgs=int(d.size), sustain_s=a.sustain,
                           mix=bool(a.mix), mix_8bit=bool(a.mix and a.mix_8bit)))
    if a.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import oracle_lib as O
        n_chk = min(a.streams, 64 if a.mix else 3)
        k = n_chk * n_msgs
        part = np.ascontiguousarray(d[:k]).view(O.MSG_DESC) if d.dtype != O.MSG_DESC else np.ascontiguousarray(d[:k])
        ref = np.full(dst_bytes, 0xEE, dtype=np.uint8)
        rc = O.lib().ohp_msg_process_batch(part.ctypes.data_as(C.c_void_p), part.size, src.ctypes.data_as(C.c_void_p),
                                            ref.ctypes.data_as(C.c_void_p))
        assert rc == 0
        got = ctx.download(d_dst, dst_bytes)
        n = stream_dst_end[n_chk - 1]
        out["check"] = "ok" if np.array_equal(got[:n], ref[:n]) else "MISMATCH"
    if a.cpu:
        import time
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes as C
        import oracle_lib as O
        k = min(a.streams, 16) * n_msgs
        part = np.ascontiguousarray(d[:k])
        ref = np.zeros(dst_bytes, dtype=np.uint8)
        t0 = time.perf_counter()
        rc = O.lib().ohp_msg_process_batch(part.ctypes.data_as(C.c_void_p), part.size, src.ctypes.data_as(C.c_void_p),
                                            ref.ctypes.data_as(C.c_void_p))
        dt = time.perf_counter() - t0
        assert rc == 0
        out["cpu_baseline"] = dict(value=round(min(a.streams, 16) * a.frames / dt / 1e6, 2), unit="Msamples/s", cores=1, kind="port",
                                   sample=f"{min(a.streams, 16)} of the streams, one thread (gcc -O2 oracle, {dt:.2f} s)")
    print(json.dumps(out))
    ctx.batch_destroy(batch)
    ctx.close()


if __name__ == "__main__":
    main()
