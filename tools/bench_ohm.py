#!/usr/bin/env python3
"""Throughput of the Songcast sender frames on one GPU: HIP events around ohgpu_ohm_batch_run,
algorithmic bytes (source audio read once + datagrams written once) against the 8 TB/s HBM peak.  One JSON line per case.
A frame is 5 ms of one stream (240 sample instants at 48 kHz); --streams x --packets frames per launch.
Usage: python tools/bench_ohm.py [--streams 1024] [--packets 256]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(capi, n_streams, packets, samples, ch, bits, ramped, codec=b"FLAC"):
    wire_b = min(bits // 8, 3)
    wire_ch = min(ch, 2)
    header = 58 + len(codec)
    frame_bytes = header + samples * wire_ch * wire_b
    in_b = samples * ch * bits // 8
    n = n_streams * packets
    streams = np.zeros(n_streams, dtype=capi.OHM_STREAM)
    streams["sample_rate"], streams["bit_rate"], streams["src_channels"], streams["src_bits"] = 48000, 48000 * bits * ch, ch, bits
    streams["codec_bytes"] = len(codec)
    streams["codec"][:, :len(codec)] = np.frombuffer(codec, dtype=np.uint8)
    frames = np.zeros(n, dtype=capi.OHM_FRAME_DESC)
    k = np.arange(n, dtype=np.uint64)
    frames["dst_offset"] = k * frame_bytes                      # datagrams back to back, no alignment
    frames["stream"] = k // packets
    frames["frame"] = k % packets
    frames["sample_start"] = (k % packets) * samples
    frames["first_fragment"], frames["n_fragments"], frames["flags"] = k, 1, capi.OHM_FLAG_LOSSLESS
    frags = np.zeros(n, dtype=capi.OHM_FRAGMENT)
    frags["src_offset"], frags["n_frames"], frags["attenuation"] = k * in_b, samples, 256
    if ramped:
        frags["flags"], frags["ramp_start"], frags["ramp_end"] = 1, 16384, 8192
    return streams, frames, frags, n * in_b, n * frame_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--packets", type=int, default=256)
    ap.add_argument("--samples", type=int, default=240)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of back-to-back launches before the timed ones (untimed), so that they see the clock the chip holds under this load")
    ap.add_argument("--case", type=int, default=-1, help="run only this case (profiling runs)")
    a = ap.parse_args()
    from ohpipeline_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(1)
    cases = [("stereo S24 plain", 2, 24, False), ("stereo S32 -> S24 plain", 2, 32, False), ("stereo S24 ramped", 2, 24, True),
             ("6-channel S24 plain (channel select)", 6, 24, False), ("6-channel S24 ramped (select + ramp in one pass)", 6, 24, True)]
    if a.case >= 0:
        cases = cases[a.case:a.case + 1]
    for name, ch, bits, ramped in cases:
        streams, frames, frags, src_bytes, dst_bytes = build(capi, a.streams, a.packets, a.samples, ch, bits, ramped)
        src = rng.integers(0, 256, size=src_bytes, dtype=np.uint8)
        d_src, d_dst = ctx.upload(src), ctx.malloc(dst_bytes)
        t0 = time.perf_counter()
        b = ctx.ohm_batch(streams, frames, frags, src_bytes, dst_bytes)
        plan_ms = (time.perf_counter() - t0) * 1e3
        ctx.ohm_run(b, d_src, d_dst)
        ctx.sync()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < a.sustain:                         # steady state first, by the clock (bench.py does the same)
            for _ in range(16):
                ctx.ohm_run(b, d_src, d_dst)
            ctx.sync()
        ev = [(ctx.event(), ctx.event()) for _ in range(a.steps)]
        for e0, e1 in ev:
            ctx.record(e0); ctx.ohm_run(b, d_src, d_dst); ctx.record(e1)
        ctx.sync()
        ms = sum(ctx.elapsed_ms(e0, e1) for e0, e1 in ev) / len(ev)
        algo = src_bytes + dst_bytes
        line = dict(kernel="ohm frames: " + name, ms_avg=round(ms, 4), gbps=round(algo / ms / 1e6, 1),
                    frac_of_8TBps=round(algo / ms / 1e6 / 8000.0, 4), frames=len(frames), mframes_per_s=round(len(frames) / ms / 1e3, 1),
                    plan_ms=round(plan_ms, 1))
        print(json.dumps(line))
        ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.close()


if __name__ == "__main__":
    main()
