#!/usr/bin/env python3
"""FlywheelRamper on one GPU: `streams` starving streams, each 1 ms of training audio -> 20 ms of
ramp audio per channel (StarvationRamper.cpp:374-375).  Prints one JSON line: ramp output samples per second, kernel time,
and the same batch on the CPU oracle (one thread) for scale.  Usage: python tools/bench_flywheel.py [--streams 2048]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
JIFFIES_PER_MS = 56448


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2048)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from ohpipeline_amd import capi
    jps = 56448000 // a.rate
    in_samples, out_frames, block = JIFFIES_PER_MS // jps, 20 * JIFFIES_PER_MS // jps, JIFFIES_PER_MS // jps
    ch = a.channels
    rng = np.random.default_rng(5)
    t = np.arange(in_samples)
    planes = np.empty((a.streams, ch, in_samples), dtype=">i4")
    for s in range(a.streams):
        for c in range(ch):
            f = 200.0 + 37.0 * ((s * ch + c) % 97)
            x = 0.6 * np.sin(2 * np.pi * f * t / a.rate + 0.1 * s) + 0.01 * rng.standard_normal(in_samples)
            planes[s, c] = np.round(x * (2 ** 31 - 1)).astype(np.int64).clip(-2 ** 31, 2 ** 31 - 1)
    src = planes.view(np.uint8).reshape(-1)
    d = np.zeros(a.streams, dtype=capi.FLYWHEEL_DESC)
    d["src_offset"] = np.arange(a.streams, dtype=np.uint64) * (ch * in_samples * 4)
    d["channel_bytes"] = in_samples * 4
    d["dst_offset"] = np.arange(a.streams, dtype=np.uint64) * (out_frames * ch * 4)
    d["in_samples"], d["out_frames"], d["block_frames"], d["sample_rate"], d["channels"] = in_samples, out_frames, block, a.rate, ch
    dst_bytes = a.streams * out_frames * ch * 4
    ctx = capi.Context(0)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dst_bytes)
    batch = ctx.flywheel_batch(d, src.size, dst_bytes)
    for _ in range(a.warmup):
        ctx.flywheel_run(batch, d_src, d_dst)
    ctx.sync()
    ev = [(ctx.event(), ctx.event()) for _ in range(a.steps)]
    for e0, e1 in ev:
        ctx.record(e0); ctx.flywheel_run(batch, d_src, d_dst); ctx.record(e1)
    ctx.sync()
    ms = sorted(ctx.elapsed_ms(e0, e1) for e0, e1 in ev)
    avg = sum(ms) / len(ms)
    lanes = a.streams * ch
    out = dict(metric="FlywheelRamper ramp samples/s (per channel)", ms_avg=round(avg, 4), ms_min=round(ms[0], 4),
               msamples_per_s=round(lanes * out_frames / avg / 1e3, 1), lanes=lanes, waves=(lanes + 63) // 64,
               config=dict(streams=a.streams, rate=a.rate, channels=ch, in_samples=in_samples, out_frames=out_frames))
    if not a.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        n = min(a.streams, 256)
        ref = np.zeros(out_frames * ch * 4, dtype=np.uint8)
        got = ctx.download(d_dst, dst_bytes)
        t0 = time.perf_counter()
        ok = True
        for s in range(n):
            blob = src[s * ch * in_samples * 4:(s + 1) * ch * in_samples * 4]
            O.lib().ohp_flywheel_ramp(blob.ctypes.data, in_samples * 4, in_samples, a.rate, ch, out_frames, block, ref.ctypes.data)
            ok = ok and np.array_equal(ref, got[s * ref.size:(s + 1) * ref.size])
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(n * ch * out_frames / dt / 1e6, 2), unit="Msamples/s", cores=1, kind="port",
                                   sample=f"{n} of the streams, one thread (includes the ctypes call per stream)")
        out["check"] = "ok" if ok else "MISMATCH"
    print(json.dumps(out))
    ctx.batch_destroy(batch)
    ctx.close()


if __name__ == "__main__":
    main()
