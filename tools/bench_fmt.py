#!/usr/bin/env python3
"""Throughput of the layout-changing processors on one GPU, HIP events around the
launch, algorithmic bytes (read once + written once) against the 8 TB/s HBM peak.  One JSON line per kind.
Usage: python tools/bench_fmt.py [--descs 65536] [--frames 240]"""
import argparse
import json
import os
import sys

import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--descs", type=int, default=262144)
    ap.add_argument("--frames", type=int, default=240, help="frames per descriptor (a 5 ms message)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of back-to-back launches before the timed ones (untimed), so that they see the clock the chip holds under this load")
    ap.add_argument("--case", type=int, default=-1, help="run only this case (profiling runs)")
    a = ap.parse_args()
    from ohpipeline_amd import capi
    ctx = capi.Context(0)
    n, f = a.descs, a.frames
    cases = [("a11 unpack to planar (S24 stereo -> 2 x S32 planes)", capi.FMT_UNPACK_PLANAR, 2, 24, 0, f * 2 * 3, f * 2 * 4),
             ("a13 Songcast sender pack (S32 stereo -> S24)", capi.FMT_SENDER_PACK, 2, 32, 0, f * 2 * 4, f * 2 * 3),
             ("a13 Songcast sender pack (S24 six channels -> two)", capi.FMT_SENDER_PACK, 6, 24, 0, f * 6 * 3, f * 2 * 3),
             ("a14 FLAC packer (2 x TInt32 planes -> S24 interleaved)", capi.FMT_FLAC_PACK, 2, 32, 24, f * 2 * 4, f * 2 * 3)]
    rng = np.random.default_rng(1)
    if a.case >= 0:
        cases = cases[a.case:a.case + 1]
    for name, kind, ch, sbits, dbits, in_b, out_b in cases:
        d = np.zeros(n, dtype=capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["n_frames"] = kind, ch, sbits, dbits, f
        d["src_offset"] = np.arange(n, dtype=np.uint64) * in_b
        d["dst_offset"] = np.arange(n, dtype=np.uint64) * out_b
        d["src_plane_stride"] = f * 4
        d["dst_plane_stride"] = f * 4
        src = rng.integers(0, 256, size=n * in_b, dtype=np.uint8)
        d_src, d_dst = ctx.upload(src), ctx.malloc(n * out_b)
        b = ctx.fmt_batch(d, src.size, n * out_b)
        ctx.fmt_run(b, d_src, d_dst)
        ctx.sync()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < a.sustain:                         # steady state first, by the clock (bench.py does the same)
            for _ in range(16):
                ctx.fmt_run(b, d_src, d_dst)
            ctx.sync()
        ev = [(ctx.event(), ctx.event()) for _ in range(a.steps)]
        for e0, e1 in ev:
            ctx.record(e0); ctx.fmt_run(b, d_src, d_dst); ctx.record(e1)
        ctx.sync()
        ms = sum(ctx.elapsed_ms(e0, e1) for e0, e1 in ev) / len(ev)
        algo = n * (in_b + out_b)
        print(json.dumps(dict(kernel=name, ms_avg=round(ms, 4), gbps=round(algo / ms / 1e6, 1),
                              frac_of_8TBps=round(algo / ms / 1e6 / 8000.0, 4), descs=n, frames=f)))
        ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.close()


if __name__ == "__main__":
    main()
