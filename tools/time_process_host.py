#!/usr/bin/env python3
"""Latency of the host-buffer convenience calls at a live pipeline's cadence: one 5 ms stereo S24 message per call
(H2D, launch, D2H, sync).  Usage: python tools/time_process_host.py [--calls 200]"""
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
    ap.add_argument("--calls", type=int, default=200)
    a = ap.parse_args()
    from ohpipeline_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(1)
    for n_msgs in (1, 16, 256):
        frames = 240
        d = np.zeros(n_msgs, dtype=capi.MSG_DESC)
        d["src_offset"] = np.arange(n_msgs) * frames * 6
        d["dst_offset"] = np.arange(n_msgs) * frames * 6
        d["n_frames"], d["ramp_start"], d["ramp_end"], d["attenuation"] = frames, 16384, 8192, 256
        d["channels"], d["src_bits"], d["src_endian"], d["dst_bits"], d["dst_endian"], d["flags"] = 2, 24, 1, 24, 2, 1
        src = rng.integers(0, 256, n_msgs * frames * 6, dtype=np.uint8)
        dst = np.zeros_like(src)
        for _ in range(10):
            ctx.pcm_process_host(d, src, dst)
        t = []
        for _ in range(a.calls):
            t0 = time.perf_counter()
            ctx.pcm_process_host(d, src, dst)
            t.append(time.perf_counter() - t0)
        t.sort()
        print(json.dumps(dict(call="ohgpu_pcm_process_host", msgs_per_call=n_msgs, median_us=round(t[len(t) // 2] * 1e6, 1),
                              p90_us=round(t[int(len(t) * 0.9)] * 1e6, 1), min_us=round(t[0] * 1e6, 1))))
    ctx.close()


if __name__ == "__main__":
    main()
