"""Diagnosis helper: the groups of bench.py's config 4 against the oracle, group by group (run on the GPU box)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
from ohpipeline_amd import capi  # noqa: E402

secs, streams = float(sys.argv[1]), int(sys.argv[2])
args = argparse.Namespace(config=4, streams=streams, seconds=secs, rate_in=44100, channels=2)
ctx = capi.Context(0)
groups, _ = bench.build_groups(capi, args, 0, 1)
for g in groups:
    g.attach(ctx)
    ctx.src_run(g.batch, g.d_src, g.d_dst)
    ctx.sync()
    got = ctx.download(g.d_dst, g.dst_bytes)
    ref = O.Src(g.rate_in, 48000, g.taps, 9.0, 20000.0)
    want = np.zeros(g.dst_bytes, np.uint8)
    assert ref.process_batch(g.descs.view(O.SRC_MSG_DESC), g.src, want) == 0
    bad = np.nonzero(got != want)[0]
    per = g.dst_bytes // len(g.stream_ids)
    print(g.rate_in, g.channels, "streams", len(g.stream_ids), "bad bytes", bad.size, g.plan)
    if bad.size:
        fb = g.channels * 3
        frames = np.unique((bad % per) // fb)
        L_blk = 160 if g.rate_in == 44100 else 128
        blocks = np.unique(frames // L_blk)
        print("  first bad frames", frames[:12], "bad blocks (of stream)", blocks[:40], "...", blocks.size, "of", per // fb // L_blk)
        print("  bad frame-in-block histogram (first 24 bins of 8)", np.bincount((frames % L_blk) // 8, minlength=L_blk // 8)[:24])
        print("  bad channel histogram", np.bincount(((bad % per) % fb) // 3, minlength=g.channels))
        b0 = bad[0]
        print("  at first bad byte: got", got[b0 - 2:b0 + 6], "want", want[b0 - 2:b0 + 6])
    g.detach(ctx)
