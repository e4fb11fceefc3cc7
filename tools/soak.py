#!/usr/bin/env python3
"""Randomised soak of the GPU paths against the oracle with seeds the test suite does not use (run on the GPU box; not part
of the suite because of its length).  Usage: python tools/soak.py [--minutes 3] [--first-seed 1001]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=3.0)
    ap.add_argument("--first-seed", type=int, default=1001)
    a = ap.parse_args()
    import test_gpu_bench_workloads as B
    import test_gpu_flywheel as F
    import test_gpu_fmt as M
    import test_gpu_parity as P
    import test_gpu_songcast as S
    from ohpipeline_amd import capi
    ctx = capi.Context(0)
    deadline = time.time() + a.minutes * 60
    seed, counts = a.first_seed - 1, {"src_tilings": 0, "long_rows": 0, "songcast": 0, "pcm_matrix": 0, "flywheel": 0, "fmt_mixed": 0, "wg_ragged": 0}
    wg_formats = [(44100, 2, 24), (44100, 6, 24), (44100, 8, 24), (96000, 2, 24), (96000, 6, 24), (96000, 8, 24), (44100, 2, 16)]
    layouts = ["stereo_s24", "six_s24", "halfband_stereo", "halfband_eight", "mono_s16", "stereo_s32", "planar16", "five_s24"]
    real_rng = np.random.default_rng
    formats = [(48000, 24, 2), (44100, 16, 2), (96000, 32, 2), (44100, 24, 1), (48000, 24, 6), (48000, 32, 8), (44100, 16, 6), (48000, 8, 2), (192000, 24, 2)]
    last = time.time()
    while time.time() < deadline:
        seed += 1
        if time.time() - last > 30:                          # (a silent GPU job is taken to be hung)
            print("soak:", counts, flush=True)
            last = time.time()
        P.test_src_block_kernel_irregular_message_tilings(ctx, seed)
        counts["src_tilings"] += 1
        if seed % 4 == 0:                                    # round 3: rows of several blocks, forced onto small batches, by layout
            P.test_long_row_units_match_the_oracle(ctx, layouts[(seed // 4) % len(layouts)], seed)
            counts["long_rows"] += 1
        if seed % 2 == 0:                                    # round 4: ragged messages and arbitrary ramps on the workgroup matrix kernel, by format
            B.test_ragged_messages_and_arbitrary_ramps_on_the_workgroup_kernel(ctx, *wg_formats[(seed // 2) % len(wg_formats)], seed)
            counts["wg_ragged"] += 1
        rng = np.random.default_rng(seed)
        w = S.Workload()
        for k in range(int(rng.integers(1, 12))):
            rate, bits, ch = formats[int(rng.integers(0, len(formats)))]
            w.add_stream(rng, rate, bits, ch, int(rng.integers(1, 30)), codec=bytes(rng.integers(65, 91, int(rng.integers(0, 30)), dtype=np.uint8)),
                         latency_ms=int(rng.integers(0, 500)), sample_start=int(rng.integers(0, 1 << 40)), samples_total=int(rng.integers(0, 1 << 40)),
                         halt_last=bool(rng.integers(0, 2)), gap=int(rng.integers(0, 9)), attenuate=True)
        out, want = w.run(ctx)
        S.check(out, want, w.expected)
        counts["songcast"] += 1
        descs, src, dst_bytes = P.matrix_descs(rng, [8, 16, 24, 32], [1, 2, 3, 6, 8], [1, 2, 7, 43, 220, 1000], [P.LE, P.BE],
                                               [(8, P.BE), (16, P.BE), (24, P.BE), (32, P.BE), (16, P.LE), (24, P.LE), (32, P.LE)])
        got = P.run_pcm(ctx, descs, src, dst_bytes)
        wantp = P.oracle_pcm(descs, src, dst_bytes)
        assert np.array_equal(got, wantp), f"pcm matrix seed {seed}"
        counts["pcm_matrix"] += 1
        if seed % 10 == 0:                                   # the suites' own randomised tests, reseeded
            np.random.default_rng = lambda s0=None, _k=seed: real_rng(None if s0 is None else int(s0) + 7919 * _k)
            try:
                F.test_flywheel_batch_matches_oracle(ctx)
                M.test_line_kernel_equals_byte_kernel_on_mixed_batches(ctx)
                M.test_sender_pack_batches_of_wider_streams(ctx)
                pair = [(16, 16), (24, 24), (32, 24), (16, 24), (24, 32), (32, 32), (24, 16), (16, 32), (32, 16)][(seed // 10) % 9]
                P.test_pcm_ramped_groups_every_channel_count(ctx, *pair)
                P.test_pcm_uniform_batches_every_depth_pair(ctx, *pair)
            finally:
                np.random.default_rng = real_rng
            counts["flywheel"] += 1
            counts["fmt_mixed"] += 1
    print("soak ok:", counts, "seeds", a.first_seed, "..", seed)
    ctx.close()


if __name__ == "__main__":
    main()
