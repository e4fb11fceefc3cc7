"""bench.py's own workloads, small, through the same Group objects and against the oracle: every (filter, layout) group of
config 4 (44.1 / 96 kHz x 2 / 6 / 8 channels, the host model's ramp schedule, full-scale seeded noise) and config 3.
Round 2 shipped a block kernel for a few hours whose 6- and 8-channel outputs were one LSB off in four of ten -- every
kernel-level parity test was green (their random data happened not to provoke it), the bench's own check was not: these
are that check, as a test."""
import argparse
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
from ohpipeline_amd import capi  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def run_groups(ctx, groups):
    for g in groups:
        g.attach(ctx)
        try:
            ctx.src_run(g.batch, g.d_src, g.d_dst)
            ctx.sync()
            got = ctx.download(g.d_dst, g.dst_bytes)
            plan = g.plan
        finally:
            g.detach(ctx)
        ref = O.Src(g.rate_in, bench.RATE_OUT, g.taps, bench.BETA, bench.F_PASS)
        want = np.zeros(g.dst_bytes, dtype=np.uint8)
        assert ref.process_batch(g.oracle_descs.view(O.SRC_MSG_DESC), g.src if g.oracle_src is None else g.oracle_src, want) == 0
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (g.rate_in, g.channels, int(bad.size), int(bad[0]))
        assert plan["block_kernel_out_frames"] > 0.9 * len(g.stream_ids) * g.out_total       # (and it ran on the block kernel)


@pytest.mark.parametrize("variant", [0, 2], ids=["lean", "round1"])
def test_config4_groups_are_bit_exact(ctx, variant):
    args = argparse.Namespace(config=4, streams=96, seconds=0.6, rate_in=44100, channels=2)
    groups, scaling = bench.build_groups(capi, args, 0, 1)
    assert scaling == "strong" and sorted((g.rate_in, g.channels) for g in groups) == \
        [(44100, 2), (44100, 6), (44100, 8), (96000, 2), (96000, 6), (96000, 8)]
    ctx.set_kernel_variant(variant)
    try:
        run_groups(ctx, groups)
    finally:
        ctx.set_kernel_variant(0)


def test_config3_group_is_bit_exact(ctx):
    args = argparse.Namespace(config=3, streams=24, seconds=1.0, rate_in=44100, channels=2)
    run_groups(ctx, bench.build_groups(capi, args, 0, 1)[0])
