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


def run_groups(ctx, groups, kernel=None):
    for g in groups:
        g.attach(ctx)
        try:
            if kernel is not None:
                assert ctx.src_kernel_name(g.batch) == kernel, (g.rate_in, g.channels, ctx.src_kernel_name(g.batch))
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


@pytest.mark.parametrize("variant", [0, 4, 2], ids=["tuned", "lean", "round1"])
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


def lean_instantiations():
    """(T, channels, source bytes, source LE, destination bytes, destination LE, list) of every block-kernel instantiation the
    library is built with, read from the lists the kernels are instantiated from (csrc/src_block_common.h).  `list` tells what
    runs it: "block" = both block kernels, "planar" (source bytes 0: the TInt32 planes), "halfband" and "lean_only" = the lean
    kernel alone.  A half-band instantiation has a plain T = 64 twin in the block list, which the bench's own 96 -> 48 kHz
    filter never reaches any more (it IS half-band): test_plain_64_tap_kernels_still_run covers those."""
    import re
    text = open(os.path.join(ROOT, "ohpipeline_amd", "csrc", "src_block_common.h")).read()
    out = set()
    for macro, kind in (("OHGPU_BLOCK_KERNELS_1", "block"), ("OHGPU_BLOCK_KERNELS_2", "block"), ("OHGPU_BLOCK_KERNELS_3", "block"),
                        ("OHGPU_LEAN_PLANAR_KERNELS", "planar"), ("OHGPU_LEAN_HB_KERNELS", "halfband"), ("OHGPU_LEAN_ONLY_KERNELS", "lean_only"), ("OHGPU_LEAN_MORE_KERNELS", "lean_only")):
        defs = [m.end() for m in re.finditer(r"#define %s\(X\)" % macro, text)]
        body = text[defs[-1]:]                                           # (the last definition: the one behind the diagnostic #else)
        body = body[:body.index("\n#")]
        for t, c, s_, sl, d, dl in re.findall(r"X\((\d+), (\d+), (\d+), (true|false), (\d+), (true|false)\)", body):
            out.add((int(t), int(c), int(s_), sl == "true", int(d), dl == "true", kind))
    assert len(out) >= 30
    return sorted(out)


def noise_le(stream_id, n_subsamples, bits):
    """bench.py's seeded full-scale noise at another depth, little endian (the top bytes of the same LCG words)."""
    x = bench.lcg_block((0x9E3779B9 * (stream_id + 1)) & bench.MASK, n_subsamples)
    b = np.empty((x.size, bits // 8), dtype=np.uint8)
    for k in range(bits // 8):
        b[:, k] = (x >> (8 * (4 - bits // 8 + k))) & 0xFF
    return b


@pytest.mark.parametrize("inst", lean_instantiations(), ids=lambda i: "T%d_ch%d_s%d%s_d%d%s_%s" % (i[0], i[1], i[2], "le" if i[3] else "be", i[4], "le" if i[5] else "be", i[6]))
def test_every_block_kernel_instantiation_on_the_bench_workload(ctx, inst):
    """One bench-shaped group per instantiation -- the host model's 50 ms / 500 ms ramp schedule, 5 ms messages, full-scale
    noise, enough streams for workgroups of several waves -- on the lean kernel and (the layouts it has) on round 1's.  The
    whole batch must be on the block kernel: a uniform batch of any of these layouts leaves nothing but block-unaligned ends
    to the generic one."""
    T, ch, sb, src_le, db, dst_le, kind = inst
    planar = kind == "planar"
    rate = 96000 if T == 64 else 44100
    src_bits = 16 if planar else sb * 8
    n_streams = 24 if ch <= 2 else 12
    g = bench.Group(capi, rate, ch, range(900, 900 + n_streams), int(round(0.7 * rate)), src_bits=src_bits,
                    src_endian=capi.ENDIAN_LITTLE if src_le else capi.ENDIAN_BIG, planar=planar,
                    dst_bits=db * 8, dst_endian=capi.ENDIAN_LITTLE if dst_le else capi.ENDIAN_BIG)
    assert g.taps == T
    per = g.in_frames * ch
    if planar:                                                           # the decoder's planes, and the oracle's packed composition
        v = [(noise_le(sid, per, 16).view("<i2").astype(np.int32).reshape(g.in_frames, ch)) for sid in g.stream_ids]
        g.src = np.ascontiguousarray(np.stack([x.T for x in v])).reshape(-1).view(np.uint8)
        g.oracle_src = np.concatenate([x.reshape(-1).astype(">i2").view(np.uint8) for x in v])
        g.oracle_descs["src_endian"] = capi.ENDIAN_BIG
    else:
        le = [noise_le(sid, per, src_bits) for sid in g.stream_ids]
        g.src = np.concatenate([(x if src_le else x[:, ::-1]).reshape(-1) for x in le])
    # (variant 0 is the matrix-pipe kernel for the layouts it serves: variant 4 keeps the lean kernel covered there)
    for variant in ((0, 2, 4) if kind == "block" else (0,)):
        ctx.set_kernel_variant(variant)
        try:
            run_groups(ctx, [g])
        finally:
            ctx.set_kernel_variant(0)
    # nothing but the streams' block-unaligned ends went to the generic kernel (at most a head and a tail message piece per stream
    # and message it cuts): the layout is on the fast path
    assert g.plan["block_kernel_out_frames"] >= 0.9 * len(g.stream_ids) * g.out_total, g.plan


def test_plain_64_tap_kernels_still_run(ctx):
    """The T = 64 instantiations serve every 64-tap filter that is NOT half-band.  The bench's own 96 -> 48 kHz design is
    (host_design.cpp), so take it and break the property -- one odd tap set to a non-zero value, in the oracle's table and in
    the library's alike: ohgpu_src_create must then choose the plain kernel, and the audio must still be the integer model's."""
    g = bench.Group(capi, 96000, 2, range(40, 52), 67200)
    ref = O.Src(g.rate_in, bench.RATE_OUT, g.taps, bench.BETA, bench.F_PASS)
    assert g.taps == 64 and all(ref.coef_q28[k] == 0 for k in range(1, 64, 2) if k != 31)       # half-band as designed
    table = np.ctypeslib.as_array(O.lib().ohp_src_coef_q28(ref.h), shape=(64,))                   # (the oracle's own table, in place)
    table[5] = 12345
    coef = np.array(table, dtype=np.int32)
    per = g.in_frames * g.channels
    le = [noise_le(sid, per, 24) for sid in g.stream_ids]
    g.src = np.concatenate([x.reshape(-1) for x in le])
    h = ctx.src_create(g.L, g.M, g.taps, coef)
    d_src, d_dst = ctx.upload(g.src), ctx.malloc(g.dst_bytes)
    ctx.memset(d_dst, 0, g.dst_bytes)
    b = ctx.src_batch(h, g.descs, g.src_bytes, g.dst_bytes)
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, g.dst_bytes)
    want = np.zeros(g.dst_bytes, dtype=np.uint8)
    assert ref.process_batch(g.descs, g.src, want) == 0
    assert ctx.src_plan(b)["block_kernel_out_frames"] > 0
    assert np.array_equal(got, want)
    ctx.batch_destroy(b)
    ctx.src_destroy(h)
    ctx.free(d_src)
    ctx.free(d_dst)


@pytest.mark.parametrize("channels", [2, 6, 8])
@pytest.mark.parametrize("src_le", [True, False], ids=["sle", "sbe"])
@pytest.mark.parametrize("dst_le", [True, False], ids=["dle", "dbe"])
@pytest.mark.parametrize("n_streams, seconds", [(3, 0.31), (14, 0.83)])
@pytest.mark.parametrize("rate", [44100, 96000])
def test_workgroup_matrix_kernel_takes_packed_s24_of_two_six_and_eight_channels(ctx, channels, src_le, dst_le, n_streams, seconds, rate):
    """44.1 -> 48 kHz (32 taps per phase) and 96 -> 48 kHz (the half-band decimator) S24 in either byte order, ramped heads and
    tails, stream lengths that leave the last unit of a stream partly filled (a pass of 16 / 5 / 4 rows with fewer blocks than
    rows, and one with none) and put units at both ends of the arena (the checked loads): src_mfma_wg_kernel runs them all and the
    audio is the integer model's, byte for byte."""
    g = bench.Group(capi, rate, channels, range(300, 300 + n_streams), int(round(seconds * rate)), src_bits=24,
                    src_endian=capi.ENDIAN_LITTLE if src_le else capi.ENDIAN_BIG, dst_bits=24,
                    dst_endian=capi.ENDIAN_LITTLE if dst_le else capi.ENDIAN_BIG)
    per = g.in_frames * channels
    le = [noise_le(sid, per, 24) for sid in g.stream_ids]
    g.src = np.concatenate([(x if src_le else x[:, ::-1]).reshape(-1) for x in le])
    run_groups(ctx, [g], kernel="src_mfma_wg_kernel")


@pytest.mark.parametrize("channels", [2, 6, 8])
@pytest.mark.parametrize("rate", [47700, 43800])
def test_the_shortest_and_the_longest_input_run_the_workgroup_kernel_admits(ctx, rate, channels):
    """A pass's input is ONE run -- the union of its rows -- fetched into an LDS buffer sized for the longest the geometry admits
    (csrc/src_mfma_wg_kernel.hip WgGeom::kDmaBytes, kSpanRounds).  The headline's 147 input frames per block of 160 outputs sit in the
    middle; 47.7 kHz -> 48 kHz is 159 per block (the longest run: 967 of stereo's 972 pieces, 1004 of eight channels' 1008), 43.8 kHz
    (73 : 80) is 146 (the shortest a ratio can make of a block of 160: the last round of loads has the fewest lanes).  Odd rates,
    ordinary filters -- but ratios this
    close to one make phases whose sum |c| passes 2^29, which keeps a batch off the matrix kernel (the planner ties it to the lean
    kernel's bound), so the designed table is taken at three quarters, in the oracle's table and in the library's alike.  Streams
    short enough that units sit at both ends of the arena."""
    g = bench.Group(capi, rate, channels, range(70, 73), int(round(0.37 * rate)), src_bits=24, src_endian=capi.ENDIAN_LITTLE,
                    dst_bits=24, dst_endian=capi.ENDIAN_BIG)
    assert (g.L, g.M) in ((160, 159), (80, 73))
    ref = O.Src(g.rate_in, bench.RATE_OUT, g.taps, bench.BETA, bench.F_PASS)
    table = np.ctypeslib.as_array(O.lib().ohp_src_coef_q28(ref.h), shape=(g.L * g.taps,))       # (the oracle's own table, in place)
    table[:] = (table.astype(np.int64) * 3) // 4
    coef = np.array(table, dtype=np.int32)
    assert max(int(np.abs(coef[p * g.taps:(p + 1) * g.taps].astype(np.int64)).sum()) for p in range(g.L)) < 2 ** 29
    per = g.in_frames * channels
    g.src = np.concatenate([noise_le(sid, per, 24).reshape(-1) for sid in g.stream_ids])
    h = ctx.src_create(g.L, g.M, g.taps, coef)
    d_src, d_dst = ctx.upload(g.src), ctx.malloc(g.dst_bytes)
    ctx.memset(d_dst, 0, g.dst_bytes)
    b = ctx.src_batch(h, g.descs, g.src_bytes, g.dst_bytes)
    try:
        assert ctx.src_kernel_name(b) == "src_mfma_wg_kernel"
        ctx.src_run(b, d_src, d_dst)
        got = ctx.download(d_dst, g.dst_bytes)
    finally:
        ctx.batch_destroy(b)
        ctx.src_destroy(h)
        ctx.free(d_src)
        ctx.free(d_dst)
    want = np.zeros(g.dst_bytes, dtype=np.uint8)
    assert ref.process_batch(g.descs.view(O.SRC_MSG_DESC), g.src, want) == 0
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (rate, channels, int(bad.size), int(bad[0]))


@pytest.mark.parametrize("rate, channels, src_bits", [(44100, 2, 24), (44100, 6, 24), (44100, 8, 24), (96000, 2, 24), (96000, 6, 24),
                                                      (96000, 8, 24), (44100, 2, 16)])
def test_the_device_keeps_as_many_workgroups_per_cu_as_the_kernel_is_laid_out_for(ctx, rate, channels, src_bits):
    """The workgroup matrix kernel is sized to the LDS -- three workgroups per CU at 50.7-52.0 KB each since its input run has a
    buffer of its own (csrc/src_mfma_wg_kernel.hip WgGeom::kDma) -- and with a granule too many the device would keep two: a third of the
    throughput gone, no error anywhere.  ohgpu_src_batch_occupancy asks the device what it grants the instantiation the batch runs."""
    g = bench.Group(capi, rate, channels, range(40, 44), int(round(0.4 * rate)), src_bits=src_bits, src_endian=capi.ENDIAN_LITTLE,
                    dst_bits=24, dst_endian=capi.ENDIAN_BIG)
    g.src = np.zeros(g.src_bytes, dtype=np.uint8)
    g.attach(ctx)
    try:
        assert ctx.src_kernel_name(g.batch) == "src_mfma_wg_kernel"
        occ = ctx.src_occupancy(g.batch)
    finally:
        g.detach(ctx)
    assert occ["designed_for"] == 3 and occ["workgroups_per_cu"] >= occ["designed_for"], occ
    assert occ["lds_bytes"] <= 52 * 1024, occ


@pytest.mark.parametrize("rate, n_streams", [(44100, 6), (44100, 1), (88200, 3)])
def test_a_timed_run_is_the_same_run_with_its_time_on_the_events(ctx, rate, n_streams):
    """ohgpu_src_batch_run_timed: the caller's two events bracket the batch's device work -- on the dispatch itself for a batch that is
    one launch of the workgroup matrix kernel (44.1 kHz stereo in whole blocks), recorded around the launches otherwise (a stream
    whose length leaves block-unaligned pieces for the generic kernel; 88.2 kHz: the lean kernel).  The audio is the plain run's."""
    g = bench.Group(capi, rate, 2, range(90, 90 + n_streams), int(round(0.53 * rate)) + (0 if n_streams > 1 else 77))
    per = g.in_frames * 2
    g.src = np.concatenate([noise_le(sid, per, 24).reshape(-1) for sid in g.stream_ids])
    g.attach(ctx)
    try:
        ctx.src_run(g.batch, g.d_src, g.d_dst)
        want = ctx.download(g.d_dst, g.dst_bytes)
        ctx.memset(g.d_dst, 0, g.dst_bytes)
        e0, e1 = ctx.event(), ctx.event()
        ctx.src_run(g.batch, g.d_src, g.d_dst, events=(e0, e1))
        ms = ctx.elapsed_ms(e0, e1)
        got = ctx.download(g.d_dst, g.dst_bytes)
        ctx.event_destroy(e0)
        ctx.event_destroy(e1)
    finally:
        g.detach(ctx)
    assert np.array_equal(got, want)
    assert 0.0 < ms < 50.0, ms


def test_a_timed_run_leaves_nothing_behind_that_needs_its_stream(ctx):
    """After ohgpu_src_batch_run_timed the batch's "last launch done" is not an event of the library's (the caller's two rode on the
    dispatch): a run on ANOTHER stream, a rewrite of the ramps and the destroy must still wait for that launch -- and must not ask
    the first stream, which the caller may have destroyed by then."""
    g = bench.Group(capi, 44100, 2, range(120, 126), int(round(0.53 * 44100)))
    per = g.in_frames * 2
    g.src = np.concatenate([noise_le(sid, per, 24).reshape(-1) for sid in g.stream_ids])
    g.attach(ctx)
    try:
        ctx.src_run(g.batch, g.d_src, g.d_dst)
        want = ctx.download(g.d_dst, g.dst_bytes)
        ctx.memset(g.d_dst, 0, g.dst_bytes)
        ctx.sync()
        s1, s2 = ctx.stream_create(), ctx.stream_create()
        e0, e1 = ctx.event(), ctx.event()
        ctx.src_run(g.batch, g.d_src, g.d_dst, stream=s1, events=(e0, e1))
        assert ctx.elapsed_ms(e0, e1) > 0.0                  # (synchronises on the stop event)
        ctx.stream_destroy(s1)
        ctx.src_run(g.batch, g.d_src, g.d_dst, stream=s2)    # another stream: the library waits for the device, not for s1
        ctx.sync()
        got = ctx.download(g.d_dst, g.dst_bytes)
        e2, e3 = ctx.event(), ctx.event()
        ctx.src_run(g.batch, g.d_src, g.d_dst, stream=s2, events=(e2, e3))
        ctx.stream_destroy(s2)                               # (destroying a stream waits for its work)
        for e in (e0, e1, e2, e3):
            ctx.event_destroy(e)
    finally:
        g.detach(ctx)                                        # (the batch's destroy behind a timed run whose stream is gone)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("channels, src_le, dst_le", [(6, True, False), (2, False, True)])
def test_a_plan_only_the_workgroup_kernel_reads_never_reaches_another_kernel(ctx, channels, src_le, dst_le):
    """Six-channel units are cut 30 rows long for the workgroup kernel, and big-endian-in / little-endian-out has no other block
    kernel at all: such a batch, planned under variant 0 and RUN under variant 4, must not reach the lean kernel.  A batch planned
    for the block kernels keeps no per-message descriptors (round 5: their conversion was half the validation pass), so the generic
    kernel cannot take it either: the run is REFUSED, loudly, nothing is launched -- and the same batch created while variant 4 (or
    1) is in force runs, bit-exact."""
    g = bench.Group(capi, 44100, channels, range(700, 705), int(round(0.4 * 44100)), src_bits=24,
                    src_endian=capi.ENDIAN_LITTLE if src_le else capi.ENDIAN_BIG, dst_bits=24,
                    dst_endian=capi.ENDIAN_LITTLE if dst_le else capi.ENDIAN_BIG)
    per = g.in_frames * channels
    le = [noise_le(sid, per, 24) for sid in g.stream_ids]
    g.src = np.concatenate([(x if src_le else x[:, ::-1]).reshape(-1) for x in le])
    g.attach(ctx)
    try:
        assert ctx.src_kernel_name(g.batch) == "src_mfma_wg_kernel"
        ref = O.Src(g.rate_in, bench.RATE_OUT, g.taps, bench.BETA, bench.F_PASS)
        want = np.zeros(g.dst_bytes, dtype=np.uint8)
        assert ref.process_batch(g.oracle_descs.view(O.SRC_MSG_DESC), g.src, want) == 0
        for other in (4, 1):
            ctx.set_kernel_variant(other)
            try:
                assert ctx.src_kernel_name(g.batch) == "src_kernel_v1"
                ctx.memset(g.d_dst, 0x5A, g.dst_bytes)
                with pytest.raises(capi.OhGpuError) as e:
                    ctx.src_run(g.batch, g.d_src, g.d_dst)
                assert e.value.code == capi.ERR_UNSUPPORTED and "ohgpu_set_kernel_variant(1)" in str(e.value)
                ctx.sync()
                assert np.all(ctx.download(g.d_dst, g.dst_bytes) == 0x5A)         # nothing was launched
                # ... created under that variant, the same messages run (the lean kernel where it has the layout, else the generic one)
                b2 = ctx.src_batch(g.h, g.descs, g.src.size, g.dst_bytes)
                ctx.memset(g.d_dst, 0, g.dst_bytes)
                ctx.src_run(b2, g.d_src, g.d_dst)
                ctx.sync()
                assert np.array_equal(ctx.download(g.d_dst, g.dst_bytes), want), other
                ctx.batch_destroy(b2)
            finally:
                ctx.set_kernel_variant(0)
        ctx.memset(g.d_dst, 0, g.dst_bytes)
        ctx.src_run(g.batch, g.d_src, g.d_dst)                  # ... and back on its own kernel
        ctx.sync()
        assert np.array_equal(ctx.download(g.d_dst, g.dst_bytes), want)
    finally:
        g.detach(ctx)


@pytest.mark.parametrize("src_le", [True, False], ids=["sle", "sbe"])
@pytest.mark.parametrize("dst_le", [True, False], ids=["dle", "dbe"])
@pytest.mark.parametrize("n_streams, seconds", [(3, 0.31), (14, 0.83)])
def test_workgroup_matrix_kernel_takes_packed_16_bit_stereo(ctx, src_le, dst_le, n_streams, seconds):
    """CD audio -- 44.1 kHz 16-bit stereo, either byte order -- to 48 kHz S24: the sample is the 16 bits over a zero byte, the same
    tiles; src_mfma_wg_kernel runs it (the lean kernel under variant 4) and both are the integer model's bytes."""
    g = bench.Group(capi, 44100, 2, range(500, 500 + n_streams), int(round(seconds * 44100)), src_bits=16,
                    src_endian=capi.ENDIAN_LITTLE if src_le else capi.ENDIAN_BIG, dst_bits=24,
                    dst_endian=capi.ENDIAN_LITTLE if dst_le else capi.ENDIAN_BIG)
    per = g.in_frames * 2
    le = [noise_le(sid, per, 16) for sid in g.stream_ids]
    g.src = np.concatenate([(x if src_le else x[:, ::-1]).reshape(-1) for x in le])
    run_groups(ctx, [g], kernel="src_mfma_wg_kernel")
    if not dst_le:                                              # (S16 -> S24 LE has no lean instantiation: that batch is the workgroup kernel's alone)
        ctx.set_kernel_variant(4)
        try:
            run_groups(ctx, [g], kernel="src_lean_kernel")
        finally:
            ctx.set_kernel_variant(0)


@pytest.mark.parametrize("rate, channels, src_bits", [(44100, 2, 24), (44100, 6, 24), (44100, 8, 24), (96000, 2, 24), (96000, 6, 24),
                                                      (96000, 8, 24), (44100, 2, 16)])
@pytest.mark.parametrize("seed", [1, 2])
def test_ragged_messages_and_arbitrary_ramps_on_the_workgroup_kernel(ctx, rate, channels, src_bits, seed):
    """Not the host model's schedule: every stream cut into messages of 1 .. 700 output frames at random, any message ramped with
    any endpoints (up, down, flat, one frame long), a few of them empty, streams of different lengths in one arena -- so that units
    meet ramps anywhere, block-unaligned heads and tails go to the generic kernel mid-stream, and the last unit of a stream holds
    any number of blocks.  The workgroup kernel runs the whole blocks; every byte is the integer model's."""
    rng = np.random.default_rng(1000 * seed + channels + src_bits + rate // 1000)
    ref = O.Src(rate, bench.RATE_OUT, bench.taps_for(rate), bench.BETA, bench.F_PASS)
    L, M = ref.L, ref.M
    fb_s, fb_d = channels * src_bits // 8, channels * 3
    descs, src_parts, sp, dp = [], [], 0, 0
    for s in range(9):
        in_frames = int(rng.integers(int(0.05 * rate), int(0.6 * rate)))
        out_total = (in_frames * L + M - 1) // M
        x = noise_le(4000 + 97 * seed + s, in_frames * channels, src_bits).reshape(-1)
        first = 0
        while first < out_total:
            n = int(min(out_total - first, rng.integers(1, 701)))
            if rng.random() < 0.03:
                n = 0                                            # an empty message, in the stream's order
            d = np.zeros(1, dtype=capi.SRC_MSG_DESC)
            d["src_offset"], d["src_frame0"], d["src_frames"] = sp, 0, in_frames
            d["out_frame0"], d["dst_offset"], d["n_frames"] = first, dp + first * fb_d, n
            if rng.random() < 0.3 and n > 0:
                d["flags"] = capi.FLAG_RAMP
                a, b = (int(v) for v in rng.choice([0, 1, 5, 8191, 8192, 12345, O.RAMP_MAX - 1, O.RAMP_MAX], size=2))
                d["ramp_start"], d["ramp_end"] = a, b
            else:
                d["ramp_start"] = d["ramp_end"] = O.RAMP_MAX
            d["attenuation"], d["channels"], d["src_bits"], d["src_endian"] = 256, channels, src_bits, capi.ENDIAN_LITTLE
            d["dst_bits"], d["dst_endian"] = 24, capi.ENDIAN_BIG
            descs.append(d)
            first += n
        src_parts.append(x)
        pad = (-x.size) % 16
        src_parts.append(np.zeros(pad, dtype=np.uint8))
        sp += x.size + pad
        dp += out_total * fb_d
    descs = np.concatenate(descs)
    src = np.concatenate(src_parts)
    h = ctx.src_create(L, M, ref.T, ref.coef_q28)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dp)
    ctx.memset(d_dst, 0xA5, dp)
    b = ctx.src_batch(h, descs, src.size, dp)
    try:
        assert ctx.src_kernel_name(b) == "src_mfma_wg_kernel"
        assert ctx.src_plan(b)["block_kernel_out_frames"] > 0 and ctx.src_plan(b)["generic_pieces"] > 0
        ctx.src_run(b, d_src, d_dst)
        ctx.sync()
        got = ctx.download(d_dst, dp)
        want = np.full(dp, 0xA5, dtype=np.uint8)
        assert ref.process_batch(descs.view(O.SRC_MSG_DESC), src, want) == 0
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (int(bad.size), int(bad[0]))
    finally:
        ctx.batch_destroy(b)
        ctx.src_destroy(h)
        ctx.free(d_src)
        ctx.free(d_dst)
