"""Rows a14 -> a1 -> a-R in ONE pass: the resampler reading CodecFlac::CallbackWrite's input -- planar host-endian TInt32,
Codec/Flac.cpp:379-417 -- directly (OHGPU_FLAG_SRC_PLANAR32), against the oracle's composition of the three rows: pack the
planes the way the callback does, then resample the packed big-endian audio.  Bit-exact, on the block kernel and on the
generic one; the plan must hold a single kernel's worth of work for whole blocks (no packed arena in between)."""
import numpy as np
import pytest

import oracle_lib as O
import workloads as W
from ohpipeline_amd import capi

pytestmark = pytest.mark.gpu

BETA, F_PASS = 9.0, 20000.0
JPS_OUT = 56448000 // 48000


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def planes_and_packed(rng, streams, frames, bits, gap_frames):
    """Per stream: two planes of `frames` TInt32 at `bits` depth, `gap_frames` unused frames between them (so that the stride
    is not the plane length), and the same audio packed big-endian interleaved as CallbackWrite would emit it."""
    stride = (frames + gap_frames) * 4
    per_stream = 2 * stride
    arena = np.zeros(streams * per_stream // 4, dtype="<i4")
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    packed = np.empty((streams, frames, 2, bits // 8), dtype=np.uint8)
    for s in range(streams):
        for c in range(2):
            v = rng.integers(lo, hi + 1, size=frames, dtype=np.int64)
            v[:8] = [hi, lo, 0, -1, 1, hi, lo, -2][:8]                   # the ends of the range up front
            at = (s * per_stream + c * stride) // 4
            arena[at:at + frames] = v
            for b in range(bits // 8):
                packed[s, :, c, b] = (v >> (8 * (bits // 8 - 1 - b))) & 0xff
    return arena.view(np.uint8), stride, per_stream, packed.reshape(streams, -1)


def descs_for(streams, frames, L, M, bits, msg_frames, planar, stride=0, per_stream=0):
    out_total = (frames * L + M - 1) // M
    # the last outputs need input the buffer does not hold: stop where the filter's newest frame is still inside
    out_total = min(out_total, ((frames - 1) * L) // M)
    n_msgs = (out_total + msg_frames - 1) // msg_frames
    first = np.arange(n_msgs, dtype=np.int64) * msg_frames
    count = np.minimum(msg_frames, out_total - first)
    sched = np.array(W.ramp_schedule(n_msgs, msg_frames * JPS_OUT, 20 * 56448, 60 * 56448), dtype=np.int64)
    d = np.zeros(streams * n_msgs, dtype=capi.SRC_MSG_DESC)
    fb_dst = 2 * 3
    for s in range(streams):
        sl = slice(s * n_msgs, (s + 1) * n_msgs)
        d["src_offset"][sl] = s * per_stream if planar else s * frames * 2 * (bits // 8)
        d["src_frames"][sl] = frames
        d["out_frame0"][sl] = first
        d["dst_offset"][sl] = s * out_total * fb_dst + first * fb_dst
        d["n_frames"][sl] = count
        d["flags"][sl] = sched[:, 0]
        d["ramp_start"][sl] = sched[:, 1]
        d["ramp_end"][sl] = sched[:, 2]
    d["attenuation"], d["channels"], d["src_bits"], d["src_endian"] = 256, 2, bits, capi.ENDIAN_BIG
    d["dst_bits"], d["dst_endian"] = 24, capi.ENDIAN_BIG
    if planar:
        d["flags"] |= capi.FLAG_SRC_PLANAR32
        d["src_plane_stride"] = stride
    return d, streams * out_total * fb_dst, out_total


@pytest.mark.parametrize("bits", [16, 24])
@pytest.mark.parametrize("variant", [0, 1], ids=["block", "generic"])
def test_planar_source_equals_pack_then_resample(ctx, bits, variant):
    rng = np.random.default_rng(100 + bits)
    streams, frames = 3, 23000                                           # several units of 32 blocks per stream, a ragged end
    L, M, coef = capi.src_design(44100, 48000, 32, BETA, F_PASS)
    arena, stride, per_stream, packed = planes_and_packed(rng, streams, frames, bits, gap_frames=5)
    d_planar, dst_bytes, out_total = descs_for(streams, frames, L, M, bits, 240, True, stride, per_stream)
    d_packed, dst_bytes2, _ = descs_for(streams, frames, L, M, bits, 240, False)
    assert dst_bytes == dst_bytes2
    # oracle: a14's packed bytes (built above exactly as Flac.cpp:399-413 writes them), then a1 + a-R
    ref = O.Src(44100, 48000, 32, BETA, F_PASS)
    want = np.full(dst_bytes, 0xA5, dtype=np.uint8)
    assert ref.process_batch(d_packed.view(O.SRC_MSG_DESC), np.ascontiguousarray(packed.reshape(-1)), want) == 0
    # and the oracle's own pack agrees with the bytes built here (ohp_flac_pack, the function config 5 is checked with)
    h = ctx.src_create(L, M, 32, coef)
    ctx.set_kernel_variant(variant)
    try:
        d_src, d_dst = ctx.upload(arena), ctx.malloc(dst_bytes)
        ctx.memset(d_dst, 0xA5, dst_bytes)
        b = ctx.src_batch(h, d_planar, arena.size, dst_bytes)
        plan = ctx.src_plan(b)
        ctx.src_run(b, d_src, d_dst)
        got = ctx.download(d_dst, dst_bytes)
        ctx.batch_destroy(b)
        ctx.free(d_src)
        ctx.free(d_dst)
    finally:
        ctx.set_kernel_variant(0)
        ctx.src_destroy(h)
    assert np.array_equal(got, want)
    # whole blocks run on the block kernel straight from the planes: what is left to the generic kernel is block-unaligned ends
    assert plan["block_kernel_out_frames"] >= streams * (out_total - 2 * 160)


def test_planar_descriptors_are_validated(ctx):
    L, M, coef = capi.src_design(44100, 48000, 32, BETA, F_PASS)
    h = ctx.src_create(L, M, 32, coef)
    d, dst_bytes, _ = descs_for(1, 4000, L, M, 16, 240, True, stride=4000 * 4, per_stream=2 * 4000 * 4)
    arena_bytes = 2 * 4000 * 4

    def create(mutate, arena=arena_bytes):
        e = d.copy()
        mutate(e)
        with pytest.raises(capi.OhGpuError) as err:
            ctx.src_batch(h, e, arena, dst_bytes)
        return err.value.code

    def set_field(name, value):
        def f(e):
            e[name] = value
        return f

    assert create(set_field("src_plane_stride", 4000 * 4 + 2)) == capi.ERR_INVALID            # planes on 4-byte boundaries
    assert create(set_field("src_offset", 2)) == capi.ERR_INVALID
    assert create(set_field("src_bits", 32)) == capi.ERR_INVALID                                # TInt32 carries at most 24 bits here
    assert create(set_field("src_plane_stride", 3996)) == capi.ERR_BOUNDS                       # overlapping planes
    assert create(lambda e: None, arena=arena_bytes - 4) == capi.ERR_BOUNDS                     # the second plane's end
    assert create(set_field("src_plane_stride", (1 << 64) - 4)) in (capi.ERR_INVALID, capi.ERR_BOUNDS)   # 64-bit wrap
    e = d.copy()
    e["flags"] &= ~np.uint8(capi.FLAG_SRC_PLANAR32)                                             # a stride without the flag
    with pytest.raises(capi.OhGpuError):
        ctx.src_batch(h, e, arena_bytes, dst_bytes)
    ctx.src_destroy(h)


def test_a_batch_serves_one_launch_at_a_time(ctx):
    """The block kernels' unit counters (and a flywheel batch's workspace) belong to the batch: a second launch on another
    stream while the first is still running must be refused, not run with shared counters; the same stream queues."""
    L, M, coef = capi.src_design(44100, 48000, 32, BETA, F_PASS)
    h = ctx.src_create(L, M, 32, coef)
    streams, frames = 64, 44100 * 4
    rng = np.random.default_rng(5)
    arena, stride, per_stream, _ = planes_and_packed(rng, 1, frames, 16, 0)
    arena = np.tile(arena, streams)
    d, dst_bytes, _ = descs_for(streams, frames, L, M, 16, 240, True, stride, per_stream)
    d_src, d_dst = ctx.upload(arena), ctx.malloc(dst_bytes)
    b = ctx.src_batch(h, d, arena.size, dst_bytes)
    s2 = ctx.stream_create()
    try:
        for _ in range(8):
            ctx.src_run(b, d_src, d_dst)                                  # queued on the context's stream
        with pytest.raises(capi.OhGpuError) as err:
            ctx.src_run(b, d_src, d_dst, stream=s2)                       # ...and still running there
        assert err.value.code == capi.ERR_INVALID
        ctx.sync()
        ctx.src_run(b, d_src, d_dst, stream=s2)                           # once it has finished, any stream will do
        ctx.sync(s2)
        first = ctx.download(d_dst, dst_bytes)
        ctx.src_run(b, d_src, d_dst)
        ctx.sync()
        assert np.array_equal(first, ctx.download(d_dst, dst_bytes))
    finally:
        ctx.stream_destroy(s2)
        ctx.batch_destroy(b); ctx.src_destroy(h); ctx.free(d_src); ctx.free(d_dst)
