"""BASELINE config 5 end to end: FLAC streams decoded by the reference's own libFLAC (oracle/_ref, tests/flac_ref.py) into
the planar TInt32 frames CodecFlac::CallbackWrite receives, then ON THE DEVICE: a14 pack (one descriptor per CallbackWrite
chunk, Flac.cpp:379-417) -> resample 44.1 -> 48 kHz -> ramp -> S24 (or, for streams that need no resampling, the PCM
message path: ramp -> format).  Checks: bit-exact against the oracle's composition on the same frames, and the lossless
property -- the device's packed audio is byte for byte the packed form of the PCM that was encoded."""
import numpy as np
import pytest

import flac_workload as FW
import oracle_lib as O
import workloads as W
from ohpipeline_amd import capi

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not FW.ensure_ref(), reason="oracle/_ref/libflac_ref.so not built (needs /root/reference)")]


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def plane_arena(frames, n_streams, rotate):
    """Every stream's decoded audio as planes [stream][channel][frame] (what a host decoder would upload), each stream the
    fixture rotated by a different number of frames so that streams differ.  Returns (planes int32, pcm int32 [stream][frame][ch])."""
    pcm = np.concatenate([f[4] for f in frames], axis=1)                 # [ch][frames]
    planes = np.stack([np.roll(pcm, -s * rotate, axis=1) for s in range(n_streams)])
    return np.ascontiguousarray(planes), np.ascontiguousarray(planes.transpose(0, 2, 1))


def flac_pack_descs(frames, n_streams, n_in, ch, bits):
    """One FLAC_PACK descriptor per CallbackWrite chunk per stream."""
    chunks = FW.callback_write_chunks(frames)
    d = np.zeros(n_streams * len(chunks), dtype=capi.FMT_DESC)
    frame_start = np.cumsum([0] + [f[0] for f in frames])
    k = 0
    for s in range(n_streams):
        for fi, first, n in chunks:
            f0 = int(frame_start[fi]) + first
            d[k]["src_offset"] = (s * ch * n_in + f0) * 4
            d[k]["dst_offset"] = (s * n_in + f0) * ch * (bits // 8)
            d[k]["n_frames"] = n
            k += 1
    d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["src_plane_stride"] = capi.FMT_FLAC_PACK, ch, 32, bits, n_in * 4
    return d


def run_pack(ctx, frames, n_streams, info, rotate=97):
    ch, bits, n_in = info["channels"], info["bits"], info["frames"]
    planes, pcm = plane_arena(frames, n_streams, rotate)
    src = planes.view(np.uint8).reshape(-1)
    packed_bytes = n_streams * n_in * ch * (bits // 8)
    d_planes, d_packed = ctx.upload(src), ctx.malloc(packed_bytes)
    descs = flac_pack_descs(frames, n_streams, n_in, ch, bits)
    b = ctx.fmt_batch(descs, src.size, packed_bytes)
    ctx.fmt_run(b, d_planes, d_packed)
    got = ctx.download(d_packed, packed_bytes)
    ctx.batch_destroy(b)
    ctx.free(d_planes)
    want = FW.pack_be(pcm.reshape(-1, ch), bits)                          # lossless: the packed form of what was encoded
    assert np.array_equal(got, want), "a14 on the decoded frames is not the original audio"
    return d_packed, want, packed_bytes


@pytest.mark.parametrize("name", ["s24_stereo_44k1_b4096_l8", "s24_stereo_44k1_b576_l0", "s16_stereo_44k1_b1152_l5"])
def test_flac_to_resampled_ramped_s24(ctx, name):
    info, _stream, frames, md5_ok = FW.load(name)
    assert md5_ok
    ch, bits, n_in, n_streams = info["channels"], info["bits"], info["frames"], 16
    d_packed, packed_ref, packed_bytes = run_pack(ctx, frames, n_streams, info)
    L, M, coef = capi.src_design(44100, 48000, 32, 9.0, 20000.0)
    ref = O.Src(44100, 48000, 32, 9.0, 20000.0)
    h = ctx.src_create(L, M, 32, coef)
    out_total = ref.out_frames(n_in)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 20 * O.JIFFIES_PER_MS, 40 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, n_in, L, M, 240, ch, bits, O.ENDIAN_BIG, 24, O.ENDIAN_BIG, sched)
    assert sbytes == packed_bytes
    d_out = ctx.malloc(dbytes)
    sb = ctx.src_batch(h, descs, packed_bytes, dbytes)
    ctx.src_run(sb, d_packed, d_out)
    got = ctx.download(d_out, dbytes)
    want = np.zeros(dbytes, dtype=np.uint8)
    assert ref.process_batch(descs, packed_ref, want) == 0
    assert np.array_equal(got, want)
    plan = ctx.src_plan(sb)
    assert plan["block_kernel_out_frames"] > 0.9 * n_streams * out_total      # the fast path carries it
    ctx.batch_destroy(sb); ctx.src_destroy(h); ctx.free(d_packed); ctx.free(d_out)


@pytest.mark.parametrize("name,dst_bits", [("s24_6ch_48k_b4608_l3", 24), ("s24_6ch_48k_b4608_l3", 16), ("s8_mono_8k_b256_l2", 8)])
def test_flac_to_ramped_messages_without_resampling(ctx, name, dst_bits):
    info, _stream, frames, md5_ok = FW.load(name)
    assert md5_ok
    ch, bits, n_in, n_streams = info["channels"], info["bits"], info["frames"], 8
    d_packed, packed_ref, packed_bytes = run_pack(ctx, frames, n_streams, info, rotate=31)
    per_msg = info["rate"] // 200                                         # 5 ms messages
    n_msgs = (n_in + per_msg - 1) // per_msg
    sched = W.ramp_schedule(n_msgs, per_msg * (O.JIFFIES_PER_SEC // info["rate"]), 20 * O.JIFFIES_PER_MS, 40 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes = W.pcm_stream_descs(n_streams, n_in, per_msg, ch, bits, O.ENDIAN_BIG, dst_bits, O.ENDIAN_BIG, sched)[:3]
    assert sbytes == packed_bytes
    d_out = ctx.malloc(dbytes)
    b = ctx.pcm_batch(descs, packed_bytes, dbytes)
    ctx.pcm_run(b, d_packed, d_out)
    got = ctx.download(d_out, dbytes)
    want = np.zeros(dbytes, dtype=np.uint8)
    assert O.msg_process_batch(descs, packed_ref, want) == 0
    assert np.array_equal(got, want)
    ctx.batch_destroy(b); ctx.free(d_packed); ctx.free(d_out)


@pytest.mark.parametrize("name", ["s24_stereo_44k1_b4096_l8", "s24_stereo_44k1_b576_l0", "s16_stereo_44k1_b1152_l5"])
def test_flac_planes_to_resampled_ramped_s24_in_one_pass(ctx, name):
    """Config 5 as bench.py runs it since round 2: ONE batch, ONE pass.  The resampler's descriptors point at the decoder's
    planes (OHGPU_FLAG_SRC_PLANAR32); there is no FLAC_PACK batch, no packed arena.  Expected: the oracle's composition of
    the three rows -- the planes packed as Flac.cpp:379-417 packs them, then resampled and ramped."""
    info, _stream, frames, md5_ok = FW.load(name)
    assert md5_ok
    ch, bits, n_in, n_streams = info["channels"], info["bits"], info["frames"], 16
    planes, pcm = plane_arena(frames, n_streams, 97)
    packed_ref = FW.pack_be(pcm.reshape(-1, ch), bits)
    L, M, coef = capi.src_design(44100, 48000, 32, 9.0, 20000.0)
    ref = O.Src(44100, 48000, 32, 9.0, 20000.0)
    h = ctx.src_create(L, M, 32, coef)
    out_total = ref.out_frames(n_in)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 20 * O.JIFFIES_PER_MS, 40 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, n_in, L, M, 240, ch, bits, O.ENDIAN_BIG, 24, O.ENDIAN_BIG, sched)
    want = np.zeros(dbytes, dtype=np.uint8)
    assert ref.process_batch(descs, packed_ref, want) == 0
    fused = descs.copy().view(capi.SRC_MSG_DESC)
    per_packed, per_planes = n_in * ch * (bits // 8), ch * n_in * 4
    fused["src_offset"] = (descs["src_offset"] // per_packed) * per_planes             # the stream's plane 0
    fused["src_plane_stride"] = n_in * 4
    fused["flags"] |= capi.FLAG_SRC_PLANAR32
    src = planes.view(np.uint8).reshape(-1)
    d_src, d_out = ctx.upload(src), ctx.malloc(dbytes)
    sb = ctx.src_batch(h, fused, src.size, dbytes)
    ctx.src_run(sb, d_src, d_out)                                          # the step's only launch sequence: this batch
    got = ctx.download(d_out, dbytes)
    assert np.array_equal(got, want)
    plan, bi = ctx.src_plan(sb), ctx.batch_info(sb)
    assert plan["block_kernel_out_frames"] > 0.9 * n_streams * out_total   # whole blocks straight from the planes on the block kernel
    assert plan["generic_pieces"] <= 2 * n_streams                         # only a stream's block-unaligned ends are left over
    assert bi["src_bytes_touched"] >= n_streams * n_in * ch * 4 * 0.99     # what is read is the planes (4 bytes a subsample), nothing packed
    ctx.batch_destroy(sb); ctx.src_destroy(h); ctx.free(d_src); ctx.free(d_out)
