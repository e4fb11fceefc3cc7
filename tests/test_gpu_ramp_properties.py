"""Row a7 (RampApplicator, Msg.cpp:812-899) pinned by what the reference's own suite holds, on the GPU's output DIRECTLY:
  * TestMsg.cpp:1447-1591's checks -- first / last subsample within 2 of the endpoint product, monotone, left == right,
    negative input stays <= 0, a muted ramp gives zeros -- at 8 / 16 / 24 / 32 bit.  No oracle in these tests: the expected
    values are the reference test's own constants and RampArray.h's table (tests/golden/ramp_table_q15.json).
  * the scalar core exhaustively: every 16-bit subsample x every one of the 512 table entries (and the clamp of the index
    above 511), one launch, against the oracle's restatement of Msg.cpp:832-899.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from ohpipeline_amd import capi

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BE = capi.ENDIAN_BIG
kMax, kMin = 1 << 14, 0                                  # Ramp::kMax / kMin
kAudioDataSize = 792                                     # TestMsg.cpp:1448


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=[0, 1], ids=["tuned", "v1"])
def vctx(ctx, request):
    ctx.set_kernel_variant(request.param)
    yield ctx
    ctx.set_kernel_variant(0)


def gpu_ramp(ctx, data, bits, channels, start, end):
    """RampApplicator(ramp).Start(data, bits, channels) + GetNextSample() for every sample, on the device."""
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    n = src.size // (channels * bits // 8)
    d = np.zeros(1, dtype=capi.MSG_DESC)
    d["n_frames"], d["ramp_start"], d["ramp_end"], d["attenuation"] = n, start, end, 256
    d["channels"], d["src_bits"], d["dst_bits"], d["src_endian"], d["dst_endian"] = channels, bits, bits, BE, BE
    d["flags"] = capi.FLAG_RAMP
    d_src, d_dst = ctx.upload(src), ctx.malloc(src.size)
    ctx.memset(d_dst, 0xA5, src.size)
    b = ctx.pcm_batch(d, src.size, src.size)
    ctx.pcm_run(b, d_src, d_dst)
    out = ctx.download(d_dst, src.size)
    ctx.batch_destroy(b)
    ctx.free(d_src)
    ctx.free(d_dst)
    return out


def values(out, bits):
    """(samples, 2) unsigned big-endian subsample values, as the reference test assembles them."""
    sb = bits // 8
    b = out.reshape(-1, 2, sb).astype(np.int64)
    v = np.zeros(b.shape[:2], dtype=np.int64)
    for k in range(sb):
        v = (v << 8) | b[:, :, k]
    return v


def test_ramp_applicator_properties_on_the_gpu_output(vctx):
    table = json.load(open(os.path.join(GOLDEN, "ramp_table_q15.json")))["values"]      # RampArray.h
    audio = bytes([0x7f]) * kAudioDataSize
    # [Max..Min], 8 bit (:1455-1470): starts close to the input, left == right, never rises, ends at zero
    v = values(gpu_ramp(vctx, audio, 8, 2, kMax, kMin), 8)
    assert v[0, 0] >= 0x7d and (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[-1, 0] == 0
    # negative subsamples (:1473-1492): stay <= 0
    v = values(gpu_ramp(vctx, bytes([0xff]) * kAudioDataSize, 8, 2, kMax, kMin), 8)
    assert v[0, 0] >= 0xfd and (((v[:, 0] & 0x80) != 0) | (v[:, 0] == 0)).all()
    assert (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[-1, 0] == 0
    # 16 / 24 / 32 bit (:1494-1531)
    for bits, first in ((16, 0x7f7f), (24, 0x7f7f7f), (32, 0x7f7f7f7f)):
        v = values(gpu_ramp(vctx, audio, bits, 2, kMax, kMin), bits)
        assert v.shape[0] == kAudioDataSize // (2 * bits // 8)
        assert (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[0, 0] <= first
        assert v[-1, 0] == 0                                                           # (the full ramp ends in silence)
        assert first - v[0, 0] <= (2 << (bits - 8))                                    # (and starts within 2 of the top byte)
    # [Min..Max] (:1533-1548)
    v = values(gpu_ramp(vctx, audio, 8, 2, kMin, kMax), 8)
    assert v[0, 0] <= 0x02 and (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) >= 0).all() and v[-1, 0] >= 0x7d
    # [Max..50%], [Min..50%], [50%..25%] (:1550-1591): Ramp::Set's endpoints are the KATs of :1391-1443
    half, quarter = (kMax - kMin) // 2, (kMax - kMin) // 4
    end_guess = (0x7f * table[256]) >> 15
    v = values(gpu_ramp(vctx, audio, 8, 2, kMax, half), 8)
    assert v[0, 0] >= 0x7d and 0 <= end_guess - v[-1, 0] <= 0x02
    v = values(gpu_ramp(vctx, audio, 8, 2, kMin, half), 8)
    assert v[0, 0] <= 0x02 and 0 <= end_guess - v[-1, 0] <= 0x02
    v = values(gpu_ramp(vctx, audio, 8, 2, kMax // 2, quarter), 8)
    assert 0 <= end_guess - v[0, 0] < 0x02 and 0 <= ((0x7f * table[384]) >> 15) - v[-1, 0] <= 0x02


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
def test_a_muted_ramp_is_silence_on_the_gpu(vctx, bits):
    """TestMsg.cpp:1715-1746: Ramp::SetMuted is [Min..Min]; whatever the audio, the playable reads as zeros."""
    rng = np.random.default_rng(bits)
    audio = rng.integers(0, 256, size=768 * (bits // 8), dtype=np.uint8)
    for channels in (1, 2):
        out = gpu_ramp(vctx, audio, bits, channels, kMin, kMin)
        assert out.size == audio.size and (out == 0).all()


def test_every_subsample_times_every_table_entry(vctx):
    """All 65 536 subsample16 values x all 512 multipliers (+ the index clamp above 511): messages whose ramp starts and ends
    at kMax - 32 k sit on table entry k for every frame (Msg.cpp:835-838); 16-bit stereo, messages of at most 9216 bytes
    (DecodedAudio::kMaxBytes)."""
    per_msg = 9216 // 4                                                   # frames of a full message
    every = np.arange(65536, dtype=">u2").view(np.uint8)                  # big-endian 16-bit: 0x0000 .. 0xffff
    n_frames = 65536 // 2
    levels = [kMax - 32 * k for k in range(512)] + [15, 0]                # the last two round up to index 512 -> clamped to 511
    starts = np.arange(0, n_frames, per_msg)
    d = np.zeros(len(levels) * len(starts), dtype=capi.MSG_DESC)
    i = 0
    for li, level in enumerate(levels):
        for s in starts:
            d[i]["src_offset"] = s * 4
            d[i]["dst_offset"] = (li * n_frames + s) * 4
            d[i]["n_frames"] = min(per_msg, n_frames - s)
            d[i]["ramp_start"] = d[i]["ramp_end"] = level
            i += 1
    d["attenuation"], d["channels"], d["src_bits"], d["dst_bits"] = 256, 2, 16, 16
    d["src_endian"], d["dst_endian"], d["flags"] = BE, BE, capi.FLAG_RAMP
    dst_bytes = len(levels) * every.size
    d_src, d_dst = vctx.upload(every), vctx.malloc(dst_bytes)
    vctx.memset(d_dst, 0xA5, dst_bytes)
    b = vctx.pcm_batch(d, every.size, dst_bytes)
    vctx.pcm_run(b, d_src, d_dst)
    got = vctx.download(d_dst, dst_bytes)
    vctx.batch_destroy(b)
    vctx.free(d_src)
    vctx.free(d_dst)
    want = np.full(dst_bytes, 0xA5, dtype=np.uint8)
    assert O.msg_process_batch(d.view(O.MSG_DESC), every, want) == 0
    assert np.array_equal(got, want)
    # and, independent of the oracle's code: (int16 * table[k]) >> 15 from RampArray.h's data
    table = np.array(json.load(open(os.path.join(GOLDEN, "ramp_table_q15.json")))["values"], dtype=np.int64)
    x = np.arange(65536, dtype=np.uint16).astype(np.int16).astype(np.int64)
    g = got.view(">u2").reshape(len(levels), 65536).astype(np.int64)
    for li in (0, 1, 17, 255, 256, 384, 510, 511, 512, 513):
        k = min(li, 511)
        assert np.array_equal(g[li], ((x * table[k]) >> 15) & 0xffff), li
