"""FlywheelRamper on the GPU against the CPU oracle, bit for bit, through the C ABI.
The oracle itself is pinned by the reference's known-answer tests (tests/test_oracle_flywheel_kats.py)."""
import numpy as np
import pytest

import oracle_lib as O
import workloads as W

pytestmark = pytest.mark.gpu

JIFFIES_PER_MS = 56448


def jiffies_per_sample(rate):
    jps = O.lib().ohp_jiffies_per_sample(rate)
    assert jps > 0
    return jps


def training_planes(rng, kind, n, channels, rate):
    """planar big-endian 32-bit audio, channels x n samples"""
    t = np.arange(n)
    planes = []
    for c in range(channels):
        if kind == "sine":
            x = 0.7 * np.sin(2 * np.pi * (997.0 + 211 * c) * t / rate + 0.3 * c) + 0.1 * np.sin(2 * np.pi * 5003.0 * t / rate)
            v = np.round(x * (2 ** 31 - 1)).astype(np.int64)
        elif kind == "noise":
            v = rng.integers(-2 ** 31, 2 ** 31, size=n, dtype=np.int64)
        elif kind == "dc":
            v = np.full(n, 0x12345678 * (1 if c % 2 == 0 else -1), dtype=np.int64)
        else:                                                   # full scale square: exercises the 16-bit wrap-around
            v = np.where((t // 3) % 2 == 0, 2 ** 31 - 1, -2 ** 31).astype(np.int64)
        planes.append(v.astype(np.int32).astype(">i4").view(np.uint8))
    return planes


def oracle_ramp(plane_bytes, channel_bytes, in_samples, rate, channels, out_frames, block_frames):
    out = np.zeros(out_frames * channels * 4, dtype=np.uint8)
    rc = O.lib().ohp_flywheel_ramp(plane_bytes.ctypes.data, channel_bytes, in_samples, rate, channels, out_frames,
                                   block_frames, out.ctypes.data)
    assert rc == 0
    return out


@pytest.fixture(scope="module")
def ctx():
    from ohpipeline_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def test_flywheel_batch_matches_oracle(ctx):
    from ohpipeline_amd import capi
    rng = np.random.default_rng(7)
    reqs = []
    src_parts, src_off, dst_off = [], 0, 0
    cases = [(44100, 2, "sine"), (48000, 2, "noise"), (96000, 6, "sine"), (192000, 8, "noise"), (176400, 1, "square"),
             (88200, 10, "sine"), (44100, 2, "dc"), (384000, 2, "sine"), (32000, 3, "noise"), (48000, 2, "square")]
    for k, (rate, ch, kind) in enumerate(cases * 3):
        jps = jiffies_per_sample(rate)
        in_samples = JIFFIES_PER_MS // jps                      # kTrainingJiffies = 1 ms, StarvationRamper.cpp:374
        out_frames = (20 * JIFFIES_PER_MS) // jps               # kRampDownJiffies = 20 ms, :375
        block = JIFFIES_PER_MS // jps                           # kMaxOutputJiffiesBlockSize = 1 ms
        extra = (k % 3) * 4                                     # sometimes "slightly too much data" (FlywheelRamper.cpp:189-194)
        planes = training_planes(rng, kind, in_samples + extra // 4, ch, rate)
        channel_bytes = planes[0].size
        pad = (k * 3) % 5                                       # unaligned arena offsets
        src_parts.append(np.zeros(pad, dtype=np.uint8)); src_off += pad
        blob = np.concatenate(planes)
        reqs.append(dict(src_offset=src_off, channel_bytes=channel_bytes, dst_offset=dst_off + (k % 4), in_samples=in_samples,
                         out_frames=out_frames, block_frames=block, sample_rate=rate, channels=ch, blob=blob))
        src_parts.append(blob); src_off += blob.size
        dst_off += out_frames * ch * 4 + 8
    src = np.concatenate(src_parts)
    descs = np.zeros(len(reqs), dtype=capi.FLYWHEEL_DESC)
    for i, r in enumerate(reqs):
        for f in ("src_offset", "channel_bytes", "dst_offset", "in_samples", "out_frames", "block_frames", "sample_rate", "channels"):
            descs[f][i] = r[f]
    d_src = ctx.upload(src)
    d_dst = ctx.malloc(dst_off)
    ctx.memset(d_dst, 0xEE, dst_off)
    batch = ctx.flywheel_batch(descs, src.size, dst_off)
    ctx.flywheel_run(batch, d_src, d_dst)
    got = ctx.download(d_dst, dst_off)
    want = np.full(dst_off, 0xEE, dtype=np.uint8)
    for r in reqs:
        n = r["out_frames"] * r["channels"] * 4
        want[r["dst_offset"]:r["dst_offset"] + n] = oracle_ramp(r["blob"], r["channel_bytes"], r["in_samples"], r["sample_rate"],
                                                                 r["channels"], r["out_frames"], r["block_frames"])
    assert np.array_equal(got, want)
    assert not np.all(got[reqs[0]["dst_offset"]:reqs[0]["dst_offset"] + 64] == 0)     # the ramp is audio, not silence
    ctx.batch_destroy(batch)
    ctx.free(d_src); ctx.free(d_dst)


def test_flywheel_validation(ctx):
    from ohpipeline_amd import capi
    d = np.zeros(1, dtype=capi.FLYWHEEL_DESC)
    d["channel_bytes"], d["in_samples"], d["out_frames"], d["block_frames"], d["sample_rate"], d["channels"] = 176, 44, 882, 44, 44100, 2
    for field, bad, code in (("channels", 11, capi.ERR_INVALID), ("channels", 0, capi.ERR_INVALID), ("sample_rate", 768000, capi.ERR_INVALID),
                             ("in_samples", 45, capi.ERR_INVALID), ("in_samples", 3, capi.ERR_INVALID), ("block_frames", 0, capi.ERR_INVALID),
                             ("src_offset", 1 << 20, capi.ERR_BOUNDS), ("dst_offset", 1 << 20, capi.ERR_BOUNDS)):
        e = d.copy()
        e[field] = bad
        with pytest.raises(capi.OhGpuError) as ei:
            ctx.flywheel_batch(e, 176 * 2, 882 * 8)
        assert ei.value.code == code, (field, bad)
    b = ctx.flywheel_batch(d, 176 * 2, 882 * 8)
    ctx.batch_destroy(b)
