"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bit-exact for unpack / attenuate / ramp / silence / pack; the resampler is bit-exact against
the oracle's exact-integer model and within +/-1 LSB (S24) of the oracle's fp64 model.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import itertools

import numpy as np
import pytest

import oracle_lib as O
import workloads as W
from ohpipeline_amd import capi

pytestmark = pytest.mark.gpu

LE, BE = O.ENDIAN_LITTLE, O.ENDIAN_BIG
kMax = O.RAMP_MAX


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


def run_pcm(ctx, descs, src, dst_bytes, fill=0xA5):
    """GPU run of a pcm batch; untouched destination bytes keep the fill pattern."""
    d_src = ctx.upload(src if src.size else np.zeros(1, np.uint8))
    d_dst = ctx.malloc(max(dst_bytes, 1))
    ctx.memset(d_dst, fill, max(dst_bytes, 1))
    b = ctx.pcm_batch(descs, src.size, dst_bytes)
    ctx.pcm_run(b, d_src, d_dst)
    out = ctx.download(d_dst, dst_bytes)
    ctx.batch_destroy(b)
    ctx.free(d_src)
    ctx.free(d_dst)
    return out


def oracle_pcm(descs, src, dst_bytes, fill=0xA5):
    dst = np.full(max(dst_bytes, 1), fill, dtype=np.uint8)
    assert O.msg_process_batch(descs, src if src.size else np.zeros(1, np.uint8), dst) == 0
    return dst[:dst_bytes]


def run_src(ctx, src_handle, descs, src, dst_bytes, fill=0xA5):
    d_src = ctx.upload(src)
    d_dst = ctx.malloc(max(dst_bytes, 1))
    ctx.memset(d_dst, fill, max(dst_bytes, 1))
    b = ctx.src_batch(src_handle, descs, src.size, dst_bytes)
    ctx.src_run(b, d_src, d_dst)
    out = ctx.download(d_dst, dst_bytes)
    ctx.batch_destroy(b)
    ctx.free(d_src)
    ctx.free(d_dst)
    return out


def oracle_src(ref, descs, src, dst_bytes, fill=0xA5):
    dst = np.full(max(dst_bytes, 1), fill, dtype=np.uint8)
    assert ref.process_batch(descs, src, dst) == 0
    return dst[:dst_bytes]


RAMPS = [(kMax, 0), (0, kMax), (kMax, 8192), (8191, 8190), (5, 5), (kMax, kMax), (0, 0), (12345, 54), (17, 16001)]


def matrix_descs(rng, depths, chans, counts, src_endians, dst_fmts, odd_offsets=True):
    """One descriptor per combination, each reading its own slice of a shared noise arena."""
    rows, src_parts, src_pos, dst_pos = [], [], 0, 0
    combos = itertools.product(depths, chans, counts, src_endians, dst_fmts)
    for k, (bits, ch, n, se, (db, de)) in enumerate(combos):
        ramp = RAMPS[k % len(RAMPS)]
        enabled = (k % 4) != 3
        pad = int(rng.integers(0, 7)) if odd_offsets else 0      # arbitrary byte alignment (MsgPlayable::iOffset)
        nbytes = n * ch * bits // 8
        src_parts.append(rng.integers(0, 256, size=pad + nbytes, dtype=np.uint8))
        rows.append((src_pos + pad, dst_pos, n, ramp[0], ramp[1], 256, ch, bits, se, db, de,
                     (O.FLAG_RAMP if enabled else 0) | (O.FLAG_ZERO_LSB32 if (k % 5 == 0) else 0)))
        src_pos += pad + nbytes
        dst_pos += n * ch * db // 8 + int(rng.integers(0, 5))
    descs = np.array(rows, dtype=O.MSG_DESC)
    return descs, np.concatenate(src_parts), dst_pos


def test_capi_and_oracle_descriptor_layouts_agree():
    assert capi.MSG_DESC == O.MSG_DESC and capi.SRC_MSG_DESC == O.SRC_MSG_DESC


def test_pcm_matrix_bit_exact(vctx):
    """a1+a7+a11/a12: depth x channels x N x endian x dst format, ramps up/down/flat/disabled, ragged offsets."""
    rng = np.random.default_rng(1234)
    descs, src, dst_bytes = matrix_descs(
        rng, depths=[8, 16, 24, 32], chans=[1, 2, 6, 8], counts=[1, 2, 3, 42, 43, 220],
        src_endians=[LE, BE], dst_fmts=[(8, BE), (16, BE), (24, BE), (32, BE), (16, LE), (24, LE), (32, LE)])
    got = run_pcm(vctx, descs, src, dst_bytes)
    want = oracle_pcm(descs, src, dst_bytes)
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} byte mismatches, first at {bad[:5]}"


def test_pcm_extremes_bit_exact(vctx):
    """Full-scale +/-, all-ones, alternating patterns under every ramp endpoint pair, N up to a full 9216-byte cell."""
    rng = np.random.default_rng(7)
    patterns = [bytes([0x7f, 0xff, 0xff, 0xff]), bytes([0x80, 0, 0, 0]), bytes([0xff] * 4), bytes([0x80, 0x00, 0x7f, 0xff])]
    rows, parts, sp, dp = [], [], 0, 0
    for bits, ch, pat, ramp in itertools.product([8, 16, 24, 32], [2, 6], patterns, RAMPS):
        frame_bytes = ch * bits // 8
        n = O.MAX_BYTES // frame_bytes
        data = np.frombuffer((pat * (n * frame_bytes // 4 + 1))[:n * frame_bytes], dtype=np.uint8)
        parts.append(data)
        rows.append((sp, dp, n, ramp[0], ramp[1], 256, ch, bits, BE, bits, BE, O.FLAG_RAMP))
        sp += data.size
        dp += n * frame_bytes
    descs = np.array(rows, dtype=O.MSG_DESC)
    src = np.concatenate(parts)
    assert np.array_equal(run_pcm(vctx, descs, src, dp), oracle_pcm(descs, src, dp))


@pytest.mark.parametrize("bits,dbits", list(itertools.product([8, 16, 24, 32], [8, 16, 24, 32])))
def test_pcm_uniform_batches_every_depth_pair(vctx, bits, dbits):
    """Uniform batches run the line kernel's depth-specialised instantiations: every (source, destination) depth, both
    byte orders on both sides, 1/2/6 channels, ragged and multi-chunk messages, plain / ramped / silent / zero-LSB, every
    byte misalignment of both arenas."""
    rng = np.random.default_rng(bits * 100 + dbits)
    for se, de, ch in itertools.product([LE, BE], [LE, BE], [1, 2, 6]):
        rows, parts, sp, dp = [], [], 0, 0
        for k, n in enumerate([1, 2, 5, 43, 171, 220, 700, 1537]):
            pad = k % 5
            nbytes = n * ch * bits // 8
            parts.append(rng.integers(0, 256, size=pad + nbytes, dtype=np.uint8))
            flags = [0, O.FLAG_RAMP, 0, O.FLAG_RAMP | O.FLAG_ZERO_LSB32, O.FLAG_SILENCE, 0, O.FLAG_RAMP, O.FLAG_ZERO_LSB32][k]
            ramp = RAMPS[(k + ch) % len(RAMPS)]
            rows.append((sp + pad, dp + (k % 4), n, ramp[0], ramp[1], 256, ch, bits, se, dbits, de, flags))
            sp += pad + nbytes
            dp += n * ch * dbits // 8 + 7
        descs = np.array(rows, dtype=O.MSG_DESC)
        src = np.concatenate(parts)
        got, want = run_pcm(vctx, descs, src, dp), oracle_pcm(descs, src, dp)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{bits}->{dbits} se={se} de={de} ch={ch}: {bad.size} mismatches, first at {bad[:5]}"


@pytest.mark.parametrize("bits,dbits", list(itertools.product([16, 24, 32], [16, 24, 32])))
def test_pcm_ramped_groups_every_channel_count(vctx, bits, dbits):
    """The line kernel's ramped group path (a group of four subsamples inside two frames: two multiplier look-ups, one
    byte shuffle in, one multiply, one shuffle out per subsample) against the oracle's RampApplicator: every channel count
    up to eight, both byte orders on both sides, every ramp shape (down, up, flat, partial), messages of 2 frames (no
    multiplier for the division: general path) to several chunks, ragged tails."""
    rng = np.random.default_rng(7 * bits + dbits)
    for ch, (se, de) in itertools.product([2, 3, 4, 5, 6, 7, 8], [(BE, BE), (LE, BE), (BE, LE), (LE, LE)]):
        rows, parts, sp, dp = [], [], 0, 0
        for k, n in enumerate([2, 3, 4, 7, 64, 219, 240, 513, 2000]):
            if n * ch * max(bits, dbits) // 8 > 65536:
                n = 300
            for ramp in RAMPS:
                pad = (k + ramp[0]) % 5
                nbytes = n * ch * bits // 8
                parts.append(rng.integers(0, 256, size=pad + nbytes, dtype=np.uint8))
                rows.append((sp + pad, dp + (k % 3), n, ramp[0], ramp[1], 256, ch, bits, se, dbits, de, O.FLAG_RAMP))
                sp += pad + nbytes
                dp += n * ch * dbits // 8 + 5
        descs = np.array(rows, dtype=O.MSG_DESC)
        src = np.concatenate(parts)
        got, want = run_pcm(vctx, descs, src, dp), oracle_pcm(descs, src, dp)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{bits}->{dbits} ch={ch} se={se} de={de}: {bad.size} mismatches, first at {bad[:5]}"


@pytest.mark.parametrize("bits,dbits,ch", [(24, 24, 2), (16, 24, 2), (24, 16, 6), (32, 24, 2), (16, 16, 1), (24, 32, 8)])
def test_pcm_contiguous_streams_of_mixed_messages(vctx, bits, dbits, ch):
    """Back-to-back messages of one stream -- what the chunk planner merges -- with plain, ramped, silent and (16-bit)
    attenuated messages of 1..3000 frames in random order, both byte orders, and a second stream that starts unaligned."""
    rng = np.random.default_rng(bits + dbits + ch)
    for se, de in itertools.product([LE, BE], [LE, BE]):
        rows, parts, sp, dp = [], [], 3, 5
        for stream in range(2):
            for k in range(60):
                n = int(rng.choice([1, 2, 3, 4, 5, 43, 220, 240, 1000, 3000]))
                kind = int(rng.integers(0, 10))
                flags = O.FLAG_RAMP if kind < 3 else (O.FLAG_SILENCE if kind == 3 else 0)
                if kind == 4 and dbits == 32:
                    flags |= O.FLAG_ZERO_LSB32
                att = int(rng.choice([256, 256, 100, 0])) if bits == 16 else 256
                ramp = RAMPS[int(rng.integers(0, len(RAMPS)))]
                nbytes = n * ch * bits // 8
                parts.append(rng.integers(0, 256, size=nbytes, dtype=np.uint8))
                rows.append((sp, dp, n, ramp[0], ramp[1], att, ch, bits, se, dbits, de, flags))
                sp += nbytes
                dp += n * ch * dbits // 8
            parts.append(rng.integers(0, 256, size=1, dtype=np.uint8)); sp += 1; dp += 1      # the next stream starts off by one
        descs = np.array(rows, dtype=O.MSG_DESC)
        src = np.concatenate([np.zeros(3, np.uint8)] + parts)
        got, want = run_pcm(vctx, descs, src, dp), oracle_pcm(descs, src, dp)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{bits}->{dbits} ch={ch} se={se} de={de}: {bad.size} mismatches, first at {bad[:5]}"


def test_attenuation_all_s16_values(vctx):
    """a6 over every 16-bit value for att in {0,1,64,255,256} (floor semantics of the unsigned multiply/divide)."""
    vals = np.arange(65536, dtype=np.uint32)
    src = np.stack([(vals >> 8).astype(np.uint8), (vals & 0xff).astype(np.uint8)], axis=1).reshape(-1)
    src_le = np.stack([(vals & 0xff).astype(np.uint8), (vals >> 8).astype(np.uint8)], axis=1).reshape(-1)
    rows = []
    for k, att in enumerate([0, 1, 64, 255, 256, 77]):
        rows.append((0, k * 131072, 32768, kMax, kMax, att, 2, 16, BE, 16, BE, 0))
    descs = np.array(rows, dtype=O.MSG_DESC)
    assert np.array_equal(run_pcm(vctx, descs, src, 6 * 131072), oracle_pcm(descs, src, 6 * 131072))
    descs["src_endian"] = LE                    # attenuation after the LE->BE ingest, with a ramp on top
    descs["flags"] = O.FLAG_RAMP
    descs["ramp_start"], descs["ramp_end"] = 9000, 300
    assert np.array_equal(run_pcm(vctx, descs, src_le, 6 * 131072), oracle_pcm(descs, src_le, 6 * 131072))


def test_silence_including_six_channel_id_bytes(vctx):
    """a9: zeros, the 6-channel id bytes restarting every <=9216-byte chunk, through each depth conversion."""
    rows, dp = [], 0
    for bits, ch, db in itertools.product([8, 16, 24, 32], [1, 2, 6, 8], [8, 16, 24, 32]):
        frame = ch * bits // 8
        n = (2 * O.MAX_BYTES + 100) // frame + 3
        rows.append((0, dp, n, kMax, kMax, 256, ch, bits, BE, db, BE, O.FLAG_SILENCE))
        dp += n * ch * db // 8
    descs = np.array(rows, dtype=O.MSG_DESC)
    src = np.zeros(0, dtype=np.uint8)
    got, want = run_pcm(vctx, descs, src, dp), oracle_pcm(descs, src, dp)
    assert np.array_equal(got, want)
    assert want.any(), "6-channel 32-bit silence must carry channel-id bytes"


def test_empty_and_zero_length(vctx):
    got = run_pcm(vctx, np.zeros(0, dtype=O.MSG_DESC), np.zeros(0, np.uint8), 16)
    assert (got == 0xA5).all()
    descs = np.array([(0, 0, 0, kMax, 0, 256, 2, 24, LE, 24, BE, O.FLAG_RAMP),
                      (0, 4, 1, kMax, 0, 256, 2, 24, LE, 24, BE, O.FLAG_RAMP)], dtype=O.MSG_DESC)
    src = np.arange(6, dtype=np.uint8)
    got, want = run_pcm(vctx, descs, src, 16), oracle_pcm(descs, src, 16)
    assert np.array_equal(got, want) and (got[:4] == 0xA5).all() and (got[10:] == 0xA5).all()


def test_descriptor_validation_errors(ctx):
    """Errors surface as return codes (never a fault on the device): bounds, depth, attenuation, flags."""
    base = (0, 0, 10, kMax, 0, 256, 2, 24, LE, 24, BE, 0)

    def expect(code, **kw):
        d = np.array([base], dtype=O.MSG_DESC)
        for k, v in kw.items():
            d[k] = v
        with pytest.raises(capi.OhGpuError) as e:
            ctx.pcm_batch(d, 60, 60)
        assert e.value.code == code, str(e.value)

    expect(capi.ERR_BOUNDS, n_frames=11)
    expect(capi.ERR_BOUNDS, src_offset=1)
    expect(capi.ERR_BOUNDS, dst_offset=1)
    expect(capi.ERR_BOUNDS, dst_bits=32)
    expect(capi.ERR_INVALID, channels=0)
    expect(capi.ERR_INVALID, channels=9)
    expect(capi.ERR_INVALID, src_bits=12)
    expect(capi.ERR_INVALID, dst_endian=0)
    expect(capi.ERR_INVALID, flags=0x80)
    expect(capi.ERR_INVALID, ramp_start=kMax + 1)
    expect(capi.ERR_UNSUPPORTED, attenuation=128)            # ASSERT(iBitDepth == 16), Msg.cpp:2741
    b = ctx.pcm_batch(np.array([base], dtype=O.MSG_DESC), 60, 60)
    info = ctx.batch_info(b)
    assert info == {"n_msgs": 1, "in_frames": 10, "out_frames": 10, "src_bytes_touched": 60, "dst_bytes_written": 60}
    ctx.batch_destroy(b)


def test_pcm_process_host_round_trip(ctx):
    """The host-buffer convenience entry point (5 ms cadence shape): config 1, S16LE -> ramp -> S24."""
    src = W.noise_pcm(0, 220, 2, 16, LE)
    descs = np.array([(0, 0, 220, 0, 3345, 256, 2, 16, LE, 24, BE, O.FLAG_RAMP)], dtype=O.MSG_DESC)
    dst = np.zeros(220 * 6, dtype=np.uint8)
    ctx.pcm_process_host(descs, src, dst)
    assert np.array_equal(dst, oracle_pcm(descs, src, dst.size))


def test_process_host_keeps_the_bytes_no_message_covers(ctx):
    """The host-buffer calls write the messages' outputs and nothing else: with gaps between the outputs the destination is
    uploaded first, with outputs that tile it exactly (any order) that copy is skipped -- same result either way."""
    rng = np.random.default_rng(12)
    for gap in (0, 5):
        rows, parts, sp, dp = [], [], 0, 0
        for k in range(7):
            n = 100 + 13 * k
            parts.append(rng.integers(0, 256, size=n * 6, dtype=np.uint8))
            rows.append((sp, dp, n, 16384, 0, 256, 2, 24, LE, 24, BE, O.FLAG_RAMP if k % 2 else 0))
            sp += n * 6
            dp += n * 6 + gap
        descs = np.array(rows[::-1], dtype=O.MSG_DESC)                    # (not in destination order)
        src = np.concatenate(parts)
        dst = np.full(dp, 0x5A, dtype=np.uint8)
        ctx.pcm_process_host(descs, src, dst)
        want = np.full(dp, 0x5A, dtype=np.uint8)
        assert O.msg_process_batch(descs, src, want) == 0
        assert np.array_equal(dst, want), gap
    # the resampler's call: one 5 ms message per stream, outputs back to back / with a hole after each
    h, ref = make_src(ctx, 44100, 48000, 32)
    in_frames = 4410
    src = np.concatenate([W.noise_pcm(s, in_frames, 2, 24, LE) for s in range(3)])
    d, sbytes, dbytes, out_total, _ = W.src_stream_descs(3, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, None)
    pick = np.ascontiguousarray(d[5::(out_total + 239) // 240][:3]).copy()      # message 5 of every stream
    for hole in (0, 7):
        pick["dst_offset"] = np.arange(3, dtype=np.uint64) * (240 * 6 + hole)
        total = 3 * (240 * 6 + hole)
        dst = np.full(total, 0x5A, dtype=np.uint8)
        ctx.src_process_host(h, pick, src, dst)
        want = np.full(total, 0x5A, dtype=np.uint8)
        assert ref.process_batch(pick, src, want) == 0
        assert np.array_equal(dst, want), hole
    ctx.src_destroy(h)


def test_config1_stream_s16le_ramp_s24(vctx):
    """BASELINE config 1: 1 stream stereo S16LE 44.1 kHz, 220-frame msgs, up-ramp 50 ms / down-ramp 500 ms, S16->S24."""
    frames = 44100
    n_msgs = (frames + 219) // 220
    sched = W.ramp_schedule(n_msgs, 220 * 1280, 50 * O.JIFFIES_PER_MS, 500 * O.JIFFIES_PER_MS)
    descs, sb, db = W.pcm_stream_descs(1, frames, 220, 2, 16, LE, 24, BE, sched)
    assert sum(1 for s in sched if s[0]) >= 100
    src = W.noise_pcm(0, frames, 2, 16, LE)
    assert np.array_equal(run_pcm(vctx, descs, src, db), oracle_pcm(descs, src, db))


def test_passthrough_identity_and_endian_round_trip(vctx):
    """Size-independent properties: no ramp + same format = byte copy; LE->BE->LE returns the input."""
    n_streams, frames = 64, 44100
    descs, sb, db = W.pcm_stream_descs(n_streams, frames, 220, 2, 24, BE, 24, BE)
    src = np.concatenate([W.noise_pcm(s, frames, 2, 24, BE) for s in range(n_streams)])
    out = run_pcm(vctx, descs, src, db)
    assert np.array_equal(out, src)
    d1, _, _ = W.pcm_stream_descs(n_streams, frames, 220, 2, 24, LE, 24, BE)
    be = run_pcm(vctx, d1, src, db)
    d2, _, _ = W.pcm_stream_descs(n_streams, frames, 220, 2, 24, BE, 24, LE)
    assert np.array_equal(run_pcm(vctx, d2, be, db), src)


# ------------------------------------------------------------------------------------------ resampler
def make_src(ctx, rin, rout, T):
    L_, M_, coef = capi.src_design(rin, rout, T, 9.0, 20000.0)
    ref = O.Src(rin, rout, T, 9.0, 20000.0)
    assert np.array_equal(coef, ref.coef_q28)
    return ctx.src_create(L_, M_, T, coef), ref


def decode_s24_be(buf):
    b = buf.reshape(-1, 3).astype(np.int32)
    v = (b[:, 0] << 16) | (b[:, 1] << 8) | b[:, 2]
    return np.where(v >= 1 << 23, v - (1 << 24), v)


def test_src_config2_one_stream_s24(vctx):
    """BASELINE config 2: 1 stream stereo S24 44.1->48 kHz, resample + ramp + fmt; bit-exact vs the integer
    model, and the un-ramped output within +/-1 LSB of the fp64 model (tolerance stated by north_star)."""
    h, ref = make_src(vctx, 44100, 48000, 32)
    in_frames = 44100
    for kind in ("noise", "sine"):
        src = W.noise_pcm(3, in_frames, 2, 24, LE) if kind == "noise" else W.sine_impulse_pcm(in_frames, 2, 24, LE)
        out_total = ref.out_frames(in_frames)
        n_msgs = (out_total + 239) // 240
        sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * O.JIFFIES_PER_MS, 500 * O.JIFFIES_PER_MS)
        descs, sbytes, dbytes, out_total2, _ = W.src_stream_descs(1, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, sched)
        assert out_total2 == out_total == 48000
        got = run_src(vctx, h, descs, src, dbytes)
        want = oracle_src(ref, descs, src, dbytes)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{kind}: {bad.size} mismatches, first {bad[:5]}"
        plain, _, _, _, _ = W.src_stream_descs(1, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, None)
        y_gpu = decode_s24_be(run_src(vctx, h, plain, src, dbytes))
        worst = 0.0
        for d in plain[:: max(1, plain.size // 40)]:
            err, y64 = ref.process_f64(d, src)
            assert err == 0
            first = int(d["out_frame0"]) * 2
            ideal = np.clip(y64, -8388608.0, 8388607.0)
            worst = max(worst, float(np.abs(y_gpu[first:first + y64.size] - ideal).max()))
        assert worst <= 1.0, f"{kind}: {worst} LSB from the fp64 model"
    vctx.src_destroy(h)


@pytest.mark.parametrize("rin,rout,T,ch,sbits,send,dbits,dend", [
    (96000, 48000, 64, 8, 24, BE, 24, BE),
    (96000, 48000, 64, 6, 16, BE, 32, LE),
    (44100, 48000, 32, 6, 32, LE, 16, BE),
    (44100, 48000, 32, 1, 8, BE, 24, LE),
    (88200, 48000, 24, 2, 24, LE, 32, BE),
    (48000, 44100, 32, 2, 16, LE, 16, LE),
    (32000, 48000, 16, 8, 24, LE, 8, BE),
])
def test_src_formats_bit_exact(vctx, rin, rout, T, ch, sbits, send, dbits, dend):
    h, ref = make_src(vctx, rin, rout, T)
    in_frames = 6000 + 37
    src = np.concatenate([W.noise_pcm(10 + s, in_frames, ch, sbits, send) for s in range(3)])
    out_total = ref.out_frames(in_frames)
    per_msg = 240 if rout == 48000 else 220
    n_msgs = (out_total + per_msg - 1) // per_msg
    jps = O.lib().ohp_jiffies_per_sample(rout)
    sched = W.ramp_schedule(n_msgs, per_msg * jps, 20 * O.JIFFIES_PER_MS, 60 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(3, in_frames, ref.L, ref.M, per_msg, ch, sbits, send, dbits, dend, sched)
    descs["flags"][::7] |= O.FLAG_ZERO_LSB32
    got, want = run_src(vctx, h, descs, src, dbytes), oracle_src(ref, descs, src, dbytes)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}"
    vctx.src_destroy(h)


@pytest.mark.parametrize("ch,send", [(6, LE), (6, BE), (8, LE), (8, BE)])
def test_src_multichannel_block_kernel(ctx, ch, send):
    """6- and 8-channel S24 streams (BASELINE config 4's layouts) on the block kernel: line-aligned streams so that
    every stream's whole blocks take the tuned path, ramps included."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    in_frames, n_streams = 5880, 4                              # 6400 output frames = 40 blocks per stream
    src = np.concatenate([W.noise_pcm(50 + s, in_frames, ch, 24, send) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    assert (out_total * ch * 3) % 64 == 0
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 30 * O.JIFFIES_PER_MS, 50 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, ch, 24, send, 24, BE, sched)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    plan = ctx.src_plan(b)
    assert plan["block_kernel_out_frames"] == n_streams * out_total and plan["generic_pieces"] == 0
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, dbytes)
    want = oracle_src(ref, descs, src, dbytes)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}"
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


@pytest.mark.parametrize("ch", [2, 8])
def test_src_decimate_by_two_block_kernel(ctx, ch):
    """96 -> 48 kHz (L = 1, M = 2) with 64 taps: every second advance emits nothing, the window is 64 deep, the same
    coefficient row serves every output.  BASELINE config 4's other rate, on the block kernel."""
    h, ref = make_src(ctx, 96000, 48000, 64)
    assert ref.L == 1 and ref.M == 2
    in_frames, n_streams = 12800, 3
    src = np.concatenate([W.noise_pcm(70 + s, in_frames, ch, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    assert out_total == 6400
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 30 * O.JIFFIES_PER_MS, 50 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, ch, 24, LE, 24, BE, sched)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    plan = ctx.src_plan(b)
    assert plan["block_kernel_out_frames"] == n_streams * out_total and plan["generic_pieces"] == 0
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, dbytes)
    want = oracle_src(ref, descs, src, dbytes)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}"
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_src_config4_mix_runs_on_the_block_kernel(ctx):
    """BASELINE config 4's deterministic mix -- rate in {44.1k, 96k} -> 48k, channels in {2, 6, 8}, S24, by stream id -- as
    one batch per (filter, layout): every group is planned onto the block kernel and is bit-exact."""
    filters = {44100: make_src(ctx, 44100, 48000, 32), 96000: make_src(ctx, 96000, 48000, 64)}
    groups = {}
    for sid in range(24):
        groups.setdefault(((44100, 96000)[sid % 2], (2, 6, 8)[sid % 3]), []).append(sid)
    assert len(groups) == 6
    for (rate, ch), sids in groups.items():
        h, ref = filters[rate]
        in_frames = 5880 if rate == 44100 else 12800             # 6400 output frames either way
        src = np.concatenate([W.noise_pcm(sid, in_frames, ch, 24, LE) for sid in sids])
        n_msgs = (6400 + 239) // 240
        sched = W.ramp_schedule(n_msgs, 240 * 1176, 30 * O.JIFFIES_PER_MS, 50 * O.JIFFIES_PER_MS)
        descs, sbytes, dbytes, _, _ = W.src_stream_descs(len(sids), in_frames, ref.L, ref.M, 240, ch, 24, LE, 24, BE, sched)
        assert int(descs["n_frames"].max()) * ch * 3 <= O.MAX_BYTES    # <= 9216-byte messages (Msg.h:117)
        d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
        ctx.memset(d_dst, 0xA5, dbytes)
        b = ctx.src_batch(h, descs, src.size, dbytes)
        plan = ctx.src_plan(b)
        assert plan["block_kernel_out_frames"] == len(sids) * 6400 and plan["generic_pieces"] == 0, (rate, ch, plan)
        ctx.src_run(b, d_src, d_dst)
        got = ctx.download(d_dst, dbytes)
        assert np.array_equal(got, oracle_src(ref, descs, src, dbytes)), (rate, ch)
        ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    for h, _ in filters.values():
        ctx.src_destroy(h)


def test_src_one_batch_of_mixed_layouts_runs_on_the_block_kernels(ctx):
    """One resampled batch whose streams differ in channel count, depth and byte order: planned as one uniform batch per layout
    (each on its block kernel where one is instantiated), interleaved in the caller's order, bit-exact; the generic kernel
    (variant 1) agrees."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    layouts = [(2, 24, LE, 24, BE), (6, 24, LE, 24, BE), (2, 16, BE, 24, BE), (8, 24, BE, 24, LE), (2, 24, LE, 24, BE), (2, 32, LE, 32, BE)]
    in_frames, out_frames = 5880, 6400
    n_msgs = (out_frames + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 30 * O.JIFFIES_PER_MS, 50 * O.JIFFIES_PER_MS)
    per_stream, src_parts, sp, dp = [], [], 0, 0
    for sid, (ch, sbits, se, dbits, de) in enumerate(layouts):
        d, sbytes, dbytes, _, _ = W.src_stream_descs(1, in_frames, ref.L, ref.M, 240, ch, sbits, se, dbits, de, sched)
        d = d.copy()
        d["src_offset"] += sp
        d["dst_offset"] += dp
        per_stream.append(d)
        src_parts.append(W.noise_pcm(sid, in_frames, ch, sbits, se))
        assert src_parts[-1].size == sbytes
        sp += sbytes + (-sbytes) % 16
        src_parts.append(np.zeros((-sbytes) % 16, dtype=np.uint8))
        dp += dbytes + 3
    # message k of every stream, then message k + 1: layouts alternate from one descriptor to the next
    descs = np.concatenate([np.stack([d[k] for d in per_stream]) for k in range(n_msgs)])
    src = np.concatenate(src_parts)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dp)
    ctx.memset(d_dst, 0xA5, dp)
    b = ctx.src_batch(h, descs, src.size, dp)
    plan = ctx.src_plan(b)
    # (the block kernels are instantiated for the layouts the configs need -- src_block_common.h: the packed 32-bit stream with a
    # 32-bit output goes to the generic kernel message by message, the other five streams, two of them sharing a layout, to their
    # block kernels -- the eight-channel big-endian stream with little-endian output to the workgroup matrix kernel, which alone has it)
    assert plan["block_kernel_out_frames"] == 5 * out_frames and plan["generic_pieces"] == n_msgs, plan
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, dp)
    want = oracle_src(ref, descs, src, dp)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}"
    ctx.set_kernel_variant(1)                                    # (the generic kernel's own form of the messages is kept by a batch CREATED under variant 1)
    try:
        b1 = ctx.src_batch(h, descs, src.size, dp)
        ctx.memset(d_dst, 0xA5, dp)
        ctx.src_run(b1, d_src, d_dst)
        assert np.array_equal(ctx.download(d_dst, dp), want)
        ctx.batch_destroy(b1)
    finally:
        ctx.set_kernel_variant(0)
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


@pytest.mark.parametrize("rate,taps,ch", [(44100, 32, 6), (96000, 64, 6), (44100, 32, 8), (44100, 32, 2)])
def test_src_workgroups_of_several_waves(ctx, rate, taps, ch):
    """Enough units that every workgroup runs several waves side by side (bench.py's config 4 at full size found what the small
    cases could not: six-channel waves have four lanes that own no block, and their stores walked out of their corner of the
    wave's LDS into the next wave's staging rows)."""
    h, ref = make_src(ctx, rate, 48000, taps)
    n_streams, out_frames = 12, 48000
    in_frames = out_frames * ref.M // ref.L
    src = np.concatenate([W.noise_pcm(700 + s, in_frames, ch, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * O.JIFFIES_PER_MS, 100 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, ch, 24, LE, 24, BE, sched)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    assert ctx.src_plan(b)["generic_pieces"] == 0
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, dbytes)
    assert np.array_equal(got, oracle_src(ref, descs, src, dbytes))
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_src_batch_can_be_run_repeatedly(ctx):
    """A batch is planned once and launched every period: the block kernel's work counters must be back at zero after
    each launch (they reset themselves), so the second and third run write the same bytes as the first."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    in_frames, n_streams = 8820, 40                              # enough units for several claims per wave
    src = np.concatenate([W.noise_pcm(200 + s, in_frames, 2, 24, LE) for s in range(n_streams)])
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, None)
    want = oracle_src(ref, descs, src, dbytes)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    for run in range(3):
        ctx.memset(d_dst, 0xA5 + run, dbytes)
        ctx.src_run(b, d_src, d_dst)
        assert np.array_equal(ctx.download(d_dst, dbytes), want), f"run {run}"
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


@pytest.mark.parametrize("rate,taps,ch", [(44100, 32, 2), (96000, 64, 6), (44100, 32, 8)])
def test_the_same_batch_serves_the_next_period_and_other_ramps(ctx, rate, taps, ch):
    """A bulk caller's period after period: the same streams, the same tiling, a whole number of blocks on, new audio in the same
    arena -- ohgpu_src_batch_advance checks that the batch is of that kind and the plan is run again as it stands; other ramp
    endpoints on the same messages go in with ohgpu_src_batch_set_ramps.  Every period against the oracle run on that period's
    descriptors; a batch that holds a stream's start refuses to advance."""
    h, ref = make_src(ctx, rate, 48000, taps)
    n_streams, period_blocks = 9, 23
    fb = ch * 3
    whole = [W.noise_pcm(900 + s, 200000, ch, 24, LE) for s in range(n_streams)]

    # geometry from the library
    probe, sbytes, dbytes, out_total, n_msgs = W.src_stream_descs(n_streams, 40000, ref.L, ref.M, 240, ch, 24, LE, 24, BE, None)
    b0 = ctx.src_batch(h, probe, sbytes, dbytes)
    L_blk, M_blk = ctx.src_batch_block(b0)
    with pytest.raises(capi.OhGpuError) as e:                   # (it starts its streams: zeros in front of block 0 are not history)
        ctx.src_batch_advance(b0, 1)
    assert e.value.code == capi.ERR_INVALID
    ctx.batch_destroy(b0)
    out_frames = period_blocks * L_blk
    hist = M_blk + taps                                         # input frames held in front of the period's first
    win = period_blocks * M_blk + hist

    def make(first_block, ramp_pair):
        n_m = (out_frames + 239) // 240
        d = np.zeros(n_streams * n_m, dtype=O.SRC_MSG_DESC)
        first = np.arange(n_m) * 240
        for s in range(n_streams):
            sl = slice(s * n_m, (s + 1) * n_m)
            d["src_offset"][sl] = s * win * fb
            d["src_frame0"][sl] = first_block * M_blk - hist
            d["src_frames"][sl] = win
            d["out_frame0"][sl] = first_block * L_blk + first
            d["dst_offset"][sl] = (s * out_frames + first) * fb
            d["n_frames"][sl] = np.minimum(240, out_frames - first)
        d["attenuation"], d["channels"], d["src_bits"], d["src_endian"], d["dst_bits"], d["dst_endian"] = 256, ch, 24, LE, 24, BE
        d["ramp_start"], d["ramp_end"] = O.RAMP_MAX, O.RAMP_MAX
        ramped = (np.arange(d.size) % 7) < 2                    # two messages in seven carry a ramp
        d["flags"] = np.where(ramped, O.FLAG_RAMP, 0)
        d["ramp_start"] = np.where(ramped, ramp_pair[0], d["ramp_start"])
        d["ramp_end"] = np.where(ramped, ramp_pair[1], d["ramp_end"])
        arena = np.concatenate([w[(first_block * M_blk - hist) * fb:(first_block * M_blk - hist + win) * fb] for w in whole])
        return d, arena

    d1, arena1 = make(3, (16384, 2000))
    dbytes = n_streams * out_frames * fb
    d_src, d_dst = ctx.upload(arena1), ctx.malloc(dbytes)
    b = ctx.src_batch(h, d1, arena1.size, dbytes)
    assert ctx.src_plan(b)["block_kernel_out_frames"] > 0
    try:
        ctx.memset(d_dst, 0xA5, dbytes)
        ctx.src_run(b, d_src, d_dst)
        assert np.array_equal(ctx.download(d_dst, dbytes), oracle_src(ref, d1, arena1, dbytes))
        at = 3
        for step, ramps in ((period_blocks, None), (5, (100, 16000)), (period_blocks, (7000, 7000))):
            at += step
            ctx.src_batch_advance(b, step)
            if ramps is not None:
                dn, arena = make(at, ramps)
                ctx.src_batch_set_ramps(b, dn["ramp_start"], dn["ramp_end"])
            else:
                dn, arena = make(at, (16384, 2000))
            ctx.free(d_src)
            d_src = ctx.upload(arena)
            ctx.memset(d_dst, 0xA5, dbytes)
            ctx.src_run(b, d_src, d_dst)
            assert np.array_equal(ctx.download(d_dst, dbytes), oracle_src(ref, dn, arena, dbytes)), (at, ramps)
        with pytest.raises(capi.OhGpuError):
            ctx.src_batch_set_ramps(b, dn["ramp_start"][:-1], dn["ramp_end"][:-1])          # one pair per message, no fewer
        bad = dn["ramp_start"].copy()
        bad[0] = 20000                                            # (message 0 is ramped: beyond Ramp::kMax)
        with pytest.raises(capi.OhGpuError):
            ctx.src_batch_set_ramps(b, bad, dn["ramp_end"])
    finally:
        ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
        ctx.src_destroy(h)


@pytest.mark.parametrize("created_under", [0, 4, 2])
def test_a_batch_created_under_one_variant_runs_under_every_other(ctx, created_under):
    """The variant in force when a batch is RUN chooses among the kernels its plan serves (one decision for the launch and for the
    name: src_kernel_choice); whatever it chooses, the bytes are the oracle's.  Stereo S24, ramped messages included, created under
    the default / the lean kernel's plan / round 1's, run under every variant."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    in_frames, n_streams = 8820, 24
    src = np.concatenate([W.noise_pcm(300 + s, in_frames, 2, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    sched = W.ramp_schedule((out_total + 239) // 240, 240 * 1176, 20 * O.JIFFIES_PER_MS, 40 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, sched)
    want = oracle_src(ref, descs, src, dbytes)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    try:
        ctx.set_kernel_variant(created_under)
        b = ctx.src_batch(h, descs, src.size, dbytes)
        names = {}
        for run_under in (0, 2, 3, 4, 5):
            ctx.set_kernel_variant(run_under)
            names[run_under] = ctx.src_kernel_name(b)
            ctx.memset(d_dst, 0xA5, dbytes)
            ctx.src_run(b, d_src, d_dst)
            assert np.array_equal(ctx.download(d_dst, dbytes), want), (created_under, run_under, names[run_under])
        # the generic kernel needs its own form of every message, which only a batch created under variant 1 keeps: refused, not guessed
        ctx.set_kernel_variant(1)
        assert ctx.src_kernel_name(b) == "src_kernel_v1"
        with pytest.raises(capi.OhGpuError) as e:
            ctx.src_run(b, d_src, d_dst)
        assert e.value.code == capi.ERR_UNSUPPORTED
        b1 = ctx.src_batch(h, descs, src.size, dbytes)
        ctx.memset(d_dst, 0xA5, dbytes)
        ctx.src_run(b1, d_src, d_dst)
        assert np.array_equal(ctx.download(d_dst, dbytes), want)
        ctx.batch_destroy(b1)
        # (rounds 1's block kernel and round 4's unit-per-wave matrix kernel are retired from the shipped library: variants 2 and 5,
        # like 3 and 4, run the lean kernel on a plan the workgroup kernel does not take)
        if created_under == 0:
            assert names[0] == "src_mfma_wg_kernel" and all(names[v] == "src_lean_kernel" for v in (2, 3, 4, 5)), names
        else:
            assert all(names[v] == "src_lean_kernel" for v in (0, 2, 3, 4, 5)), names
        ctx.batch_destroy(b)
    finally:
        ctx.set_kernel_variant(0)
        ctx.free(d_src); ctx.free(d_dst)
        ctx.src_destroy(h)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_src_block_kernel_irregular_message_tilings(ctx, seed):
    """The block kernel under message layouts the bench never produces: messages of 1..700 frames in random order of
    size (a unit then spans anything from 8 to more than 32 messages -- beyond the LDS message table), every message
    ramped with its own endpoints or not at all, streams that start and end inside a block, some messages missing (the
    output range falls into several segments), 64-byte aligned and unaligned stream bases."""
    rng = np.random.default_rng(seed)
    h, ref = make_src(ctx, 44100, 48000, 32)
    rows, src_parts, s_pos, d_pos = [], [], 0, 0
    for stream in range(6):
        in_frames = int(rng.integers(3000, 9000))
        out_total = ref.out_frames(in_frames)
        src_parts.append(W.noise_pcm(300 + 10 * seed + stream, in_frames, 2, 24, LE))
        d_base = d_pos if stream % 2 else (d_pos + 63) // 64 * 64       # odd streams: whatever follows (the generic kernel's case)
        m = 0
        small = stream == seed % 6                                      # one stream of tiny messages: > 32 per unit
        while m < out_total:
            n = int(rng.integers(1, 9)) if small else int(rng.choice([1, 5, 43, 220, 240, 241, 700]))
            n = min(n, out_total - m)
            ramp = RAMPS[int(rng.integers(0, len(RAMPS)))]
            flags = O.FLAG_RAMP if rng.random() < 0.6 else 0
            if rng.random() > 0.03:                                     # (a few messages are simply not asked for)
                rows.append((s_pos, 0, in_frames, m, d_base + m * 6, n, ramp[0], ramp[1], 256, 2, 24, LE, 24, BE, flags, 0))
            m += n
        s_pos += in_frames * 6
        d_pos = d_base + out_total * 6 + int(rng.integers(0, 9))
    descs = np.array(rows, dtype=O.SRC_MSG_DESC)
    order = rng.permutation(descs.size)                                # the batch need not be sorted
    descs = descs[order]
    src = np.concatenate(src_parts)
    d_src, d_dst = ctx.upload(src), ctx.malloc(d_pos)
    ctx.memset(d_dst, 0xA5, d_pos)
    b = ctx.src_batch(h, descs, src.size, d_pos)
    plan = ctx.src_plan(b)
    assert plan["block_kernel_out_frames"] > 0 and plan["generic_pieces"] > 0     # both kernels take part
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, d_pos)
    want = oracle_src(ref, descs, src, d_pos)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"seed {seed}: {bad.size} mismatches, first {bad[:5]}"
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_src_downsampling_block_kernel(ctx):
    """48 -> 44.1 kHz (L = 147, M = 160): blocks of 32 phase periods (4704 outputs from 5120 inputs), some advances emit no
    output, 220-frame messages; on the block kernel."""
    h, ref = make_src(ctx, 48000, 44100, 32)
    assert (ref.L, ref.M) == (147, 160)
    in_frames, n_streams = 5120 * 3, 3
    src = np.concatenate([W.noise_pcm(400 + s, in_frames, 2, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    assert out_total == 4704 * 3 and (out_total * 6) % 64 == 0
    n_msgs = (out_total + 219) // 220
    sched = W.ramp_schedule(n_msgs, 220 * 1280, 30 * O.JIFFIES_PER_MS, 50 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 220, 2, 24, LE, 24, BE, sched)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    plan = ctx.src_plan(b)
    assert plan["block_kernel_out_frames"] == n_streams * out_total, plan
    ctx.src_run(b, d_src, d_dst)
    got = ctx.download(d_dst, dbytes)
    want = oracle_src(ref, descs, src, dbytes)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} mismatches, first {bad[:5]}"
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_src_block_kernel_takes_streams_at_any_destination_alignment(ctx):
    """The destination of a stream need not start on a 64-byte line for its whole blocks to run on the block kernel."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    in_frames = 5880                                               # 6400 output frames = 40 whole blocks
    src = np.concatenate([W.noise_pcm(500 + s, in_frames, 2, 24, LE) for s in range(4)])
    descs, sbytes, dbytes, out_total, n_msgs = W.src_stream_descs(4, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, None)
    shift = np.repeat(np.array([1, 7, 32, 63], dtype=np.uint64), n_msgs) + np.repeat(np.arange(4, dtype=np.uint64) * 64, n_msgs)
    descs["dst_offset"] += shift
    dbytes += 4 * 64 + 64
    d_src, d_dst = ctx.upload(src), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    b = ctx.src_batch(h, descs, src.size, dbytes)
    plan = ctx.src_plan(b)
    assert plan["block_kernel_out_frames"] == 4 * out_total and plan["generic_pieces"] == 0, plan
    ctx.src_run(b, d_src, d_dst)
    assert np.array_equal(ctx.download(d_dst, dbytes), oracle_src(ref, descs, src, dbytes))
    ctx.batch_destroy(b); ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_src_chunked_streaming_equals_whole(vctx):
    """Cross-chunk state = (T-1 frames of history, phase): feeding windows with src_frame0 > 0 gives the same bytes."""
    h, ref = make_src(vctx, 44100, 48000, 32)
    in_frames, ch = 22050, 2
    src = W.noise_pcm(99, in_frames, ch, 24, LE)
    whole, sbytes, dbytes, out_total, n_msgs = W.src_stream_descs(1, in_frames, ref.L, ref.M, 240, ch, 24, LE, 24, BE, None)
    want = run_src(vctx, h, whole, src, dbytes)
    chunked = whole.copy()
    for d in chunked:
        m0, n = int(d["out_frame0"]), int(d["n_frames"])
        lo = max(0, (m0 * ref.M) // ref.L - (ref.T - 1))
        hi = ((m0 + n - 1) * ref.M) // ref.L
        d["src_frame0"], d["src_frames"], d["src_offset"] = lo, hi - lo + 1, lo * ch * 3
    got = run_src(vctx, h, chunked, src, dbytes)
    assert np.array_equal(got, want)
    bad = chunked[5:6].copy()
    bad["src_frame0"] += 1
    bad["src_offset"] += ch * 3
    bad["src_frames"] -= 1
    with pytest.raises(capi.OhGpuError) as e:
        vctx.src_batch(h, bad, src.size, dbytes)
    assert e.value.code == capi.ERR_BOUNDS
    vctx.src_destroy(h)


def test_src_time_shift_invariance(vctx):
    """Size-independent exact property: delaying the input by M frames delays the output by L frames."""
    h, ref = make_src(vctx, 44100, 48000, 32)
    ch, in_frames = 2, 147 * 300
    x = W.noise_pcm(5, in_frames, ch, 24, LE)
    shifted = np.concatenate([np.zeros(ref.M * ch * 3, np.uint8), x])
    d0, _, db0, out0, _ = W.src_stream_descs(1, in_frames, ref.L, ref.M, 240, ch, 24, LE, 24, BE, None)
    d1, _, db1, out1, _ = W.src_stream_descs(1, in_frames + ref.M, ref.L, ref.M, 240, ch, 24, LE, 24, BE, None)
    y0 = run_src(vctx, h, d0, x, db0)
    y1 = run_src(vctx, h, d1, shifted, db1)
    assert out1 == out0 + ref.L
    assert not y1[: ref.L * ch * 3].any()
    assert np.array_equal(y1[ref.L * ch * 3:], y0)
    vctx.src_destroy(h)


def test_src_config3_shape_256_streams(ctx):
    """BASELINE config 3 shape at a size the oracle finishes in seconds: 256 stereo S24 streams, 0.5 s each."""
    h, ref = make_src(ctx, 44100, 48000, 32)
    n_streams, in_frames = 256, 22050
    src = np.concatenate([W.noise_pcm(s, in_frames, 2, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * O.JIFFIES_PER_MS, 200 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, sched)
    got = run_src(ctx, h, descs, src, dbytes)
    want = oracle_src(ref, descs, src, dbytes)
    assert np.array_equal(got, want)
    ctx.src_destroy(h)


def test_src_groups_overlapped_on_two_streams(ctx):
    """The bulk caller's pattern (bench.py's end_to_end.overlapped): pinned buffers, one batch per group
    of streams, uploads on one HIP stream, launch + download on a second one that waits for each group's upload
    (ohgpu_stream_wait_event).  The bytes that come back are the oracle's."""
    import ctypes as C
    h, ref = make_src(ctx, 44100, 48000, 32)
    n_streams, in_frames, groups = 32, 22050, 4
    src = np.concatenate([W.noise_pcm(s, in_frames, 2, 24, LE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * O.JIFFIES_PER_MS, 200 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, 2, 24, LE, 24, BE, sched)
    assert sbytes == src.size and len(descs) == n_streams * n_msgs
    per, sb, db = n_streams // groups, in_frames * 6, out_total * 6
    h_src, h_dst = ctx.malloc_host(sbytes), ctx.malloc_host(dbytes)
    h_src[:] = src
    h_dst[:] = 0xA5
    d_src, d_dst = ctx.malloc(sbytes), ctx.malloc(dbytes)
    ctx.memset(d_dst, 0xA5, dbytes)
    ctx.sync()
    parts = [ctx.src_batch(h, descs[g * per * n_msgs:(g + 1) * per * n_msgs], sbytes, dbytes) for g in range(groups)]
    up, down = ctx.stream_create(), ctx.stream_create()
    arrived = [ctx.event() for _ in range(groups)]
    for g in range(groups):
        s0, s1, o0, o1 = g * per * sb, (g + 1) * per * sb, g * per * db, (g + 1) * per * db
        ctx.copy_h2d(C.c_void_p(d_src.value + s0), h_src[s0:s1], up)
        ctx.record(arrived[g], up)
        ctx.wait_event(down, arrived[g])
        ctx.src_run(parts[g], d_src, d_dst, down)
        ctx.copy_d2h(h_dst[o0:o1], C.c_void_p(d_dst.value + o0), down)
    ctx.sync(up)
    ctx.sync(down)
    got = np.array(h_dst)
    want = oracle_src(ref, descs, src, dbytes)
    assert np.array_equal(got, want)
    with pytest.raises(capi.OhGpuError):
        ctx.wait_event(down, None)
    for b in parts:
        ctx.batch_destroy(b)
    ctx.stream_destroy(up)
    ctx.stream_destroy(down)
    ctx.free_host(h_src)
    ctx.free_host(h_dst)
    ctx.free(d_src)
    ctx.free(d_dst)
    ctx.src_destroy(h)


@pytest.mark.parametrize("layout", ["stereo_s24", "six_s24", "halfband_stereo", "halfband_eight", "mono_s16", "stereo_s32", "planar16", "five_s24"])
@pytest.mark.parametrize("seed", [11, 12])
def test_long_row_units_match_the_oracle(ctx, layout, seed):
    """The unit schedule of large batches -- rows of several consecutive blocks, claimed before the one-block units (round 3,
    src_plan.cpp) -- on small inputs: ohgpu_set_kernel_variant(3) forces it.  Streams long enough for several units, sparse
    ramped messages (a ramped unit stays one block long, so long and short units alternate inside a stream), ragged stream
    ends, one stream that starts mid-way (its first block is not block 0), unsorted descriptors.  Bit-exact against the oracle,
    and equal to what the default schedule writes."""
    rng = np.random.default_rng(seed)
    rate, taps, ch, bits, planar = {"stereo_s24": (44100, 32, 2, 24, False), "six_s24": (44100, 32, 6, 24, False),
                                    "halfband_stereo": (96000, 64, 2, 24, False), "halfband_eight": (96000, 64, 8, 24, False),
                                    "mono_s16": (44100, 32, 1, 16, False), "stereo_s32": (44100, 32, 2, 32, False),
                                    "planar16": (44100, 32, 2, 16, True), "five_s24": (44100, 32, 5, 24, False)}[layout]
    h, ref = make_src(ctx, rate, 48000, taps)
    fb_src, fb_dst = (4 if planar else ch * bits // 8), ch * 3
    rows, src_parts, s_pos, d_pos = [], [], 0, 0
    n_streams = 5 if ch <= 2 else 3
    for stream in range(n_streams):
        # (ten units or so per stream whatever the layout: a mono unit is 64 rows of 320 outputs)
        in_frames = int(rng.integers(38000, 56000)) * (2 if rate == 96000 else 1) * (5 if ch == 1 else 1)
        out_total = ref.out_frames(in_frames)
        if planar:                                                       # the decoder's TInt32 planes, channel after channel
            v = W.noise_pcm(700 + 10 * seed + stream, in_frames, ch, bits, LE).view("<i2").astype(np.int32).reshape(in_frames, ch)
            src_parts.append((np.ascontiguousarray(v.T).reshape(-1).view(np.uint8), v.reshape(-1).astype(">i2").view(np.uint8)))
        else:
            x = W.noise_pcm(700 + 10 * seed + stream, in_frames, ch, bits, LE)
            src_parts.append((x, x))
        d_base = (d_pos + 63) // 64 * 64
        m = int(rng.integers(400, 1200)) if stream == 1 else 0           # one stream is asked for from the middle on
        first = m
        while m < out_total:
            n = min(int(rng.choice([240, 240, 240, 480, 200])), out_total - m)
            ramp = RAMPS[int(rng.integers(0, len(RAMPS)))]
            # sparse ramps: long runs of plain units between them (and none in the first half of stream 0: one run long enough
            # for a long unit whatever the seed)
            flags = O.FLAG_RAMP if rng.random() < 0.008 and not (stream == 0 and m < out_total // 2) else 0
            src_off = s_pos
            rows.append((src_off, 0, in_frames, m, d_base + (m - first) * fb_dst, n, ramp[0], ramp[1], 256, ch, bits, LE, 24, BE,
                         flags | (capi.FLAG_SRC_PLANAR32 if planar else 0), in_frames * 4 if planar else 0))
            m += n
        s_pos += in_frames * (fb_src * ch if planar else fb_src)
        d_pos = d_base + (out_total - first) * fb_dst
    descs = np.array(rows, dtype=O.SRC_MSG_DESC)[rng.permutation(len(rows))]
    src = np.concatenate([a for a, _ in src_parts])
    d_src, d_dst = ctx.upload(src), ctx.malloc(d_pos)
    outs = []
    for variant in (3, 0):
        ctx.set_kernel_variant(variant)
        try:
            ctx.memset(d_dst, 0xA5, d_pos)
            b = ctx.src_batch(h, descs, src.size, d_pos)
            units = ctx.src_units(b)
            assert units["units"] > 0
            assert (units["long_units"] > 0) == (variant == 3), (variant, units)     # the forced schedule cut long units, the default did not
            ctx.src_run(b, d_src, d_dst)
            outs.append(ctx.download(d_dst, d_pos))
            ctx.batch_destroy(b)
        finally:
            ctx.set_kernel_variant(0)
    if planar:                                                           # the oracle's composition: the planes packed, then resampled
        od = descs.copy()
        per = np.cumsum([0] + [b.size for _, b in src_parts])
        starts = np.cumsum([0] + [a.size for a, _ in src_parts])
        for k in range(len(src_parts)):
            sel = od["src_offset"] == starts[k]
            od["src_offset"][sel] = per[k]
        od["flags"] &= ~np.uint8(capi.FLAG_SRC_PLANAR32)
        od["src_plane_stride"] = 0
        od["src_endian"] = BE
        want = oracle_src(ref, od, np.concatenate([b for _, b in src_parts]), d_pos)
    else:
        want = oracle_src(ref, descs, src, d_pos)
    bad = np.nonzero(outs[0] != want)[0]
    assert bad.size == 0, f"{layout} seed {seed}: {bad.size} mismatches, first {bad[:5]}"
    assert np.array_equal(outs[0], outs[1])
    ctx.free(d_src); ctx.free(d_dst)
    ctx.src_destroy(h)


def test_destroyed_batches_give_their_device_blocks_to_the_next(vctx):
    """A context keeps the blocks of destroyed pcm / fmt / flywheel batches (descriptors, plan arrays) by size class and hands them to
    the next batch of that size: after the first creation a loop of create -> run -> destroy allocates nothing on the device
    (ohgpu_device_allocations stops moving), the outputs stay the oracle's, and a LARGER batch still gets what it needs."""
    rng = np.random.default_rng(5)
    descs, src, dst_bytes = matrix_descs(rng, [16, 24], [2, 6], [43, 220], [LE, BE], [(24, BE), (16, LE)])
    want = oracle_pcm(descs, src, dst_bytes)
    assert np.array_equal(run_pcm(vctx, descs, src, dst_bytes), want)
    after_first = vctx.device_allocations()
    for _ in range(20):
        assert np.array_equal(run_pcm(vctx, descs, src, dst_bytes), want)
    assert vctx.device_allocations() == after_first
    big = np.concatenate([descs] * 40)                            # another size class
    assert np.array_equal(run_pcm(vctx, big, src, dst_bytes), want)
    grown = vctx.device_allocations()
    assert grown >= after_first                                   # (more, unless an earlier test of this context left blocks of that class behind)
    assert np.array_equal(run_pcm(vctx, big, src, dst_bytes), want)
    assert np.array_equal(run_pcm(vctx, descs, src, dst_bytes), want)
    assert vctx.device_allocations() == grown
