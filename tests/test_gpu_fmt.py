"""GPU parity for the layout-changing processors against the oracle's restatement
of FlywheelInput::DoProcessFragment, Sender::DoProcessFragment and CodecFlac::CallbackWrite."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ohpipeline_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def run_fmt(ctx, descs, src, dst_bytes):
    d_src, d_dst = ctx.upload(src), ctx.malloc(dst_bytes)
    ctx.memset(d_dst, 0xA5, dst_bytes)
    b = ctx.fmt_batch(descs, src.size, dst_bytes)
    ctx.fmt_run(b, d_src, d_dst)
    out = ctx.download(d_dst, dst_bytes)
    ctx.batch_destroy(b)
    ctx.free(d_src)
    ctx.free(d_dst)
    return out


def test_flywheel_unpack_planar(ctx):
    """a11: packed BE interleaved 1/2/3/4 bytes -> planar 4-byte BE left-justified (StarvationRamper.cpp:117-186)."""
    rng = np.random.default_rng(3)
    for sb, ch, n in [(1, 2, 44), (2, 2, 44), (3, 2, 44), (4, 2, 44), (3, 8, 192), (2, 10, 7), (3, 1, 1),
                      (3, 2, 45), (2, 2, 3), (4, 2, 1), (3, 2, 1027), (2, 2, 4098)]:      # stereo kernels: tails of 1..3 frames
        src = rng.integers(0, 256, size=n * ch * sb, dtype=np.uint8)
        stride = n * 4 + 8
        d = np.zeros(1, dtype=capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["n_frames"], d["dst_plane_stride"] = capi.FMT_UNPACK_PLANAR, ch, sb * 8, n, stride
        got = run_fmt(ctx, d, src, ch * stride)
        want = np.full(ch * stride, 0xA5, dtype=np.uint8)
        pos = np.zeros(ch, dtype=np.uint32)
        assert O.lib().ohp_flywheel_unpack(src.ctypes.data_as(C.c_void_p), src.size, ch, sb, want.ctypes.data_as(C.c_void_p),
                                           stride, pos.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        assert np.array_equal(got, want), (sb, ch, n)


def test_sender_pack(ctx):
    """a13: first two channels (from channel 8 when >= 10), min(bytes, 3) MSBs each (Sender.cpp:351-377)."""
    rng = np.random.default_rng(4)
    for sb, ch, n in [(2, 2, 220), (3, 2, 240), (4, 2, 240), (3, 6, 100), (3, 8, 100), (4, 10, 33), (1, 2, 5)]:
        src = rng.integers(0, 256, size=n * ch * sb, dtype=np.uint8)
        out_bytes = n * 2 * min(sb, 3)
        d = np.zeros(1, dtype=capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["n_frames"] = capi.FMT_SENDER_PACK, ch, sb * 8, n
        got = run_fmt(ctx, d, src, out_bytes)
        want = np.zeros(out_bytes, dtype=np.uint8)
        nb = C.c_uint32(0)
        assert O.lib().ohp_sender_pack(src.ctypes.data_as(C.c_void_p), src.size, ch, sb, want.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
        assert nb.value == out_bytes and np.array_equal(got, want), (sb, ch, n)


def test_sender_pack_batches_of_wider_streams(ctx):
    """Batches in which every pack drops channels run ohm_wide_kernel's plain path (a lane per two frames): every depth,
    3..10 channels, one frame to several rounds of 256, odd source and destination offsets, the last descriptor ending with
    the source arena (its last frames are read byte by byte), and untouched bytes between the outputs."""
    rng = np.random.default_rng(41)
    for sb in (1, 2, 3, 4):
        rows, parts, sp, dp, want_parts = [], [], 0, 0, []
        for k, (ch, n) in enumerate([(3, 1), (6, 240), (8, 241), (10, 33), (4, 1000), (5, 2), (7, 513), (6, 3), (3, 77)]):
            pad = k % 4
            a = rng.integers(0, 256, size=n * ch * sb, dtype=np.uint8)
            parts += [rng.integers(0, 256, size=pad, dtype=np.uint8), a]
            out_bytes = n * 2 * min(sb, 3)
            rows.append((sp + pad, dp + (k % 3), n, ch, sb * 8))
            w = np.zeros(out_bytes, dtype=np.uint8)
            nb = C.c_uint32(0)
            assert O.lib().ohp_sender_pack(a.ctypes.data_as(C.c_void_p), a.size, ch, sb, w.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
            want_parts.append((dp + (k % 3), w))
            sp += pad + a.size
            dp += out_bytes + 5
        d = np.zeros(len(rows), dtype=capi.FMT_DESC)
        for i, (so, do, n, ch, bits) in enumerate(rows):
            d["kind"][i], d["src_offset"][i], d["dst_offset"][i], d["n_frames"][i], d["channels"][i], d["src_bits"][i] = capi.FMT_SENDER_PACK, so, do, n, ch, bits
        src = np.concatenate(parts)
        got = run_fmt(ctx, d, src, dp)
        want = np.full(dp, 0xA5, dtype=np.uint8)
        for off, w in want_parts:
            want[off:off + w.size] = w
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (sb, bad[:5])
        ctx.set_kernel_variant(1)                                        # the byte kernel agrees
        try:
            assert np.array_equal(run_fmt(ctx, d, src, dp), want)
        finally:
            ctx.set_kernel_variant(0)


def test_flac_pack(ctx):
    """a14: planar TInt32 -> packed BE interleaved 8/16/24; 32-bit is unsupported as in the reference (Flac.cpp:379-417)."""
    rng = np.random.default_rng(5)
    for bits, ch, n in [(8, 2, 100), (16, 2, 4096), (24, 2, 4096), (24, 6, 1152), (16, 1, 17),
                        (24, 2, 4097), (8, 2, 1), (16, 2, 3), (24, 2, 2)]:                 # stereo kernel: odd last frame
        planes = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(ch, n + 3), dtype=np.int64).astype(np.int32)
        src = planes.view(np.uint8).reshape(-1)
        d = np.zeros(1, dtype=capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["n_frames"] = capi.FMT_FLAC_PACK, ch, 32, bits, n
        d["src_plane_stride"] = (n + 3) * 4
        out_bytes = n * ch * bits // 8
        got = run_fmt(ctx, d, src, out_bytes)
        want = np.zeros(out_bytes, dtype=np.uint8)
        ptrs = (C.POINTER(C.c_int32) * ch)(*[planes[c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(ch)])
        nb = C.c_uint32(0)
        assert O.lib().ohp_flac_pack(ptrs, ch, 0, n, bits, want.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0
        assert np.array_equal(got, want), (bits, ch, n)
    d = np.zeros(1, dtype=capi.FMT_DESC)
    d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["n_frames"], d["src_plane_stride"] = capi.FMT_FLAC_PACK, 2, 32, 32, 4, 16
    with pytest.raises(capi.OhGpuError) as e:
        ctx.fmt_batch(d, 32, 32)
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_line_kernel_equals_byte_kernel_on_mixed_batches(ctx):
    """The tuned kernel (variant 0) against the byte kernel (variant 1, the one the oracle tests above pin) on batches
    that mix the three kinds, every depth, 1-10 channels, multi-chunk lengths and unaligned arena offsets."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        rows, src_parts, sp, dp = [], [], 0, 0
        for k in range(40):
            kind = [capi.FMT_UNPACK_PLANAR, capi.FMT_SENDER_PACK, capi.FMT_FLAC_PACK][int(rng.integers(0, 3))]
            ch = int(rng.integers(1, 11))
            n = int(rng.choice([1, 2, 7, 44, 240, 1000, 3000]))
            pad = int(rng.integers(0, 9))
            d = np.zeros(1, dtype=capi.FMT_DESC)
            d["kind"], d["channels"], d["n_frames"] = kind, ch, n
            if kind == capi.FMT_FLAC_PACK:
                bits = int(rng.choice([8, 16, 24]))
                stride = (n + int(rng.integers(0, 4))) * 4
                if trial % 2 == 0:
                    stride = (stride + 15) // 16 * 16                  # (other strides take the byte kernel: still must agree)
                pad = (-sp) % 4 + 4 * (pad // 4)                      # TInt32 planes are 4-byte aligned
                nbytes = ch * stride
                d["src_bits"], d["dst_bits"], d["src_plane_stride"] = 32, bits, stride
                out = n * ch * bits // 8
            else:
                sb = int(rng.integers(1, 5))
                nbytes = n * ch * sb
                d["src_bits"] = sb * 8
                if kind == capi.FMT_UNPACK_PLANAR:
                    stride = n * 4 + int(rng.integers(0, 3)) * 4
                    d["dst_plane_stride"] = stride
                    out = ch * stride
                else:
                    out = n * min(ch, 2) * min(sb, 3)
            src_parts.append(rng.integers(0, 256, size=pad + nbytes, dtype=np.uint8))
            d["src_offset"], d["dst_offset"] = sp + pad, dp + int(rng.integers(0, 4))
            sp += pad + nbytes
            dp += out + 8
            rows.append(d)
        descs = np.concatenate(rows)
        src = np.concatenate(src_parts)
        ctx.set_kernel_variant(0)
        fast = run_fmt(ctx, descs, src, dp)
        ctx.set_kernel_variant(1)
        base = run_fmt(ctx, descs, src, dp)
        ctx.set_kernel_variant(0)
        bad = np.nonzero(fast != base)[0]
        assert bad.size == 0, f"trial {trial}: {bad.size} mismatches, first at {bad[:5]}"


def test_config5_data_path_flac_planes_to_resampled_ramped_s24(ctx):
    """BASELINE config 5 behind the (host-side) FLAC decoder: planar TInt32 blocks -> a14 pack (S24 BE interleaved) ->
    resample 44.1 -> 48 kHz -> ramp -> S24 BE, device arena to device arena, against the oracle's composition."""
    import workloads as W
    rng = np.random.default_rng(21)
    ch, n_in, n_streams = 2, 5880, 3                               # 5880 frames -> 6400 output frames per stream
    planes = rng.integers(-(1 << 23), 1 << 23, size=(n_streams, ch, n_in), dtype=np.int64).astype(np.int32)
    src_planes = planes.view(np.uint8).reshape(-1)
    # stage 1 (a14): one descriptor per libFLAC block of 1152 frames, as CodecFlac::CallbackWrite sees them
    blocks = []
    for s_ in range(n_streams):
        for f0 in range(0, n_in, 1152):
            n = min(1152, n_in - f0)
            d = np.zeros(1, dtype=capi.FMT_DESC)
            d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["n_frames"] = capi.FMT_FLAC_PACK, ch, 32, 24, n
            d["src_offset"] = (s_ * ch * n_in + f0) * 4
            d["src_plane_stride"] = n_in * 4
            d["dst_offset"] = (s_ * n_in + f0) * ch * 3
            blocks.append(d)
    fmt_descs = np.concatenate(blocks)
    packed_bytes = n_streams * n_in * ch * 3
    d_planes, d_packed = ctx.upload(src_planes), ctx.malloc(packed_bytes)
    fb = ctx.fmt_batch(fmt_descs, src_planes.size, packed_bytes)
    ctx.fmt_run(fb, d_planes, d_packed)
    # stage 2: the packed arena is the resampler's source arena
    L, M, coef = capi.src_design(44100, 48000, 32, 9.0, 20000.0)
    ref = O.Src(44100, 48000, 32, 9.0, 20000.0)
    h = ctx.src_create(L, M, 32, coef)
    out_total = ref.out_frames(n_in)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 20 * O.JIFFIES_PER_MS, 40 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, n_in, L, M, 240, ch, 24, O.ENDIAN_BIG, 24, O.ENDIAN_BIG, sched)
    assert sbytes == packed_bytes
    d_out = ctx.malloc(dbytes)
    sbatch = ctx.src_batch(h, descs, packed_bytes, dbytes)
    ctx.src_run(sbatch, d_packed, d_out)
    got = ctx.download(d_out, dbytes)
    # oracle: ohp_flac_pack per block, then the resampler model on its output
    packed_ref = np.zeros(packed_bytes, dtype=np.uint8)
    for s_ in range(n_streams):
        for f0 in range(0, n_in, 1152):
            n = min(1152, n_in - f0)
            ptrs = (C.POINTER(C.c_int32) * ch)(*[planes[s_, c].ctypes.data_as(C.POINTER(C.c_int32)) for c in range(ch)])
            nb = C.c_uint32(0)
            out = packed_ref[(s_ * n_in + f0) * ch * 3:]
            assert O.lib().ohp_flac_pack(ptrs, ch, f0, n, 24, out.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0 and nb.value == n * ch * 3
    assert np.array_equal(ctx.download(d_packed, packed_bytes), packed_ref)
    want = np.zeros(dbytes, dtype=np.uint8)
    assert ref.process_batch(descs, packed_ref, want) == 0
    assert np.array_equal(got, want)
    ctx.batch_destroy(fb); ctx.batch_destroy(sbatch); ctx.src_destroy(h)
    for p in (d_planes, d_packed, d_out):
        ctx.free(p)


def test_plane_strides_that_wrap_64_bits_are_out_of_bounds(ctx):
    """A plane stride chosen so that (channels - 1) * stride wraps to something small must fail validation (OHGPU_ERR_BOUNDS),
    not reach a kernel: unpack-to-planes destination, FLAC source planes, flywheel training planes."""
    ERR_BOUNDS = -5
    for stride in (1 << 63, (1 << 64) // 2, ((1 << 64) // 3) & ~3, (1 << 64) - 4):
        d = np.zeros(1, capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["n_frames"] = capi.FMT_UNPACK_PLANAR, 3, 16, 8
        d["dst_plane_stride"] = stride
        with pytest.raises(capi.OhGpuError) as e:
            ctx.fmt_batch(d, 4096, 4096)
        assert e.value.code == ERR_BOUNDS, (stride, str(e.value))
        d = np.zeros(1, capi.FMT_DESC)
        d["kind"], d["channels"], d["src_bits"], d["dst_bits"], d["n_frames"] = capi.FMT_FLAC_PACK, 3, 32, 16, 8
        d["src_plane_stride"] = stride
        with pytest.raises(capi.OhGpuError) as e:
            ctx.fmt_batch(d, 4096, 4096)
        assert e.value.code == ERR_BOUNDS, (stride, str(e.value))
    f = np.zeros(1, capi.FLYWHEEL_DESC)
    f["channel_bytes"], f["channels"], f["in_samples"], f["out_frames"], f["block_frames"], f["sample_rate"] = (1 << 63), 2, 48, 48, 48, 48000
    with pytest.raises(capi.OhGpuError) as e:
        ctx.flywheel_batch(f, 4096, 4096)
    assert e.value.code in (ERR_BOUNDS, -1), str(e.value)
