"""GPU parity for the Songcast sender frames: ohgpu_ohm_* against the oracle's restatement of
Sender::ProcessAudio / SendPendingAudio / DoProcessFragment (Av/Songcast/Sender.cpp:277-377), MsgPlayable::Read and
OhmSenderDriver::SendAudio + OhmMsgAudio::Serialise (OhmSender.cpp:418-480, OhmMsg.cpp:363-413).  Bit-exact."""
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


def make_messages(rng, rate, bits, ch, n_msgs, ramps=True, silence=True, attenuate=False):
    """A stream's messages the way they reach the Sender: PCM of assorted sizes, some ramped, some MsgSilence."""
    msgs, audio = [], []
    for _ in range(n_msgs):
        frames = int(rng.integers(1, rate // 50))                        # up to 20 ms
        kind = rng.integers(0, 10)
        m = O.MsgAudio()
        if silence and kind == 0:
            j = C.c_uint32(frames * (O.JIFFIES_PER_SEC // rate))
            assert O.lib().ohp_msg_audio_init_silence(m, C.byref(j), rate, bits, ch) == 0
            audio.append(None)
        else:
            assert O.lib().ohp_msg_audio_init_pcm(m, frames * ch * bits // 8, ch, rate, bits) == 0
            audio.append(rng.integers(0, 256, frames * ch * bits // 8, dtype=np.uint8))
            if ramps and kind in (1, 2, 3):
                a, b = sorted(int(v) for v in rng.integers(0, O.RAMP_MAX + 1, 2))
                if a == b:                                                 # a ramp that goes nowhere is not a valid Ramp (Msg.cpp:745-782)
                    a, b = (a - 1, b) if a > 0 else (a, b + 1)
                m.ramp = O.Ramp(b, a, O.RAMP_DOWN, 1) if kind == 1 else O.Ramp(a, b, O.RAMP_UP, 1)
            elif ramps and kind == 4:
                m.ramp = O.Ramp(0, 0, O.RAMP_MUTE, 1)                      # MsgAudioPcm::CreatePlayable -> MsgPlayableSilence
            if attenuate and bits == 16 and kind in (5, 6):
                m.attenuation = int(rng.integers(1, 256))
        msgs.append(m)
    return msgs, audio


class Workload:
    def __init__(self):
        self.streams, self.frames, self.fragments = [], [], []
        self.src = []
        self.src_bytes = 0
        self.expected = []          # (dst_offset, datagram)
        self.dst_bytes = 0

    def add_stream(self, rng, rate, bits, ch, n_msgs, codec=b"PCM", latency_ms=100, sample_start=0, samples_total=0,
                   halt_last=False, gap=0, **kw):
        msgs, audio = make_messages(rng, rate, bits, ch, n_msgs, **kw)
        wire_ch, wire_bits = min(ch, 2), min(bits, 24)
        # ---- the oracle: Sender + OhmSenderDriver ----
        d = O.OhmDriver()
        L = O.lib()
        L.ohp_ohm_driver_init(d, latency_ms)
        L.ohp_ohm_driver_set_track_position(d, samples_total, sample_start)
        cbuf = np.frombuffer(codec or b"\0", dtype=np.uint8).copy()
        assert L.ohp_ohm_driver_set_audio_format(d, rate, rate * bits * ch, wire_ch, wire_bits, 1, O._ptr(cbuf), len(codec), sample_start) == 0
        latency_ohm = d.latency_ohm
        grams = O.songcast_datagrams(d, msgs, audio, flush=True, halt_last=halt_last)
        # ---- the same packets as device descriptors ----
        s = np.zeros(1, dtype=capi.OHM_STREAM)
        s["samples_total"], s["sample_rate"], s["bit_rate"] = samples_total, rate, rate * bits * ch
        s["src_channels"], s["src_bits"], s["codec_bytes"] = ch, bits, len(codec)
        s["codec"][0, :len(codec)] = np.frombuffer(codec, dtype=np.uint8)
        stream_index = len(self.streams)
        self.streams.append(s)
        base = []
        for a in audio:
            base.append(self.src_bytes)
            if a is not None:
                pad = int(np.random.default_rng(self.src_bytes).integers(0, 5))     # arbitrary source alignment
                self.src.append(np.concatenate([a, np.zeros(pad, dtype=np.uint8)]))
                self.src_bytes += a.size + pad
        err, frags, packs = O.sender_packetise(msgs, flush=True)
        assert err == 0
        frame_no, start, k_gram = 0, sample_start, 0
        for k, pk in enumerate(packs):
            fs = frags[pk.first_fragment:pk.first_fragment + pk.n_fragments]
            samples = sum(f.playable.size_bytes for f in fs) // (ch * bits // 8)
            halt = halt_last and k == len(packs) - 1
            if samples == 0 and not halt:                                  # OhmSender.cpp:434-438: nothing to send
                continue
            fr = np.zeros(1, dtype=capi.OHM_FRAME_DESC)
            fr["dst_offset"], fr["sample_start"], fr["stream"], fr["frame"] = self.dst_bytes, start, stream_index, frame_no
            fr["media_latency"], fr["first_fragment"], fr["n_fragments"] = latency_ohm, len(self.fragments), len(fs)
            fr["flags"] = capi.OHM_FLAG_LOSSLESS | (capi.OHM_FLAG_HALT if halt else 0)
            self.frames.append(fr)
            for f in fs:
                p = f.playable
                g = np.zeros(1, dtype=capi.OHM_FRAGMENT)
                g["n_frames"] = p.size_bytes // (ch * bits // 8)
                g["attenuation"] = p.attenuation
                if p.is_silence:
                    g["flags"] = O.FLAG_SILENCE
                else:
                    g["src_offset"] = base[f.msg] + p.offset_bytes
                    if p.ramp.enabled:
                        g["flags"], g["ramp_start"], g["ramp_end"] = O.FLAG_RAMP, p.ramp.start, p.ramp.end
                self.fragments.append(g)
            gram = grams[k_gram]
            k_gram += 1
            self.expected.append((self.dst_bytes, gram))
            self.dst_bytes += gram.size + gap
            start += samples
            frame_no += 1
        assert k_gram == len(grams)

    def run(self, ctx):
        src = np.concatenate(self.src) if self.src else np.zeros(1, dtype=np.uint8)
        streams = np.concatenate(self.streams)
        frames = np.concatenate(self.frames) if self.frames else np.zeros(0, dtype=capi.OHM_FRAME_DESC)
        fragments = np.concatenate(self.fragments) if self.fragments else np.zeros(0, dtype=capi.OHM_FRAGMENT)
        dst_bytes = self.dst_bytes + 64
        d_src, d_dst = ctx.upload(src), ctx.malloc(dst_bytes)
        ctx.memset(d_dst, 0xA5, dst_bytes)
        b = ctx.ohm_batch(streams, frames, fragments, src.size, dst_bytes)
        ctx.ohm_run(b, d_src, d_dst)
        out = ctx.download(d_dst, dst_bytes)
        ctx.batch_destroy(b)
        ctx.free(d_src)
        ctx.free(d_dst)
        want = np.full(dst_bytes, 0xA5, dtype=np.uint8)
        for off, gram in self.expected:
            want[off:off + gram.size] = gram
        return out, want


def check(out, want, expected):
    if np.array_equal(out, want):
        return
    for k, (off, gram) in enumerate(expected):
        got = out[off:off + gram.size]
        if not np.array_equal(got, gram):
            bad = int(np.flatnonzero(got != gram)[0])
            raise AssertionError(f"frame {k} at {off}: first difference at byte {bad} of {gram.size}: got {got[bad]:#x}, want {gram[bad]:#x}")
    raise AssertionError("bytes outside every frame were modified")


@pytest.mark.parametrize("rate,bits,ch", [(48000, 24, 2), (44100, 16, 2), (96000, 32, 2), (44100, 24, 1), (192000, 24, 2), (48000, 8, 2)])
def test_stereo_and_mono_streams(ctx, rate, bits, ch):
    w = Workload()
    w.add_stream(np.random.default_rng(rate + bits + ch), rate, bits, ch, 40, codec=b"FLAC", sample_start=12345, samples_total=1 << 33,
                 attenuate=True)
    out, want = w.run(ctx)
    check(out, want, w.expected)


@pytest.mark.parametrize("rate,bits,ch", [(48000, 24, 6), (44100, 16, 6), (48000, 32, 8), (44100, 24, 4), (48000, 16, 3),
                                          (96000, 24, 6), (192000, 24, 6), (192000, 16, 4), (176400, 8, 3)])
def test_wider_streams_select_the_first_two_channels(ctx, rate, bits, ch):
    """Channel select after the ramp / silence the playable carries; 16-bit 6-channel silence shows the channel-id bytes of
    MsgPlayableSilence::ReadBlock (Msg.cpp:2877) in the second wire channel.  The high rates give fragments of several
    hundred frames (ohm_wide_kernel's second and later rounds of 256), the stream's last fragment ends with the arena (its
    last frames are read byte by byte)."""
    w = Workload()
    w.add_stream(np.random.default_rng(ch * 1000 + bits), rate, bits, ch, 40, codec=b"WAV", attenuate=True)
    out, want = w.run(ctx)
    check(out, want, w.expected)


def test_ten_channel_stream_sends_channels_eight_and_nine(ctx):
    w = Workload()
    w.add_stream(np.random.default_rng(10), 48000, 24, 10, 30, ramps=False, silence=False)    # Sender::FirstChannelToSend
    out, want = w.run(ctx)
    check(out, want, w.expected)


def test_many_streams_in_one_batch_with_gaps_and_a_halt(ctx):
    rng = np.random.default_rng(77)
    w = Workload()
    formats = [(48000, 24, 2), (44100, 16, 2), (96000, 24, 2), (48000, 32, 2), (44100, 24, 1), (48000, 24, 6), (192000, 24, 2), (88200, 16, 2)]
    for k in range(24):
        rate, bits, ch = formats[k % len(formats)]
        codec = bytes(rng.integers(65, 91, int(rng.integers(0, 30)), dtype=np.uint8))
        w.add_stream(rng, rate, bits, ch, int(rng.integers(1, 25)), codec=codec, latency_ms=int(rng.integers(0, 400)),
                     sample_start=int(rng.integers(0, 1 << 40)), samples_total=int(rng.integers(0, 1 << 40)),
                     halt_last=(k % 5 == 0), gap=int(rng.integers(0, 7)), attenuate=True)
    out, want = w.run(ctx)
    check(out, want, w.expected)


def test_headers_with_the_generic_kernel(ctx):
    """A mono / stereo frame's header rides with its first audio through the line kernel; when the generic kernel runs the
    audio instead (kernel variant 1), ohm_header_kernel has to write every header."""
    rng = np.random.default_rng(78)
    w = Workload()
    for k, (rate, bits, ch) in enumerate([(48000, 24, 2), (44100, 16, 2), (44100, 24, 1), (48000, 24, 6), (48000, 8, 2)]):
        w.add_stream(rng, rate, bits, ch, 12, codec=b"PCM" * (k % 3), halt_last=(k == 1), gap=k, attenuate=True)
    ctx.set_kernel_variant(1)
    try:
        out, want = w.run(ctx)
    finally:
        ctx.set_kernel_variant(0)
    check(out, want, w.expected)
    out, want = w.run(ctx)
    check(out, want, w.expected)


def test_uniform_batch_of_plain_stereo_frames(ctx):
    """The shape the throughput figure is quoted on: every frame 5 ms of plain stereo S24 at 48 kHz."""
    rng = np.random.default_rng(5)
    w = Workload()
    for _ in range(16):
        w.add_stream(rng, 48000, 24, 2, 20, ramps=False, silence=False)
    out, want = w.run(ctx)
    check(out, want, w.expected)


def test_validation(ctx):
    s = np.zeros(1, dtype=capi.OHM_STREAM)
    s["sample_rate"], s["src_channels"], s["src_bits"] = 48000, 2, 24
    fr = np.zeros(1, dtype=capi.OHM_FRAME_DESC)
    fr["n_fragments"] = 1
    fg = np.zeros(1, dtype=capi.OHM_FRAGMENT)
    fg["n_frames"], fg["attenuation"] = 961, 256                        # 961 * 6 = 5766 > kMaxSampleBytes
    with pytest.raises(capi.OhGpuError) as e:
        ctx.ohm_batch(s, fr, fg, 1 << 20, 1 << 20)
    assert e.value.code == capi.ERR_INVALID and "5760" in str(e.value)
    fg["n_frames"] = 240
    with pytest.raises(capi.OhGpuError) as e:
        ctx.ohm_batch(s, fr, fg, 1 << 20, 100)                          # the frame does not fit the destination arena
    assert e.value.code == capi.ERR_BOUNDS
    with pytest.raises(capi.OhGpuError) as e:
        ctx.ohm_batch(s, fr, fg, 100, 1 << 20)                          # the fragment reads beyond the source arena
    assert e.value.code == capi.ERR_BOUNDS
    s2 = s.copy(); s2["codec_bytes"] = 30
    with pytest.raises(capi.OhGpuError) as e:
        ctx.ohm_batch(s2, fr, fg, 1 << 20, 1 << 20)
    assert e.value.code == capi.ERR_INVALID
    s3 = s.copy(); s3["src_channels"] = 10
    fg3 = fg.copy(); fg3["flags"] = O.FLAG_RAMP
    with pytest.raises(capi.OhGpuError) as e:
        ctx.ohm_batch(s3, fr, fg3, 1 << 20, 1 << 20)
    assert e.value.code == capi.ERR_UNSUPPORTED
    hb, fb = C.c_uint32(0), C.c_uint32(0)
    assert capi.lib().ohgpu_ohm_frame_layout(s.ctypes.data_as(C.c_void_p), 240, C.byref(hb), C.byref(fb)) == 0
    assert (hb.value, fb.value) == (58, 58 + 1440)
