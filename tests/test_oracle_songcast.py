"""The Songcast sender oracle (oracle/ohp_songcast.c).

The reference holds no known-answer test for OhmMsgAudio's wire format, so the frame layout is "parity unpinned" (see
oracle/ohp_songcast.h).  What is checked here: the writer against a datagram assembled by hand from the reference's field
table (OhmMsg.cpp:391-400, 225-241; Ohm.cpp:44-52), the writer against the restated READER (OhmMsg.cpp:100-174) field by
field, and the packetiser / driver arithmetic against the properties Sender.cpp:277-321 and OhmSender.cpp:418-480 imply."""
import struct

import numpy as np
import pytest

import oracle_lib as O


def hand_built_frame(flags, samples, frame, net_ts, latency, sample_start, samples_total, rate, bit_rate, vol, depth, channels,
                     codec, audio):
    stream = struct.pack(">QIIhBBBB", samples_total, rate, bit_rate, vol, depth, channels, 0, len(codec)) + codec
    per_frame = struct.pack(">BBHIIIIQ", 50, flags, samples, frame, net_ts, latency, 0, sample_start)
    total = 8 + len(per_frame) + len(stream) + len(audio)
    return b"Ohm " + struct.pack(">BBH", 1, 3, total) + per_frame + stream + bytes(audio)


def test_frame_equals_the_hand_assembled_datagram():
    audio = bytes(range(1, 1 + 36))
    n, sh = O.ohm_stream_header(123456789012, 48000, 2304000, -3, 24, 2, b"FLAC")
    assert n == 26
    n, got = O.ohm_audio_frame(O_FLAG_LOSSLESS | O_FLAG_HALT, 6, 0x01020304, 0xa0b0c0d0, 0x00112233, 0x0102030405060708, sh, audio)
    want = hand_built_frame(0x03, 6, 0x01020304, 0xa0b0c0d0, 0x00112233, 0x0102030405060708, 123456789012, 48000, 2304000, -3,
                            24, 2, b"FLAC", audio)
    assert n == len(want) == 8 + 28 + 26 + 36
    assert bytes(got) == want
    # the first 12 bytes spelled out: "Ohm ", major 1, type 3 (audio), total length, header length 50, flags, samples
    assert bytes(got[:12]) == b"Ohm \x01\x03" + bytes([0, 98]) + b"\x32\x03\x00\x06"


O_FLAG_HALT, O_FLAG_LOSSLESS, O_FLAG_TIMESTAMPED, O_FLAG_RESENT, O_FLAG_TIMESTAMPED2 = 1, 2, 4, 8, 16


def test_timestamped_sets_both_timestamp_flags():
    _, sh = O.ohm_stream_header(0, 44100, 0, 0, 16, 2)
    _, f = O.ohm_audio_frame(O_FLAG_TIMESTAMPED, 1, 0, 99, 0, 0, sh, b"\1\2\3\4")
    assert f[9] == O_FLAG_TIMESTAMPED | O_FLAG_TIMESTAMPED2          # iTimestamped2 = iTimestamped, OhmMsg.cpp:211
    err, a = O.ohm_audio_parse(f)
    assert err == 0 and a.timestamped == 1 and a.timestamped2 == 1 and a.network_timestamp == 99


def test_writer_and_reader_agree_on_every_field():
    rng = np.random.default_rng(11)
    for _ in range(200):
        codec = bytes(rng.integers(32, 127, rng.integers(0, 30), dtype=np.uint8))
        depth, ch = int(rng.choice([8, 16, 24])), int(rng.integers(1, 3))
        samples = int(rng.integers(0, 5760 // (ch * depth // 8) + 1))
        audio = bytes(rng.integers(0, 256, samples * ch * depth // 8, dtype=np.uint8))
        total, rate, br = int(rng.integers(0, 2**63)), int(rng.choice([44100, 48000, 96000, 192000])), int(rng.integers(0, 2**32))
        vol = int(rng.integers(-32768, 32768))
        n, sh = O.ohm_stream_header(total, rate, br, vol, depth, ch, codec)
        assert n == 22 + len(codec)
        flags = int(rng.integers(0, 16))
        frame, ts, lat, start = (int(rng.integers(0, 2**32)) for _ in range(3)), None, None, None
        frame, ts, lat = frame
        start = int(rng.integers(0, 2**63))
        n, f = O.ohm_audio_frame(flags, samples, frame, ts, lat, start, sh, audio)
        assert n == 8 + 28 + 22 + len(codec) + len(audio)
        err, a = O.ohm_audio_parse(f)
        assert err == 0
        assert (a.halt, a.lossless, a.timestamped, a.resent) == (flags & 1, (flags >> 1) & 1, (flags >> 2) & 1, (flags >> 3) & 1)
        assert (a.samples, a.frame, a.network_timestamp, a.media_latency, a.media_timestamp) == (samples, frame, ts, lat, 0)
        assert (a.sample_start, a.samples_total, a.sample_rate, a.bit_rate, a.volume_offset) == (start, total, rate, br, vol)
        assert (a.bit_depth, a.channels, bytes(a.codec)[:a.codec_bytes]) == (depth, ch, codec)
        assert a.msg_type == 3 and a.msg_bytes == n - 8
        assert bytes(f[a.audio_offset:a.audio_offset + a.audio_bytes]) == audio


def test_limits_assert_like_the_reference():
    assert O.ohm_stream_header(0, 44100, 0, 0, 16, 2, b"x" * 30)[0] == O.ERR_ASSERT          # Bws<kMaxCodecBytes>
    _, sh = O.ohm_stream_header(0, 44100, 0, 0, 16, 2)
    assert O.ohm_audio_frame(0, 1441, 0, 0, 0, 0, sh, bytes(5761))[0] == O.ERR_ASSERT          # kMaxSampleBytes
    good = O.ohm_audio_frame(0, 1, 0, 0, 0, 0, sh, bytes(4))[1]
    bad = good.copy(); bad[0] = ord("o")
    assert O.ohm_audio_parse(bad)[0] == O.ERR_ASSERT                                              # THROW(OhmError), Ohm.cpp:27-29
    bad = good.copy(); bad[4] = 2
    assert O.ohm_audio_parse(bad)[0] == O.ERR_ASSERT                                              # major version, :31-33
    bad = good.copy(); bad[8] = 49
    assert O.ohm_audio_parse(bad)[0] == O.ERR_ASSERT                                              # ASSERT(headerBytes == kHeaderBytes)


def pcm_msg(frames, rate, bits, ch, offset_frames=0):
    m = O.MsgAudio()
    assert O.lib().ohp_msg_audio_init_pcm(m, frames * ch * bits // 8, ch, rate, bits) == 0
    return m


@pytest.mark.parametrize("rate", [44100, 48000, 96000, 192000])
def test_packetiser_cuts_five_millisecond_packets(rate):
    rng = np.random.default_rng(rate)
    jps = O.JIFFIES_PER_SEC // rate
    sizes = [int(v) for v in rng.integers(1, 2 * rate // 100, 60)]                    # up to 20 ms per message
    msgs = [pcm_msg(n, rate, 24, 2) for n in sizes]
    err, frags, packs = O.sender_packetise(msgs, flush=True)
    assert err == 0
    packet = 5 * O.JIFFIES_PER_MS
    consumed = [0] * len(msgs)
    for k, pk in enumerate(packs):
        fs = frags[pk.first_fragment:pk.first_fragment + pk.n_fragments]
        jiffies = sum(f.playable.jiffies for f in fs)
        if k < len(packs) - 1:
            assert jiffies == packet                                                     # Sender.cpp:288-301
        else:
            assert jiffies < packet                                                      # the MsgQuit flush sends what is left
        for f in fs:
            # fragments of one message tile its audio without gaps (MsgAudioPcm::CreatePlayable rounds the offset down, Msg.cpp:2236-2240)
            assert f.playable.offset_bytes == consumed[f.msg]
            consumed[f.msg] += f.playable.size_bytes
        samples = sum(f.playable.size_bytes for f in fs) // 6
        if k < len(packs) - 1:
            assert abs(samples * jps - packet) < 2 * jps                                 # 220 / 221 samples at 44.1 kHz
    assert consumed == [n * 6 for n in sizes]                                            # nothing lost, nothing sent twice
    # without the flush the tail stays pending
    err, frags2, packs2 = O.sender_packetise(msgs, flush=False)
    assert err == 0 and len(packs2) == len(packs) - 1


def test_packetiser_splits_ramps_with_the_messages():
    m = pcm_msg(4800, 48000, 16, 2)                                                      # 100 ms, one long down ramp
    m.ramp = O.Ramp(O.RAMP_MAX, 0, O.RAMP_DOWN, 1)
    err, frags, packs = O.sender_packetise([m], flush=True)
    assert err == 0 and len(packs) == 21 and packs[-1].n_fragments == 0                 # 20 full packets + the empty flush
    ends = [f.playable.ramp.end for f in frags]
    starts = [f.playable.ramp.start for f in frags]
    assert starts[0] == O.RAMP_MAX and ends[-1] == 0
    assert all(starts[i + 1] == ends[i] for i in range(len(frags) - 1))                  # Ramp::Split keeps the ramp continuous
    assert all(f.playable.ramp.enabled for f in frags)


def test_driver_counts_frames_and_samples():
    d = O.OhmDriver()
    L = O.lib()
    L.ohp_ohm_driver_init(d, 100)
    L.ohp_ohm_driver_set_track_position(d, 1 << 20, 5000)
    codec = np.frombuffer(b"PCM", dtype=np.uint8).copy()
    assert L.ohp_ohm_driver_set_audio_format(d, 48000, 2304000, 2, 24, 1, O._ptr(codec), 3, 5000) == 0
    assert d.latency_ohm == 100 * 48000 * 256 // 1000                                    # UpdateLatencyOhm, OhmSender.cpp:320-323
    msgs = [pcm_msg(240 * 3 + 17, 48000, 24, 2)]
    audio = [np.random.default_rng(5).integers(0, 256, (240 * 3 + 17) * 6, dtype=np.uint8)]
    grams = O.songcast_datagrams(d, msgs, audio, flush=True)
    assert len(grams) == 4
    start = 5000
    for k, g in enumerate(grams):
        err, a = O.ohm_audio_parse(g)
        assert err == 0
        assert (a.frame, a.sample_start, a.samples_total, a.media_latency, a.lossless, a.halt) == (k, start, 1 << 20, d.latency_ohm, 1, 0)
        assert a.samples == (240 if k < 3 else 17) and a.audio_bytes == a.samples * 6
        assert (a.bit_depth, a.channels, bytes(a.codec)[:3]) == (24, 2, b"PCM")
        lo = start - 5000
        assert bytes(g[a.audio_offset:]) == bytes(audio[0][lo * 6:(lo + a.samples) * 6])    # 24-bit stereo passes through unchanged
        start += a.samples
    assert (d.frame, d.sample_start) == (4, 5000 + 737)
    # an empty SendPendingAudio (ProcessMsg(MsgDecodedStream*) with nothing pending) sends nothing and counts nothing
    assert O.songcast_datagrams(d, [], [], flush=True) == [] and d.frame == 4
    L.ohp_ohm_driver_stream_interrupted(d)
    assert d.frame == 254
