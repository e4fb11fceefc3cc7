"""Pins the CPU oracle against the known answers the reference's OWN test suites hold for this path.

Each test restates a block of OpenHome/Media/Tests/TestMsg.cpp (file:line in the docstring) with the
oracle standing where the reference class stood, so the assertions read like the reference's TEST()s.
These run on the CPU (-m "not gpu").
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
kMax, kMin = O.RAMP_MAX, O.RAMP_MIN
EUp, EDown, ENone, EMute = O.RAMP_UP, O.RAMP_DOWN, O.RAMP_NONE, O.RAMP_MUTE
kPerMs = O.JIFFIES_PER_MS
SAMPLE_RATES = [7350, 8000, 11025, 12000, 14700, 16000, 22050, 24000, 29400, 32000, 44100, 48000, 88200, 96000,
                176400, 192000]


# ----------------------------------------------------------------------------- helpers
def ramp_set(ramp, start, frag, remaining, direction):
    split = O.Ramp()
    pos = C.c_uint32(0)
    rc = O.lib().ohp_ramp_set(C.byref(ramp), start, frag, remaining, direction, C.byref(split), C.byref(pos))
    return rc, split, pos.value


def new_ramp():
    r = O.Ramp()
    O.lib().ohp_ramp_reset(C.byref(r))
    return r


def msg_pcm(data_bytes, channels, rate, bits):
    m = O.MsgAudio()
    rc = O.lib().ohp_msg_audio_init_pcm(C.byref(m), data_bytes, channels, rate, bits)
    return rc, m


def msg_silence(jiffies, rate, bits, channels):
    m = O.MsgAudio()
    j = C.c_uint32(jiffies)
    rc = O.lib().ohp_msg_audio_init_silence(C.byref(m), C.byref(j), rate, bits, channels)
    assert rc == 0
    return m, j.value


def set_ramp(m, start, remaining, direction):
    split = O.MsgAudio()
    has = C.c_int(0)
    rem = C.c_uint32(remaining)
    end = C.c_uint32(0)
    rc = O.lib().ohp_msg_audio_set_ramp(C.byref(m), start, C.byref(rem), direction, C.byref(split), C.byref(has), C.byref(end))
    return rc, end.value, rem.value, (split if has.value else None)


def playable_of(m):
    p = O.Playable()
    assert O.lib().ohp_create_playable(C.byref(m), C.byref(p)) == 0
    return p


def ingest(data, bits, endian):
    err, be = O.construct_pcm(data, bits, endian)
    assert err == 0
    cell = np.zeros(O.MAX_BYTES, dtype=np.uint8)   # a DecodedAudio cell, Msg.h:134
    cell[:be.size] = be
    return cell


def read(p, cell):
    err, out, frags = O.playable_read(p, cell)
    assert err == 0
    return out, frags


# ----------------------------------------------------------------------------- RampArray.h
def test_ramp_table_matches_reference_data():
    """RampArray.h:7-74 -- all 512 Q15 multipliers, fixture extracted by tests/golden/make_ramp_table_fixture.py."""
    golden = json.load(open(os.path.join(GOLDEN, "ramp_table_q15.json")))["values"]
    table = O.ramp_table()
    assert len(golden) == 512
    assert table.tolist() == golden
    assert table[0] == 0x7FFF and table[256] == 5793 and (table[-6:] == 0).all()


@pytest.mark.skipif(not os.path.exists("/root/reference/OpenHome/Media/Pipeline/RampArray.h"),
                    reason="reference tree only exists in the build container")
def test_ramp_table_fixture_is_current():
    import re
    text = open("/root/reference/OpenHome/Media/Pipeline/RampArray.h").read()
    body = text[text.index("kRampArray[]"):text.index("};")]
    values = [int(v, 16) for v in re.findall(r"0x([0-9A-Fa-f]{4})", body)]
    assert values == json.load(open(os.path.join(GOLDEN, "ramp_table_q15.json")))["values"]


# ----------------------------------------------------------------------------- SuiteRamp: Ramp::Set
def test_suite_ramp_set_endpoints():
    """TestMsg.cpp:1391-1443 -- Ramp::Set endpoint KATs."""
    jiffies = kPerMs
    ramp = new_ramp()
    rc, split, pos = ramp_set(ramp, kMax, jiffies, jiffies, EDown)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (kMax, kMin, EDown)

    ramp = new_ramp()   # start=kMax, up: asserts (TEST_THROWS AssertionFailed, :1409)
    rc, _, _ = ramp_set(ramp, kMax, jiffies, jiffies, EUp)
    assert rc == O.ERR_ASSERT

    ramp = new_ramp()
    rc, _, _ = ramp_set(ramp, kMin, jiffies, jiffies, EUp)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (kMin, kMax, EUp)

    ramp = new_ramp()
    rc, _, _ = ramp_set(ramp, kMax, jiffies, 2 * jiffies, EDown)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (kMax, (kMax - kMin) // 2, EDown)

    ramp = new_ramp()
    rc, _, _ = ramp_set(ramp, kMin, jiffies, 2 * jiffies, EUp)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (kMin, (kMax - kMin) // 2, EUp)

    ramp = new_ramp()
    start = (kMax - kMin) // 2
    rc, _, _ = ramp_set(ramp, start, jiffies, 2 * jiffies, EDown)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (start, (kMax - kMin) // 4, EDown)

    ramp = new_ramp()
    rc, _, _ = ramp_set(ramp, start, jiffies, 2 * jiffies, EUp)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (start, kMax - ((kMax - kMin) // 4), EUp)


def _apply(ramp, data, bits, channels):
    err, out = O.ramp_apply(data, bits, channels, ramp.start, ramp.end)
    assert err == 0
    return out


def test_suite_ramp_applicator_properties():
    """TestMsg.cpp:1445-1591 -- RampApplicator at 8/16/24/32 bit: monotone, L==R, endpoints within 2."""
    kAudioDataSize = 792
    audio = bytes([0x7f]) * kAudioDataSize
    table = O.ramp_table()

    ramp = new_ramp()
    assert ramp_set(ramp, kMax, kAudioDataSize, kAudioDataSize, EDown)[0] == 0
    out = _apply(ramp, audio, 8, 2).reshape(-1, 2)
    assert out[0, 0] >= 0x7d
    assert (out[:, 0] == out[:, 1]).all()
    assert (np.diff(out[:, 0].astype(int)) <= 0).all()
    assert out[-1, 0] == 0

    signed = bytes([0xff]) * kAudioDataSize                      # :1473-1492 negative subsamples
    out = _apply(ramp, signed, 8, 2).reshape(-1, 2)
    assert out[0, 0] >= 0xfd
    assert (((out[:, 0] & 0x80) != 0) | (out[:, 0] == 0)).all()
    assert (out[:, 0] == out[:, 1]).all()
    assert (np.diff(out[:, 0].astype(int)) <= 0).all()
    assert out[-1, 0] == 0

    out = _apply(ramp, audio, 16, 2).reshape(-1, 2, 2)           # :1494-1505
    v = (out[:, :, 0].astype(int) << 8) | out[:, :, 1]
    assert (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[0, 0] <= 0x7f7f

    out = _apply(ramp, audio, 24, 2).reshape(-1, 2, 3)           # :1507-1518
    v = (out[:, :, 0].astype(int) << 16) | (out[:, :, 1].astype(int) << 8) | out[:, :, 2]
    assert (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[0, 0] <= 0x7f7f7f

    out = _apply(ramp, audio, 32, 2).reshape(-1, 2, 4)           # :1520-1531
    v = (out[:, :, 0].astype(np.int64) << 24) | (out[:, :, 1].astype(int) << 16) | (out[:, :, 2].astype(int) << 8) | out[:, :, 3]
    assert (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) <= 0).all() and v[0, 0] <= 0x7f7f7f7f

    ramp = new_ramp()                                            # :1533-1548 [Min..Max]
    assert ramp_set(ramp, kMin, kAudioDataSize, kAudioDataSize, EUp)[0] == 0
    out = _apply(ramp, audio, 8, 2).reshape(-1, 2)
    assert out[0, 0] <= 0x02 and (out[:, 0] == out[:, 1]).all()
    assert (np.diff(out[:, 0].astype(int)) >= 0).all() and out[-1, 0] >= 0x7d

    ramp = new_ramp()                                            # :1550-1563 [Max..50%]
    assert ramp_set(ramp, kMax, kAudioDataSize, kAudioDataSize * 2, EDown)[0] == 0
    out = _apply(ramp, audio, 8, 2).reshape(-1, 2)
    assert out[0, 0] >= 0x7d
    end_guess = (0x7f * int(table[256])) >> 15
    assert 0 <= end_guess - int(out[-1, 0]) <= 0x02

    ramp = new_ramp()                                            # :1565-1578 [Min..50%]
    assert ramp_set(ramp, kMin, kAudioDataSize, kAudioDataSize * 2, EUp)[0] == 0
    out = _apply(ramp, audio, 8, 2).reshape(-1, 2)
    assert out[0, 0] <= 0x02
    assert 0 <= end_guess - int(out[-1, 0]) <= 0x02

    ramp = new_ramp()                                            # :1580-1591 [50%..25%]
    assert ramp_set(ramp, kMax // 2, kAudioDataSize, kAudioDataSize * 2, EDown)[0] == 0
    out = _apply(ramp, audio, 8, 2).reshape(-1, 2)
    start_guess = (0x7f * int(table[256])) >> 15
    assert 0 <= start_guess - int(out[0, 0]) < 0x02
    end_guess = (0x7f * int(table[384])) >> 15
    assert 0 <= end_guess - int(out[-1, 0]) <= 0x02


def test_suite_ramp_intersections():
    """TestMsg.cpp:1593-1625 -- opposite-direction ramps split at their intersection; same-direction take lower."""
    jiffies = kPerMs
    ramp = new_ramp()
    assert ramp_set(ramp, kMax // 2, jiffies, jiffies, EDown)[0] == 0
    rc, split, pos = ramp_set(ramp, kMin, jiffies, 2 * jiffies, EUp)
    assert rc == 1
    assert (ramp.start, ramp.end, ramp.direction) == (0, kMax // 4, EUp)
    assert (split.start, split.end, split.direction) == (ramp.end, 0, EDown)
    assert ramp.enabled and split.enabled

    ramp = new_ramp()
    assert ramp_set(ramp, kMax // 2, jiffies, 4 * jiffies, EDown)[0] == 0
    before = (ramp.start, ramp.end, ramp.direction)
    rc, _, _ = ramp_set(ramp, (10 * kMax) // 7, jiffies, (5 * jiffies) // 2, EDown)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == before

    ramp = new_ramp()
    assert ramp_set(ramp, kMax // 2, jiffies, 2 * jiffies, EDown)[0] == 0
    start = (2 * kMax) // 5
    rc, _, _ = ramp_set(ramp, start, jiffies, jiffies, EDown)
    assert rc == 0 and (ramp.start, ramp.end, ramp.direction) == (start, 0, EDown)


def test_suite_ramp_silence_and_pcm_msgs():
    """TestMsg.cpp:1627-1712 -- ramps through MsgSilence / MsgAudioPcm, split msg shapes, 17+23 ms ramp."""
    jiffies = kPerMs
    silence, jiffies = msg_silence(jiffies, 44100, 8, 2)
    rc, end, rem, split = set_ramp(silence, kMax, jiffies, EDown)
    assert rc == 0 and end == kMin and split is None
    out, _ = read(playable_of(silence), None)
    assert out.size > 0 and (out == 0).all()

    kEncodedAudioSize = 768
    enc = bytes([0x7f]) * kEncodedAudioSize
    rc, pcm = msg_pcm(kEncodedAudioSize, 2, 44100, 16)
    assert rc == 0
    cell = ingest(enc, 16, O.ENDIAN_LITTLE)
    jiffies = pcm.size_jiffies
    rc, end, rem, split = set_ramp(pcm, kMax // 2, jiffies, EDown)
    assert end == kMin
    rc, end, rem, remaining = set_ramp(pcm, kMin, jiffies * 2, EUp)
    assert end != kMin
    assert remaining is not None and remaining.ramp.enabled and remaining.ramp.end == kMin
    assert pcm.size_jiffies == jiffies // 2 == remaining.size_jiffies
    out, _ = read(playable_of(pcm), cell)
    v = (out.reshape(-1, 2, 2)[:, :, 0].astype(int) << 8) | out.reshape(-1, 2, 2)[:, :, 1]
    assert v[0, 0] == 0 and (v[:, 0] == v[:, 1]).all() and (np.diff(v[:, 0]) >= 0).all()
    prev = v[-1, 0]
    out, _ = read(playable_of(remaining), cell)
    v = (out.reshape(-1, 2, 2)[:, :, 0].astype(int) << 8) | out.reshape(-1, 2, 2)[:, :, 1]
    assert v[-1, 0] == 0 and (v[:, 0] == v[:, 1]).all()
    assert (np.diff(np.concatenate([[prev], v[:, 0]]))[1:] <= 0).all()

    s1, size1 = msg_silence(kPerMs * 17, 44100, 16, 2)           # :1697-1712
    s2, size2 = msg_silence(kPerMs * 23, 44100, 16, 2)
    remaining_duration = s1.size_jiffies + s2.size_jiffies
    rc, cur, remaining_duration, _ = set_ramp(s1, kMax, remaining_duration, EDown)
    rc, cur, remaining_duration, _ = set_ramp(s2, cur, remaining_duration, EDown)
    assert cur == kMin


def test_suite_ramp_muted():
    """TestMsg.cpp:1715-1746 -- muted ramp is [Min..Min] and yields silence whichever order SetMuted/SetRamp come in."""
    ramp = new_ramp()
    O.lib().ohp_ramp_set_muted(C.byref(ramp))
    assert (ramp.direction, ramp.start, ramp.end) == (EMute, kMin, kMin)
    enc = bytes([0x7f]) * 768
    for order in ("mute_first", "ramp_first"):
        rc, pcm = msg_pcm(768, 1, 44100, 8)
        cell = ingest(enc, 8, O.ENDIAN_LITTLE)
        if order == "mute_first":
            O.lib().ohp_ramp_set_muted(C.byref(pcm.ramp))
            set_ramp(pcm, kMax, kPerMs * 20, EDown)
        else:
            set_ramp(pcm, kMax, kPerMs * 20, EDown)
            O.lib().ohp_ramp_set_muted(C.byref(pcm.ramp))
        out, _ = read(playable_of(pcm), cell)
        assert out.size == 768 and (out == 0).all()


# ----------------------------------------------------------------------------- SuiteMsgAudio
def test_suite_msg_audio_jiffies():
    """TestMsg.cpp:780-812 -- lower rates report more jiffies; 8/16/24-bit sizes are 1 : 1/2 : 1/3."""
    prev = 0xffffffff
    for rate in SAMPLE_RATES:
        rc, m = msg_pcm(1200, 2, rate, 8)
        assert rc == 0 and prev > m.size_jiffies
        prev = m.size_jiffies
    j = [msg_pcm(1200, 2, 44100, b)[1].size_jiffies for b in (8, 16, 24)]
    assert j[0] == 2 * j[1] == 3 * j[2]
    assert O.lib().ohp_jiffies_per_sample(44101) == O.ERR_SAMPLE_RATE          # THROW(SampleRateInvalid), Msg.cpp:472
    assert O.lib().ohp_jiffies_per_sample(44100) == 1280 and O.lib().ohp_jiffies_per_sample(48000) == 1176
    assert O.lib().ohp_jiffies_per_sample(96000) == 588


def test_suite_msg_audio_split():
    """TestMsg.cpp:814-836 -- Split lengths add up; Split(0), Split(size), Split(size+1) assert."""
    rc, m = msg_pcm(1200, 2, 44100, 8)
    jiffies = m.size_jiffies
    rem = O.MsgAudio()
    assert O.lib().ohp_msg_audio_split(C.byref(m), 800, C.byref(rem)) == 0
    assert 0 < m.size_jiffies < jiffies and 0 < rem.size_jiffies < jiffies
    assert m.size_jiffies + rem.size_jiffies == jiffies
    for bad in (0, m.size_jiffies, m.size_jiffies + 1):
        assert O.lib().ohp_msg_audio_split(C.byref(m), bad, C.byref(rem)) == O.ERR_ASSERT
    assert msg_pcm(0, 2, 44100, 8)[0] == O.ERR_ASSERT                          # zero-length msg asserts (:926)


def test_suite_msg_audio_attenuation_kat():
    """TestMsg.cpp:982-996 -- RAOP attenuation: 0x7f7f at unity/4 reads back as 0x7f7f / 4."""
    b = 0x7f
    cell = ingest(bytes([b, b, b, b]), 16, O.ENDIAN_LITTLE)
    rc, pcm = msg_pcm(4, 2, 44100, 16)
    pcm.attenuation = O.UNITY_ATTENUATION // 4
    out, _ = read(playable_of(pcm), cell)
    subsample = int(np.int16((int(out[0]) << 8) + int(out[1])))
    assert subsample == ((b << 8) + b) // 4


def test_attenuation_requires_16_bit():
    """Msg.cpp:2741 -- ASSERT(iBitDepth == 16)."""
    err, _ = O.apply_attenuation(bytes(6), 24, 128)
    assert err == O.ERR_ASSERT
    err, out = O.apply_attenuation(bytes([1, 2, 3, 4, 5, 6]), 24, 256)
    assert err == 0 and out.tolist() == [1, 2, 3, 4, 5, 6]


# ----------------------------------------------------------------------------- SuiteMsgPlayable
def _descending(n=256):
    return bytes((0xff - i) & 0xff for i in range(n))


def test_suite_msg_playable_bytes_and_passthrough():
    """TestMsg.cpp:1106-1147 -- same Bytes() at every rate; byte-exact pass-through of 0xff,0xfe,..."""
    data = _descending()
    sizes = set()
    for rate in SAMPLE_RATES:
        rc, m = msg_pcm(len(data), 2, rate, 8)
        sizes.add(playable_of(m).size_bytes)
    assert sizes == {len(data)}
    rc, m = msg_pcm(len(data), 2, 44100, 8)
    out, frags = read(playable_of(m), ingest(data, 8, O.ENDIAN_LITTLE))
    assert out.tobytes() == data and frags == [len(data)]


@pytest.mark.parametrize("split_at", ["quarter", "quarter_minus_1"])
def test_suite_msg_playable_split_msg_then_read(split_at):
    """TestMsg.cpp:1149-1172, 1199-1222 -- split MsgAudioPcm (also at a non-sample boundary), contents stay contiguous."""
    data = _descending()
    cell = ingest(data, 8, O.ENDIAN_LITTLE)
    rc, m = msg_pcm(len(data), 2, 44100, 8)
    pos = m.size_jiffies // 4 - (1 if split_at == "quarter_minus_1" else 0)
    rem = O.MsgAudio()
    assert O.lib().ohp_msg_audio_split(C.byref(m), pos, C.byref(rem)) == 0
    p, rp = playable_of(m), playable_of(rem)
    if split_at == "quarter":
        assert rp.size_bytes == 3 * p.size_bytes
    a, _ = read(p, cell)
    b, _ = read(rp, cell)
    assert a.tobytes() + b.tobytes() == data


def test_suite_msg_playable_split_playable():
    """TestMsg.cpp:1174-1197, 1239-1251 -- MsgPlayable::Split; Split(Bytes()) -> nullptr; Split(0), Split(Bytes()+1) assert."""
    data = _descending()
    cell = ingest(data, 8, O.ENDIAN_LITTLE)
    rc, m = msg_pcm(len(data), 2, 44100, 8)
    p = playable_of(m)
    rem, has = O.Playable(), C.c_int(0)
    assert O.lib().ohp_playable_split(C.byref(p), p.size_bytes // 4, C.byref(rem), C.byref(has)) == 0
    assert has.value == 1 and rem.size_bytes == 3 * p.size_bytes
    a, _ = read(p, cell)
    b, _ = read(rem, cell)
    assert a.tobytes() + b.tobytes() == data
    p = playable_of(m)
    assert O.lib().ohp_playable_split(C.byref(p), p.size_bytes, C.byref(rem), C.byref(has)) == 0 and has.value == 0
    assert O.lib().ohp_playable_split(C.byref(p), 0, C.byref(rem), C.byref(has)) == O.ERR_ASSERT
    assert O.lib().ohp_playable_split(C.byref(p), p.size_bytes + 1, C.byref(rem), C.byref(has)) == O.ERR_ASSERT


def test_suite_msg_playable_split_at_one_jiffy():
    """TestMsg.cpp:1224-1237 -- first part has 0 bytes, the remainder carries everything."""
    data = _descending()
    cell = ingest(data, 8, O.ENDIAN_LITTLE)
    rc, m = msg_pcm(len(data), 2, 44100, 8)
    rem = O.MsgAudio()
    assert O.lib().ohp_msg_audio_split(C.byref(m), 1, C.byref(rem)) == 0
    a, fa = read(playable_of(m), cell)
    b, _ = read(playable_of(rem), cell)
    assert a.size == 0 and fa == [] and b.tobytes() == data


def test_suite_msg_playable_silence():
    """TestMsg.cpp:1253-1313 -- silence sizes grow with rate; contents are zeros; splits keep total length."""
    prev = 0
    for rate in SAMPLE_RATES:
        s, _ = msg_silence(kPerMs * 5, rate, 8, 2)
        b = playable_of(s).size_bytes
        assert prev < b
        prev = b
    s, size = msg_silence(kPerMs, 44100, 8, 1)
    p = playable_of(s)
    total = p.size_bytes
    out, _ = read(p, None)
    assert out.size == total and (out == 0).all()
    rem, has = O.Playable(), C.c_int(0)
    assert O.lib().ohp_playable_split(C.byref(p), p.size_bytes // 4, C.byref(rem), C.byref(has)) == 0
    assert 3 * p.size_bytes == rem.size_bytes and p.size_bytes + rem.size_bytes == total
    p = playable_of(s)
    assert O.lib().ohp_playable_split(C.byref(p), p.size_bytes // 4 - 1, C.byref(rem), C.byref(has)) == 0
    assert p.size_bytes + rem.size_bytes == total
    s10, size10 = msg_silence(kPerMs, 192000, 32, 10)                          # :1292-1297
    p10 = playable_of(s10)
    assert p10.size_bytes == (size10 // O.lib().ohp_jiffies_per_sample(192000)) * 40
    out, _ = read(p10, None)
    assert (out == 0).all()
    s, _ = msg_silence(kPerMs, 44100, 8, 1)                                    # :1299-1307
    rem_s = O.MsgAudio()
    assert O.lib().ohp_msg_audio_split(C.byref(s), 1, C.byref(rem_s)) == 0
    assert playable_of(s).size_bytes == 0 and playable_of(rem_s).size_bytes == total


# ----------------------------------------------------------------------------- survey-recorded vector
def test_survey_recorded_ramp_vector():
    """the compiled reference turned S24 85 39 1c into 85 39 00 under an enabled unity ramp."""
    v = json.load(open(os.path.join(GOLDEN, "survey_vectors.json")))["ramp_s24_unity"]
    frame = bytes.fromhex(v["first_subsample_in"]) * v["channels"]
    data = frame * v["n_frames"]
    err, out = O.ramp_apply(data, v["bit_depth"], v["channels"], v["ramp_start"], v["ramp_end"])
    assert err == 0 and out[:3].tobytes().hex() == v["first_subsample_out"]
