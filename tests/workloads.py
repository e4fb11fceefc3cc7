"""Synthetic PCM workloads shared by the tests: seeded LCG noise, sine + impulses,
message layouts and ramp schedules.  Inputs only -- nothing here computes an expected output."""
import ctypes as C

import numpy as np

import oracle_lib as O

LCG_A, LCG_C = 1664525, 1013904223
MASK = 0xFFFFFFFF


def lcg_sequence(seed, n):
    """x[0] = lcg(seed), x[i+1] = lcg(x[i]); vectorised by block doubling (uint32 wraparound)."""
    if n == 0:
        return np.zeros(0, dtype=np.uint32)
    out = np.empty(n, dtype=np.uint64)
    out[0] = (seed * LCG_A + LCG_C) & MASK
    have = 1
    a_k, c_k = LCG_A, LCG_C           # x -> a_k*x + c_k advances `have` steps
    while have < n:
        take = min(have, n - have)
        out[have:have + take] = (out[:take] * a_k + c_k) & MASK
        c_k = (c_k * a_k + c_k) & MASK
        a_k = (a_k * a_k) & MASK
        have += take
    return out.astype(np.uint32)


def stream_seed(stream_id):
    return (0x9E3779B9 * (stream_id + 1)) & MASK


def pack_subsamples(values_s32, bits, endian):
    """values_s32: int32 left-justified 32-bit subsamples -> packed bytes at `bits`, given byte order."""
    v = values_s32.astype(np.uint32)
    nb = bits // 8
    be = [((v >> (24 - 8 * b)) & 0xFF).astype(np.uint8) for b in range(nb)]
    cols = be if (endian == O.ENDIAN_BIG) else be[::-1]
    return np.stack(cols, axis=1).reshape(-1)


def noise_pcm(stream_id, n_frames, channels, bits, endian):
    """Uniform full-scale noise: subsample = top `bits` bits of the LCG word."""
    x = lcg_sequence(stream_seed(stream_id), n_frames * channels)
    return pack_subsamples(x.view(np.int32), bits, endian)


def sine_impulse_pcm(n_frames, channels, bits, endian, rate=44100, freq=997.0, dbfs=-1.0):
    """-1 dBFS 997 Hz sine with +/- full-scale impulses every 1000 frames (parity-only distribution)."""
    t = np.arange(n_frames, dtype=np.float64)
    amp = (10.0 ** (dbfs / 20.0)) * (2 ** 31 - 1)
    s = np.round(amp * np.sin(2 * np.pi * freq * t / rate)).astype(np.int64)
    s[::1000] = 2 ** 31 - 1
    s[500::1000] = -(2 ** 31)
    frames = np.repeat(s[:, None], channels, axis=1)
    frames[:, 1::2] = -frames[:, 1::2] - 1 if channels > 1 else frames[:, 1::2]
    frames = np.clip(frames, -(2 ** 31), 2 ** 31 - 1).astype(np.int64).astype(np.int32)
    return pack_subsamples(frames.reshape(-1), bits, endian)


def ramp_schedule(n_msgs, msg_jiffies, up_jiffies, down_jiffies):
    """Per-message (enabled, start, end): ramp up over the first up_jiffies, unity, ramp down over the
    last down_jiffies -- endpoints from the oracle's restatement of MsgAudio::SetRamp (Msg.cpp:1989-2046),
    driven the way Ramper::ProcessAudio (Ramper.cpp:114-134) drives it.  msg_jiffies: int or per-msg list."""
    sizes = [msg_jiffies] * n_msgs if np.isscalar(msg_jiffies) else list(msg_jiffies)
    out = [(0, O.RAMP_MAX, O.RAMP_MAX)] * n_msgs
    L = O.lib()

    def run(indices, start_value, total, direction):
        cur, remaining = start_value, total
        for i in indices:
            if remaining == 0:
                break
            m = O.MsgAudio()
            O.lib().ohp_ramp_reset(C.byref(m.ramp))
            m.size_jiffies = sizes[i]
            m.sample_rate, m.bit_depth, m.channels, m.attenuation = 48000, 24, 2, 256
            split, has = O.MsgAudio(), C.c_int(0)
            rem, end = C.c_uint32(max(remaining, sizes[i])), C.c_uint32(0)
            rc = L.ohp_msg_audio_set_ramp(C.byref(m), cur, C.byref(rem), direction, C.byref(split), C.byref(has), C.byref(end))
            assert rc == 0 and not has.value
            out[i] = (1, m.ramp.start, m.ramp.end)
            cur, remaining = end.value, rem.value

    up_idx, acc = [], 0
    for i in range(n_msgs):
        if acc >= up_jiffies:
            break
        up_idx.append(i)
        acc += sizes[i]
    down_idx, acc = [], 0
    for i in range(n_msgs - 1, -1, -1):
        if acc >= down_jiffies or i in up_idx:
            break
        down_idx.append(i)
        acc += sizes[i]
    down_idx.reverse()
    run(up_idx, O.RAMP_MIN, sum(sizes[i] for i in up_idx), O.RAMP_UP)
    run(down_idx, O.RAMP_MAX, sum(sizes[i] for i in down_idx), O.RAMP_DOWN)
    return out


def pcm_stream_descs(n_streams, frames_per_stream, frames_per_msg, channels, src_bits, src_endian, dst_bits, dst_endian,
                     schedule=None, dtype=None):
    """Contiguous streams in both arenas, fixed-size messages (last one ragged)."""
    dtype = dtype or O.MSG_DESC
    n_msgs = (frames_per_stream + frames_per_msg - 1) // frames_per_msg
    d = np.zeros(n_streams * n_msgs, dtype=dtype)
    fb_s, fb_d = channels * src_bits // 8, channels * dst_bits // 8
    j = np.arange(n_msgs)
    first = j * frames_per_msg
    count = np.minimum(frames_per_msg, frames_per_stream - first)
    for s in range(n_streams):
        sl = slice(s * n_msgs, (s + 1) * n_msgs)
        d["src_offset"][sl] = s * frames_per_stream * fb_s + first * fb_s
        d["dst_offset"][sl] = s * frames_per_stream * fb_d + first * fb_d
        d["n_frames"][sl] = count
        if schedule is not None:
            sch = np.array(schedule, dtype=np.int64)
            d["flags"][sl] = sch[:, 0]
            d["ramp_start"][sl] = sch[:, 1]
            d["ramp_end"][sl] = sch[:, 2]
        else:
            d["ramp_start"][sl] = O.RAMP_MAX
            d["ramp_end"][sl] = O.RAMP_MAX
    d["attenuation"] = 256
    d["channels"], d["src_bits"], d["src_endian"] = channels, src_bits, src_endian
    d["dst_bits"], d["dst_endian"] = dst_bits, dst_endian
    return d, n_streams * frames_per_stream * fb_s, n_streams * frames_per_stream * fb_d


def src_stream_descs(n_streams, in_frames, L, M, out_frames_per_msg, channels, src_bits, src_endian, dst_bits, dst_endian,
                     schedule=None, dtype=None):
    """Each stream: whole input resident (src_frame0 = 0), output cut into fixed-size messages."""
    dtype = dtype or O.SRC_MSG_DESC
    out_total = (in_frames * L + M - 1) // M
    n_msgs = (out_total + out_frames_per_msg - 1) // out_frames_per_msg
    d = np.zeros(n_streams * n_msgs, dtype=dtype)
    fb_s, fb_d = channels * src_bits // 8, channels * dst_bits // 8
    j = np.arange(n_msgs)
    first = j * out_frames_per_msg
    count = np.minimum(out_frames_per_msg, out_total - first)
    for s in range(n_streams):
        sl = slice(s * n_msgs, (s + 1) * n_msgs)
        d["src_offset"][sl] = s * in_frames * fb_s
        d["src_frame0"][sl] = 0
        d["src_frames"][sl] = in_frames
        d["out_frame0"][sl] = first
        d["dst_offset"][sl] = s * out_total * fb_d + first * fb_d
        d["n_frames"][sl] = count
        if schedule is not None:
            sch = np.array(schedule, dtype=np.int64)
            d["flags"][sl] = sch[:, 0]
            d["ramp_start"][sl] = sch[:, 1]
            d["ramp_end"][sl] = sch[:, 2]
        else:
            d["ramp_start"][sl] = O.RAMP_MAX
            d["ramp_end"][sl] = O.RAMP_MAX
    d["attenuation"] = 256
    d["channels"], d["src_bits"], d["src_endian"] = channels, src_bits, src_endian
    d["dst_bits"], d["dst_endian"] = dst_bits, dst_endian
    return d, n_streams * in_frames * fb_s, n_streams * out_total * fb_d, out_total, n_msgs
