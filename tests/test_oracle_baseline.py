"""The timed CPU baseline (ohp_src_msg_process_batch_steady: scratch allocated once, window bounds checked once per message)
computes the same bytes as the plain oracle composition it restructures (ohp_src_msg_process_batch)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import workloads as W


@pytest.mark.parametrize("rate_in,taps,ch,bits", [(44100, 32, 2, 24), (96000, 64, 8, 24), (44100, 32, 6, 16), (48000, 32, 2, 24)])
def test_steady_baseline_equals_plain_oracle(rate_in, taps, ch, bits):
    ref = O.Src(rate_in, 44100 if rate_in == 48000 else 48000, taps, 9.0, 20000.0)
    in_frames, n_streams = 3000, 3
    src = np.concatenate([W.noise_pcm(s, in_frames, ch, bits, O.ENDIAN_LITTLE) for s in range(n_streams)])
    out_total = ref.out_frames(in_frames)
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 10 * O.JIFFIES_PER_MS, 20 * O.JIFFIES_PER_MS)
    descs, sbytes, dbytes, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, ch, bits, O.ENDIAN_LITTLE, 24, O.ENDIAN_BIG, sched)
    a, b = np.zeros(dbytes, np.uint8), np.zeros(dbytes, np.uint8)
    assert ref.process_batch(descs, src, a) == 0
    d = np.ascontiguousarray(descs)
    assert O.lib().ohp_src_msg_process_batch_steady(ref.h, d.ctypes.data_as(C.c_void_p), d.size, src.ctypes.data_as(C.c_void_p),
                                                    b.ctypes.data_as(C.c_void_p)) == 0
    assert a.any() and np.array_equal(a, b)
