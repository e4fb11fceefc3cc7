"""The resampler designs of DESIGN.md 4, checked on the CPU from the coefficient tables themselves (oracle and product make the
same table; tests/test_capi_loads.py compares them): pass band flat, everything that could alias or image back below f_pass
down by the Kaiser window's 90 dB, DC gain L, the exact-accumulation bound -- for the 44.1 -> 48 kHz polyphase filter and for
the odd-length half-band 2:1 decimator of round 3, whose zeros are asserted too.  And the integer model against the fp64 model
with unquantised coefficients: within one LSB of S24 (BASELINE.json's tolerance for the resampler; the GPU equals the integer
model bit for bit, tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import oracle_lib as O

BETA, F_PASS = 9.0, 20000.0


def prototype(ref):
    """h[p + k L] = coef[p T + k]: the polyphase table back in prototype order, unity DC gain per phase (Q28 -> 1.0)."""
    L, T = ref.L, ref.T
    h = np.zeros(L * T)
    for p in range(L):
        h[p + np.arange(T) * L] = ref.coef_q28[p * T:(p + 1) * T]
    return h / 2.0 ** 28


def response_db(h, fs, freqs):
    n = np.arange(h.size)
    H = np.array([np.sum(h * np.exp(-2j * np.pi * f / fs * n)) for f in freqs])
    return 20 * np.log10(np.maximum(np.abs(H), 1e-30))


@pytest.mark.parametrize("rate_in,taps,edge_db", [(44100, 32, -85.0), (96000, 64, -60.0), (88200, 64, -60.0)])
def test_pass_band_is_flat_and_the_stop_band_is_85_db_down(rate_in, taps, edge_db):
    """44.1 -> 48 kHz, 32 taps per phase: -85 dB from the stop edge (28 kHz) on.  The 64-tap decimators' transition band is a
    shade wider than the 8 kHz between the edges (Kaiser(9) over 63 taps: 8.8 kHz at 96 kHz), so they are -60 dB AT 28 kHz
    and -85 dB from 28.5 kHz on -- round 2's even-length 64-tap design measured -64.1 dB at the edge and -85 dB from 28.34 kHz,
    round 3's 63-tap half-band -61.5 dB and 28.40 kHz (what lies between aliases to 19.6-20 kHz)."""
    ref = O.Src(rate_in, 48000, taps, BETA, F_PASS)
    h = prototype(ref)
    fs_up = ref.L * rate_in
    f_stop = ref.f_stop
    assert f_stop == 48000 - F_PASS
    gain = 20 * np.log10(ref.L)
    passband = response_db(h, fs_up, np.linspace(0, F_PASS, 81)) - gain
    assert np.max(np.abs(passband)) < 0.01, passband                     # dB: flat to a hundredth
    assert response_db(h, fs_up, [f_stop])[0] - gain < edge_db
    stop = response_db(h, fs_up, np.linspace(f_stop + (0 if edge_db <= -85.0 else 500.0), fs_up / 2, 2000))
    assert np.max(stop) - gain < -85.0, np.max(stop) - gain
    assert abs(np.sum(h) - ref.L) < 1e-6 * ref.L                         # DC gain L (unity per phase on average)
    assert ref.sum_abs_max < (1 << 29)                                   # the lean kernel's rounding bias; < 2^30 keeps fp64 exact


def test_the_two_to_one_decimator_is_odd_length_and_half_band():
    ref = O.Src(96000, 48000, 64, BETA, F_PASS)
    c = ref.coef_q28
    assert (ref.L, ref.M, ref.T) == (1, 2, 64)
    assert c[63] == 0                                                    # 63 taps, stored as 64
    assert np.array_equal(c[:63], c[62::-1])                             # symmetric about tap 31: whole-sample delay
    odd = [k for k in range(1, 64, 2) if c[k] != 0]
    assert odd == [31]                                                   # every second coefficient EXACTLY zero but the centre
    assert abs(int(c[31]) - (1 << 27)) < 1024                            # the centre tap is one half (DC gain 1, Q28)
    assert np.count_nonzero(c) == 33


@pytest.mark.parametrize("rate_in,taps", [(44100, 32), (96000, 64)])
def test_integer_model_is_within_one_lsb_of_the_fp64_model(rate_in, taps):
    """Full-scale noise and a full-scale step: the Q28 rounding of the coefficients and the final rounding to S24 together stay
    inside +-1 LSB of the double-precision filter with unquantised coefficients."""
    ref = O.Src(rate_in, 48000, taps, BETA, F_PASS)
    frames, ch = 6000, 2
    rng = np.random.default_rng(5)
    x = rng.integers(-(1 << 23), 1 << 23, size=(frames, ch), dtype=np.int64)
    x[3000:3400] = (1 << 23) - 1
    x[3400:3800] = -(1 << 23)
    src = np.zeros((frames * ch, 3), dtype=np.uint8)
    v = x.reshape(-1) & 0xffffff
    src[:, 0], src[:, 1], src[:, 2] = (v >> 16) & 0xff, (v >> 8) & 0xff, v & 0xff
    src = src.reshape(-1)
    n_out = ref.out_frames(frames)
    d = np.zeros(1, dtype=O.SRC_MSG_DESC)
    d["src_frames"], d["n_frames"], d["channels"], d["src_bits"], d["dst_bits"] = frames, n_out, ch, 24, 24
    d["src_endian"], d["dst_endian"], d["attenuation"] = O.ENDIAN_BIG, O.ENDIAN_BIG, 256
    out = np.zeros(n_out * ch * 3, dtype=np.uint8)
    assert ref.process_batch(d, src, out) == 0
    o = out.reshape(-1, 3).astype(np.int64)
    got = (o[:, 0] << 16) | (o[:, 1] << 8) | o[:, 2]
    got = np.where(got >= 1 << 23, got - (1 << 24), got)
    err, y = ref.process_f64(d, src)
    assert err == 0
    want = np.clip(y, -(1 << 23), (1 << 23) - 1)
    assert np.max(np.abs(got - want)) <= 1.0, np.max(np.abs(got - want))
