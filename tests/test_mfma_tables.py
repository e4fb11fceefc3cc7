"""The matrix-pipe resampler kernel's host tables (ohpipeline_amd/csrc/src_mfma_kernel.hip, `ohgpu_src_mfma_tables`), on the CPU.

The kernel computes an output as six int8 dot products (offset digits of the samples x balanced digits of the Q28
coefficients) and recombines them in 32-bit integers.  Everything that decides WHICH bytes meet and what the constants are
lives in two host-made tables; this test replays the kernel's arithmetic from those tables in numpy -- digit planes, the A rows
taken from the padded coefficient rows at the recorded offsets, the accumulators' initial values, the recombination -- and
compares every output of whole block rows with the oracle's integer model.  No GPU: the device code is checked by the `gpu`
tests; this pins the tables and the arithmetic identity they rely on."""
import numpy as np
import pytest

import oracle_lib
from ohpipeline_amd import capi

SRC_DESC = oracle_lib.SRC_MSG_DESC


def oracle_outputs(ref, x, n_out):
    """x: int32 [frames] (one channel, 24-bit values) -> the first n_out outputs of the integer model, stream starting at frame 0."""
    raw = np.zeros((x.size, 3), dtype=np.uint8)
    u = x.astype(np.int64) & 0xFFFFFF
    raw[:, 0], raw[:, 1], raw[:, 2] = u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF
    d = np.zeros(1, dtype=SRC_DESC)
    d["src_offset"], d["src_frame0"], d["src_frames"], d["out_frame0"] = 0, 0, x.size, 0
    d["dst_offset"], d["n_frames"] = 0, n_out
    d["ramp_start"], d["ramp_end"], d["attenuation"] = 16384, 16384, 256
    d["channels"], d["src_bits"], d["src_endian"], d["dst_bits"], d["dst_endian"] = 1, 24, 1, 24, 1
    out = np.zeros(n_out * 3, dtype=np.uint8)
    assert ref.process_batch(d, raw.reshape(-1), out) == 0
    o = out.reshape(-1, 3).astype(np.int64)
    v = o[:, 0] | (o[:, 1] << 8) | (o[:, 2] << 16)
    return np.where(v >= 1 << 23, v - (1 << 24), v)


def emulate_row(dig, steps, L_blk, kb, frames):
    """frames: int64 [32 + M_blk kb + slack], frames[32 + k] = input frame k of the row (frames[0..31] its history).
    Returns the row's L_blk * kb outputs computed the kernel's way."""
    u = frames & 0xFFFFFF
    # offset digits of the two low bytes, the signed top byte
    d = [((u & 0xFF) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64),
         (((u >> 8) & 0xFF) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64),
         ((u >> 16) & 0xFF).astype(np.uint8).view(np.int8).astype(np.int64)]
    out = np.zeros(L_blk * kb, dtype=np.int64)
    for t in range(L_blk * kb // 16):
        st = steps[t]
        k0 = int(st["kc"]) * 16
        for m in range(16):
            aoff = int(st["aoff"][m])
            p, o = divmod(aoff, 96)
            a = [dig[j, p, o:o + 64].astype(np.int64) for j in range(4)]
            win = [dd[k0:k0 + 64] for dd in d]
            init = [int(st["b0"][m]), 0, int(np.int32(st["b1"][m])), 0, int(np.int32(st["b2"][m])), 0]
            assert 0 <= init[0] < 1 << 16 and abs(init[2]) < 1 << 28 and init[4] == 0     # the constant: 16 bits, the rest in ONE accumulator
            s = list(init)
            for i in range(3):
                for j in range(4):
                    s[i + j] += int(np.dot(a[j], win[i]))
            assert all(abs(v - b) < 1 << 22 for v, b in zip(s, init)), "a sum of products left the range the recombination assumes"
            t0 = (s[1] << 8) + s[0]
            uu = (s[3] << 8) + s[2] + (t0 >> 16)
            w = (s[5] << 8) + s[4] + (uu >> 16)
            assert abs(t0) < 1 << 31 and abs(uu) < 1 << 31 and abs(w << 4) < 1 << 31
            y = (w << 4) | ((uu >> 12) & 15)
            out[16 * t + m] = min(max(y, -(1 << 23)), (1 << 23) - 1)
    return out


@pytest.mark.parametrize("rates", [(44100, 48000), (48000, 44100), (32000, 48000)])
def test_tables_reproduce_the_integer_model(rates):
    ref = oracle_lib.Src(*rates)
    L, M, T = ref.L, ref.M, ref.T
    dig, steps, L_blk = capi.src_mfma_tables(L, M, T, ref.coef_q28, 8)
    assert dig.shape == (4, L, 96) and L_blk % 16 == 0 and L_blk % L == 0
    M_blk = L_blk * M // L
    assert steps.size == L_blk // 16 * 8
    # digits: balanced, and they recompose to the coefficients, oldest tap last
    c = ref.coef_q28.reshape(L, T).astype(np.int64)
    rec = sum(dig[j, :, 32:64].astype(np.int64) << (8 * j) for j in range(4))
    assert np.array_equal(rec[:, ::-1], c)
    assert not dig[:, :, :32].any() and not dig[:, :, 64:].any()
    rng = np.random.default_rng(7)
    kb = 3
    n_in = 2 * M_blk * kb + 64
    for kind in ("noise", "extremes"):
        if kind == "noise":
            x = rng.integers(-(1 << 23), 1 << 23, size=n_in, dtype=np.int64)
        else:
            x = rng.choice(np.array([-(1 << 23), (1 << 23) - 1, 0, -1, 1, 0x7FFF80, -0x7FFF80]), size=n_in)
        want = oracle_outputs(ref, x.astype(np.int32), 2 * L_blk * kb)
        # row 0 of a stream (zero history) and the row after it (its history = the stream's own frames)
        for r in range(2):
            first = r * M_blk * kb
            frames = np.zeros(32 + M_blk * kb + 80, dtype=np.int64)
            lo = first - 32
            src = x[max(lo, 0):first + M_blk * kb + 80]
            frames[max(-lo, 0):max(-lo, 0) + src.size] = src
            got = emulate_row(dig, steps, L_blk, kb, frames)
            assert np.array_equal(got, want[r * L_blk * kb:(r + 1) * L_blk * kb]), (kind, r)


def test_unsupported_geometries_are_refused():
    ref = oracle_lib.Src(96000, 48000, 64)
    with pytest.raises(capi.OhGpuError):
        capi.src_mfma_tables(ref.L, ref.M, ref.T, ref.coef_q28, 8)
    ref = oracle_lib.Src(88200, 48000)                   # 32 taps, but 15 M / L = 27 frames per tile: beyond the 64-frame window
    with pytest.raises(capi.OhGpuError):
        capi.src_mfma_tables(ref.L, ref.M, ref.T, ref.coef_q28, 8)


def emulate_halfband_row(image, bias, L_blk, frames):
    """The half-band form (csrc/src_mfma_wg_kernel.hip, HB) replayed from its one coefficient image: frames[64 + k] = input frame k
    of the row (frames[0..63] its history).  A step's K groups 0..2 are the EVEN frames' chunks s .. s + 2, group 3 the ODD frames'
    chunk s + 1; lane (g, n) of the image holds output n's sixteen coefficient digits for group g."""
    u = frames & 0xFFFFFF
    d = [((u & 0xFF) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64),
         (((u >> 8) & 0xFF) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64),
         ((u >> 16) & 0xFF).astype(np.uint8).view(np.int8).astype(np.int64)]
    even = [dd[0::2] for dd in d]
    odd = [dd[1::2] for dd in d]
    b = [bias & 0xFFFF, bias >> 16, 0]                        # as the kernel carries it: 16 bits in class 0's accumulator, the rest in class 2's
    assert abs(b[1]) < 1 << 28
    out = np.zeros(L_blk, dtype=np.int64)
    for s_ in range(L_blk // 16):
        for n in range(16):
            acc = [b[0], 0, b[1], 0, b[2], 0]
            for g in range(4):
                for i in range(3):
                    samples = even[i][16 * (s_ + g):16 * (s_ + g) + 16] if g < 3 else odd[i][16 * (s_ + 1):16 * (s_ + 1) + 16]
                    for j in range(4):
                        acc[i + j] += int(np.dot(image[j, g, n].astype(np.int64), samples))
            assert all(abs(v - i0) < 1 << 22 for v, i0 in zip(acc, [b[0], 0, b[1], 0, b[2], 0])), "a sum of products left the range the recombination assumes"
            t0 = (acc[1] << 8) + acc[0]
            uu = (acc[3] << 8) + acc[2] + (t0 >> 16)
            w = (acc[5] << 8) + acc[4] + (uu >> 16)
            assert abs(t0) < 1 << 31 and abs(uu) < 1 << 31 and abs(w << 4) < 1 << 31
            y = (w << 4) | ((uu >> 12) & 15)
            out[16 * s_ + n] = min(max(y, -(1 << 23)), (1 << 23) - 1)
    return out


def test_halfband_tables_reproduce_the_integer_model():
    ref = oracle_lib.Src(96000, 48000, 64)
    assert (ref.L, ref.M, ref.T) == (1, 2, 64)
    image, bias, L_blk = capi.src_mfma_halfband_tables(ref.coef_q28)
    assert L_blk == 128
    c = ref.coef_q28.astype(np.int64)
    assert bias == 32896 * int(c.sum()) + (1 << 27)
    # the image recomposes to the even taps along the band of groups 0..2 and to the centre tap on the diagonal of group 3
    rec = sum(image[j].astype(np.int64) << (8 * j) for j in range(4))        # [group][output][k]
    for n in range(16):
        for g in range(3):
            for i in range(16):
                m = 32 + n - 16 * g - i
                assert rec[g, n, i] == (c[2 * m] if 0 <= m <= 31 else 0)
        assert np.array_equal(rec[3, n], np.where(np.arange(16) == n, c[31], 0))
    rng = np.random.default_rng(11)
    M_blk = 2 * L_blk
    n_in = 3 * M_blk + 64
    for kind in ("noise", "extremes"):
        if kind == "noise":
            x = rng.integers(-(1 << 23), 1 << 23, size=n_in, dtype=np.int64)
        else:
            x = rng.choice(np.array([-(1 << 23), (1 << 23) - 1, 0, -1, 1, 0x7FFF80, -0x7FFF80]), size=n_in)
        want = oracle_outputs(ref, x.astype(np.int32), 3 * L_blk)
        for r in range(3):                                      # the stream's first block (zero history) and the two after it
            first = r * M_blk
            frames = np.zeros(64 + M_blk, dtype=np.int64)
            lo = first - 64
            src = x[max(lo, 0):first + M_blk]
            frames[max(-lo, 0):max(-lo, 0) + src.size] = src
            got = emulate_halfband_row(image, bias, L_blk, frames)
            assert np.array_equal(got, want[r * L_blk:(r + 1) * L_blk]), (kind, r)


def test_halfband_tables_refuse_other_filters():
    ref = oracle_lib.Src(96000, 48000, 64)
    coef = np.array(ref.coef_q28, dtype=np.int32)
    coef[5] = 12345                                             # an odd tap that is not the centre: no longer half-band
    with pytest.raises(capi.OhGpuError):
        capi.src_mfma_halfband_tables(coef)


def test_a_coefficient_whose_top_digit_is_no_int8_is_refused():
    """(127 << 24) + 0x7f7f7f is the largest coefficient with balanced digits; one above it carries into a top digit of 128, which an
    int8 cast would wrap to -128 (round 4 accepted up to (127 << 24) + 0x7fffff and made silently wrong tables).  The extremes that
    do have digits recompose; a second tap cancels the first in the bias sum, so that only the digit test can refuse."""
    L, M, T = 160, 147, 32
    base = np.zeros(L * T, dtype=np.int32)
    top = (127 << 24) + 0x7F7F7F
    for c, partner, ok in ((top, -top, True), (top + 1, -(top + 1), False), ((127 << 24) + 0x7FFFFF, -(127 << 24) - 0x7FFFFF, False),
                           (-(1 << 31), top, True)):               # (every negative int32 has digits: the bottom of the range is -(128 << 24) - 0x808080)
        coef = base.copy()
        coef[5], coef[6] = c, partner
        if not ok:
            with pytest.raises(capi.OhGpuError):
                capi.src_mfma_tables(L, M, T, coef, 8)
            continue
        dig, _, _ = capi.src_mfma_tables(L, M, T, coef, 8)
        rec = sum(dig[j, 0, 32:64].astype(np.int64) << (8 * j) for j in range(4))[::-1]
        assert rec[5] == c and rec[6] == partner
    hb = np.zeros(64, dtype=np.int32)
    hb[0], hb[2] = top + 1, -(top + 1)
    with pytest.raises(capi.OhGpuError):
        capi.src_mfma_halfband_tables(hb)
    hb[0], hb[2] = top, -top
    image, _, _ = capi.src_mfma_halfband_tables(hb)
    rec = sum(image[j].astype(np.int64) << (8 * j) for j in range(4))
    assert rec[2, 0, 0] == top          # K group 2, output 0, sample 0 meets tap 2 * (32 + 0 - 32 - 0) = 0
