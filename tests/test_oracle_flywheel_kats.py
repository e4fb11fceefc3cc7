"""The reference's own known-answer tests for FlywheelRamper, restated against the CPU oracle
(OpenHome/Media/Tests/TestFlywheelRamper.cpp; line numbers in each test).  They pin oracle/ohp_flywheel.c."""
import ctypes as C

import numpy as np

import oracle_lib as O


def feedback(degree, descale, coeff_fmt, data_fmt, out_fmt, coeffs, samples, n):
    c = np.array(coeffs, dtype=np.uint32).astype(np.int32)
    s = np.array(samples, dtype=np.uint32).astype(np.int32)
    m = O.FeedbackModel()
    O.lib().ohp_feedback_init(C.byref(m), degree, descale, coeff_fmt, data_fmt, out_fmt, c.ctypes.data, s.ctypes.data)
    return [O.lib().ohp_feedback_next_sample(C.byref(m)) & 0xffffffff for _ in range(n)]


def test_feedback_model_algorithm():                    # Test1, :111-157
    got = feedback(4, 8, 1, 1, 1, [0x01000000, 0x02000000, 0x04000000, 0x08000000],
                   [0x01000000, 0x02000000, 0x04000000, 0x08000000], 4)
    assert got == [0x00aa0000, 0x00555400, 0x002b5200, 0x0016fa00]


def test_feedback_model_scaling():                      # Test2, :160-271
    cases = {(1, 1, 1): (0x20000, 0x400), (2, 1, 1): (0x40000, 0x1000), (3, 1, 1): (0x80000, 0x4000),
             (4, 1, 1): (0x100000, 0x10000), (1, 2, 1): (0x40000, 0x800), (1, 3, 1): (0x80000, 0x1000),
             (1, 4, 1): (0x100000, 0x2000), (1, 1, 2): (0x10000, 0x200), (1, 1, 3): (0x8000, 0x100),
             (1, 1, 4): (0x4000, 0x80), (2, 2, 2): (0x40000, 0x1000)}
    for (cf, df, of), want in cases.items():
        assert tuple(feedback(2, 8, cf, df, of, [0x01000000, 0], [0x01000000, 0], 2)) == want, (cf, df, of)


def test_feedback_model_step_response():                # Test3, :274-320
    assert feedback(6, 8, 2, 2, 2, [0x40000000, 0, 0, 0, 0, 0], [0x40000000, 0, 0, 0, 0, 0], 10) == [0x40000000] * 10


def test_feedback_model_periodic_impulse():             # Test4, :323-400
    one = 0x40000000
    assert feedback(6, 8, 2, 2, 2, [0, one, 0, 0, 0, 0], [one, 0, 0, 0, 0, 0], 8) == [0, one] * 4
    assert feedback(6, 8, 2, 2, 2, [0, 0, one, 0, 0, 0], [one, 0, 0, 0, 0, 0], 6) == [0, 0, one] * 2


def test_feedback_model_oscillator():                   # Test5, :403-520
    one, neg = 0x40000000, 0xc0000000
    assert feedback(6, 8, 2, 2, 2, [neg, 0, 0, 0, 0, 0], [one, 0, 0, 0, 0, 0], 6) == [neg, one] * 3
    assert feedback(6, 8, 2, 2, 2, [0, neg, 0, 0, 0, 0], [one, 0, 0, 0, 0, 0], 6) == [0, neg, 0, one, 0, neg]
    assert feedback(6, 8, 2, 2, 2, [0, 0, neg, 0, 0, 0], [one, 0, 0, 0, 0, 0], 12) == [0, 0, neg, 0, 0, one] * 2


BURG_IN_1 = [111411200, 110362624, 108855296, 107937792, 108265472, 108462080, 108199936, 108527616, 107479040,
             105578496, 102170624, 97845248, 93257728, 88342528, 83034112, 77004800, 70844416, 63963136, 56885248,
             51183616, 46399488, 41418752, 36306944, 31457280, 27000832, 21430272, 15597568, 10027008, 4521984, 196608,
             -5439488, -10420224, -15335424, -20905984, -26083328, -32112640, -37552128, -42270720, -47251456,
             -52232192, -55836672, -59834368, -63700992, -67960832]                                   # :535-539
BURG_IN_2 = [80150528, 78249984, 75628544, 74055680, 73924608, 73924608, 73400320, 72744960, 72351744, 70189056,
             67174400, 64225280, 60948480, 57999360, 53673984, 49676288, 46596096, 42598400, 38731776, 36044800,
             34144256, 31588352, 28966912, 26673152, 24838144, 21889024, 18087936, 14548992, 9961472, 7208960, 3735552,
             131072, -3342336, -7602176, -10616832, -14417920, -18546688, -21626880, -25296896, -28901376, -32505856,
             -35913728, -38731776, -42401792]                                                         # :542-545


def burg(samples32, degree=3):
    x = (np.array(samples32, dtype=np.int64) >> 16).astype(np.int16)        # *(samples+i) = (TInt16)(input>>16), :579-582
    out = np.zeros(degree, dtype=np.int16)
    h = np.zeros(degree, dtype=np.int16)
    per = np.zeros(x.size, dtype=np.int16)
    pef = np.zeros(x.size, dtype=np.int16)
    O.lib().ohp_burgs_method(x.ctypes.data, x.size, degree, out.ctypes.data, h.ctypes.data, per.ctypes.data, pef.ctypes.data)
    return out.tolist()


def test_burgs_method_known_answers():                  # Test6, :549-612
    assert burg(BURG_IN_1) == [-16619, 8835, -374]
    assert burg(BURG_IN_2) == [-14748, 5235, 1360]


def test_decimation_and_coefficient_overflow():         # FlywheelRamper.cpp:316-331, 342-372
    L = O.lib()
    assert [L.ohp_flywheel_decimation_factor(r) for r in (44100, 48000, 88200, 96000, 176400, 192000, 352800, 384000)] == \
        [1, 1, 2, 2, 4, 4, 1, 1]
    def overflow(c):
        a = np.array(c, dtype=np.int16)
        return L.ohp_flywheel_coeff_overflow(a.ctypes.data, a.size, 3)
    assert overflow([-16619, 8835, -374]) == 0                            # -8158 is inside [-1.0, 1.0] in 3.13
    assert overflow([-16619, 300, 0]) == -16619 + 300 + 8192              # below -1.0: the (negative) excess
    assert overflow([100, 200, 300]) == 0
    assert overflow([8000, 300, 0]) == 108
