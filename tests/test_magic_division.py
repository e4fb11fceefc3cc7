"""The exact-division multipliers the line kernels use instead of an integer divide (csrc/ohgpu_internal.h magic_u31):
x / d == umulhi(x, m) >> s for every x < 2^31.  The formula is restated here and checked against Python's integers on
the divisors the kernels meet (channels 1..10, frames - 1 up to 131070) and on adversarial numerators."""
import numpy as np


def magic_u31(d):
    if d <= 1:
        return 0, 0
    l = (d - 1).bit_length()                   # 2^(l-1) < d <= 2^l
    return ((1 << (31 + l)) // d) + 1, l - 1


def divide(x, m, s):
    return x if m == 0 else ((x * m) >> 32) >> s


def test_multiplier_fits_32_bits_and_is_exact():
    rng = np.random.default_rng(0)
    divisors = list(range(1, 2049)) + [9215, 9216, 65535, 65536, 131069, 131070, (1 << 31) - 1] + \
        [int(v) for v in rng.integers(2, 1 << 31, size=2000)]
    for d in divisors:
        m, s = magic_u31(d)
        assert m < (1 << 32) and s < 32
        xs = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d,
              ((1 << 31) - 1) // d * d - 1} | {int(v) for v in rng.integers(0, 1 << 31, size=64)}
        for x in xs:
            if 0 <= x < (1 << 31):
                assert divide(x, m, s) == x // d, (x, d)
