// host_design.cpp -- host-side tables the device kernels consume: the RampArray multipliers and the
// polyphase resampler's Q28 coefficients.  Product code (never calls into oracle/).
#include <cmath>
#include <cstdint>
#include <vector>

#include "ohgpu_internal.h"

namespace ohgpu {

// RampArray.h:7-74 describes its 512 Q15 entries as "a ramp down curve over 0 to -60dB".  Every entry
// equals min(0x7FFF, round(32768 * (1 - i/512)^2.5)); with n = 512 - i that is round(sqrt(n^5 / 2^15)),
// which is evaluated here in exact integer arithmetic so that no libm rounding can move an entry.
// tests/test_capi_loads.py compares all 512 with the values extracted from the reference header.
void build_ramp_table(uint16_t out[512])
{
    for (uint32_t i = 0; i < 512; i++) {
        const uint64_t n = 512u - i;
        const uint64_t n5 = n * n * n * n * n;
        // v = round-half-up(sqrt(n5 / 2^15)): the largest v with (2v - 1)^2 * 2^13 <= n5
        uint64_t lo = 0, hi = 32768;
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) / 2;
            const uint64_t t = 2 * mid - 1;
            if (t * t * 8192u <= n5) lo = mid; else hi = mid - 1;
        }
        out[i] = (uint16_t)(lo > 32767u ? 32767u : lo);
    }
}

static uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    while (b) { const uint32_t t = a % b; a = b; b = t; }
    return a;
}

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x * 0.25;
    for (int k = 1; k < 500; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-20) break;
    }
    return sum;
}

// Resampler specification (DESIGN.md "Resampler"; the reference has no sample-rate converter):
//   L/M = rate_out/rate_in reduced, N = L*T prototype taps, Kaiser(beta)-windowed sinc,
//   stop edge f_stop = rate_out - f_pass, cutoff midway, DC gain L, Q28 rounding half up,
//   polyphase order coef[p*T + k] = h[p + k*L].
//   An integer decimator (L = 1) gets an ODD length, N = T - 1, centred on a tap (a type I linear-phase filter, whole-sample
//   delay), stored with coef[T - 1] = 0.  For 2:1 (96 -> 48 kHz: cutoff midway between 20 and 28 kHz = exactly a quarter of the
//   input rate) that prototype is a HALF-BAND filter: sinc(d / 2) vanishes at every even distance d from the centre, so every
//   second coefficient is exactly zero after the Q28 rounding and an output needs T / 2 products plus the centre tap
//   (src_lean_kernel's half-band instantiations; any other kernel just multiplies by the zeros).
int design_src(uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass,
               std::vector<int32_t>* coef_q28, uint32_t* L_out, uint32_t* M_out)
{
    if (rate_in == 0 || rate_out == 0 || T == 0) return set_error(OHGPU_ERR_INVALID, "src design: zero rate or taps");
    const uint32_t g = gcd_u32(rate_in, rate_out);
    const uint32_t L = rate_out / g, M = rate_in / g;
    if ((uint64_t)L * T > (1u << 22)) return set_error(OHGPU_ERR_INVALID, "src design: L*T too large (%u*%u)", L, T);
    *L_out = L;
    *M_out = M;
    if (coef_q28 == nullptr) return OHGPU_OK;
    const uint32_t N = (L == 1 && T > 1) ? T - 1 : L * T;
    double f_stop = (double)rate_out - f_pass;
    if (f_stop > (double)rate_in - f_pass && rate_out > 2 * rate_in) f_stop = (double)rate_in - f_pass;
    const double fs_up = (double)L * (double)rate_in;
    const double fc = 0.5 * (f_pass + f_stop);
    const double wc = 2.0 * fc / fs_up;
    const double centre = 0.5 * (double)(N - 1);
    const double i0b = bessel_i0(beta);
    std::vector<double> h(N);
    double sum = 0.0;
    for (uint32_t n = 0; n < N; n++) {
        const double d = (double)n - centre;
        const double x = wc * d;
        const double sinc = (std::fabs(x) < 1e-12) ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        const double r = (centre > 0.0) ? d / centre : 0.0;
        const double arg = 1.0 - r * r;
        const double w = bessel_i0(beta * std::sqrt(arg > 0.0 ? arg : 0.0)) / i0b;
        h[n] = wc * sinc * w;
        sum += h[n];
    }
    const double scale = (double)L / sum;
    coef_q28->assign((size_t)L * T, 0);
    int64_t worst = 0;
    for (uint32_t p = 0; p < L; p++) {
        int64_t sabs = 0;
        for (uint32_t k = 0; k < T; k++) {
            const double v = (p + k * L < N) ? h[p + k * L] * scale : 0.0;
            const int32_t q = (int32_t)std::floor(v * 268435456.0 + 0.5);
            (*coef_q28)[p * T + k] = q;
            sabs += q < 0 ? -(int64_t)q : (int64_t)q;
        }
        if (sabs > worst) worst = sabs;
    }
    // sum|c| < 2^30 and |x| <= 2^23 keep every partial sum an integer below 2^53: fp64 accumulation is then exact
    if (worst >= ((int64_t)1 << 30))
        return set_error(OHGPU_ERR_INVALID, "src design: sum|c| = %lld breaks the exact-accumulation bound", (long long)worst);
    return OHGPU_OK;
}

}  // namespace ohgpu
