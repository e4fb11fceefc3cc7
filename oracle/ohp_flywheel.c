/* ohp_flywheel.c -- see ohp_flywheel.h (TEST INFRASTRUCTURE ONLY). */
#include "ohp_flywheel.h"

#include <stdlib.h>
#include <string.h>

static const uint32_t kBurgDataDescaleBitCount = 1;     /* FlywheelRamper.cpp:11 */
static const uint32_t kBurgOutputFormat = 3;            /* :12 */
#define BURG_SCALE_SHIFT (16 - 3)                        /* kBurgScaleShift, :13 */
static const uint32_t kFeedbackDataDescaleBitCount = 0; /* :16 */
static const uint32_t kFeedbackDataFormat = 1;          /* :17 */

static int16_t wrap16(int32_t v) { return (int16_t)(uint16_t)(uint32_t)v; }
static int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t shl32(int32_t a, uint32_t s) { return (int32_t)((uint32_t)a << s); }

void ohp_burgs_method(const int16_t* samples, uint32_t count, uint32_t degree, int16_t* out, int16_t* h,
                      int16_t* per, int16_t* pef)
{
    uint32_t limit1 = count - 1, limit2 = limit1;                       /* :248-249 */
    for (uint32_t n = 0; n < degree; n++) {
        int32_t sn = 0, sd = 0;
        for (uint32_t j = 0; j < limit1; j++) {                         /* :256-265 */
            const int16_t t1 = wrap16((int32_t)samples[j + n + 1] + pef[j]);
            const int16_t t2 = wrap16((int32_t)samples[j] + per[j]);
            const int32_t t1t1 = mul32(t1, t1), t2t2 = mul32(t2, t2), t1t2 = mul32(t1, t2);
            sn = sub32(sn, mul32(2, t1t2));
            sd = add32(sd, add32(t1t1, t2t2));
        }
        limit1--;
        int16_t t3 = 0;
        if (sn != 0) {                                                  /* :271-275; C division truncates */
            const int64_t ratio = (int64_t)((uint64_t)(int64_t)sn << BURG_SCALE_SHIFT) / (int64_t)sd;
            t3 = (int16_t)(uint16_t)(uint64_t)ratio;
        }
        out[n] = t3;
        if (n > 0) {                                                    /* :279-294 */
            for (uint32_t j = 0; j < n; j++) {
                const int32_t prod = mul32(t3, out[n - j - 1]);
                h[j] = wrap16(prod >> BURG_SCALE_SHIFT);
                h[j] = wrap16((int32_t)h[j] + out[j]);
            }
            for (uint32_t j = 0; j < n; j++) out[j] = h[j];
            limit2--;
        }
        if (n == degree - 1) break;                                     /* :296-299 */
        for (uint32_t j = 0; j < limit2; j++) {                         /* :301-312 */
            const uint32_t i = j + 1;
            int32_t p = (int32_t)pef[j] + samples[i + n];
            p = mul32(p, t3);
            per[j] = wrap16((int32_t)per[j] + wrap16(p >> BURG_SCALE_SHIFT));
            int32_t f = (int32_t)per[i] + samples[i];
            f = mul32(f, t3);
            pef[j] = wrap16(f >> BURG_SCALE_SHIFT);
            pef[j] = wrap16((int32_t)pef[j] + pef[i]);
        }
    }
}

uint32_t ohp_flywheel_decimation_factor(uint32_t sample_rate)
{
    switch (sample_rate) {
    case 192000: case 176400: return 4;
    case 88200: case 96000: return 2;
    default: return 1;
    }
}

int16_t ohp_flywheel_coeff_overflow(const int16_t* coeffs, uint32_t count, uint32_t format)
{
    const int16_t one = (int16_t)(1 << (16 - format));
    int16_t total = 0;
    for (uint32_t j = 0; j < count; j++) total = wrap16((int32_t)total + coeffs[j]);
    if (total <= one && total >= -one) return 0;
    return (total & 0x8000) ? wrap16((int32_t)total + one) : wrap16((int32_t)total - one);
}

void ohp_feedback_init(ohp_feedback_model* m, uint32_t state_count, uint32_t data_descale_bits, uint32_t coeff_format,
                       uint32_t data_format, uint32_t output_format, int32_t* coeffs, int32_t* samples)
{
    m->coeffs = coeffs;
    m->samples = samples;
    m->state_count = state_count;
    m->data_descale_bits = data_descale_bits;
    m->coeff_format = coeff_format;
    m->scale_shift_for_output = (int32_t)(data_format + data_descale_bits) - (int32_t)output_format;   /* :432 */
    for (uint32_t j = 0; j < state_count; j++) samples[j] >>= data_descale_bits;                          /* :442-445 */
}

int32_t ohp_feedback_next_sample(ohp_feedback_model* m)
{
    int32_t sum = 0;
    for (uint32_t j = 0; j < m->state_count; j++) {                     /* :457-466 */
        const int64_t product = (int64_t)m->samples[j] * (int64_t)m->coeffs[j];
        sum = add32(sum, (int32_t)(product >> 32));
    }
    for (uint32_t j = m->state_count - 1; j > 0; j--) m->samples[j] = m->samples[j - 1];   /* :470-473 */
    sum = shl32(sum, m->coeff_format);                                  /* :476-477 */
    m->samples[0] = sum;
    if (m->scale_shift_for_output < 0) sum >>= (uint32_t)(-m->scale_shift_for_output);    /* :479-486 */
    else sum = shl32(sum, (uint32_t)m->scale_shift_for_output);
    return sum;
}

int ohp_flywheel_ramp(const uint8_t* training, uint64_t channel_bytes, uint32_t in_samples, uint32_t sample_rate,
                      uint32_t channels, uint32_t out_frames, uint32_t block_frames, uint8_t* out)
{
    const uint32_t degree = OHP_FLYWHEEL_DEGREE;
    if (channels < 1 || channels > OHP_FLYWHEEL_MAX_CHANNELS || block_frames == 0) return -1;
    if ((uint64_t)in_samples * 4 > channel_bytes) return -1;            /* ASSERT(aSamples.Bytes() >= expectedBytes), :180 */
    const uint32_t dec = ohp_flywheel_decimation_factor(sample_rate);
    const uint32_t count = in_samples / dec;                            /* aSamples.Bytes() / (kBytesPerSample * decFactor), :196 */
    if (count < degree + 1) return -1;
    int16_t* in16 = (int16_t*)calloc(count, sizeof(int16_t));
    int16_t* per = (int16_t*)calloc(count, sizeof(int16_t));
    int16_t* pef = (int16_t*)calloc(count, sizeof(int16_t));
    int32_t (*fb_samples)[OHP_FLYWHEEL_DEGREE] = calloc(channels, sizeof(*fb_samples));
    int32_t (*fb_coeffs)[OHP_FLYWHEEL_DEGREE] = calloc(channels, sizeof(*fb_coeffs));
    ohp_feedback_model* models = (ohp_feedback_model*)calloc(channels, sizeof(ohp_feedback_model));
    if (!in16 || !per || !pef || !fb_samples || !fb_coeffs || !models) {
        free(in16); free(per); free(pef); free(fb_samples); free(fb_coeffs); free(models);
        return -1;
    }
    for (uint32_t c = 0; c < channels; c++) {                           /* InitChannels + Initialise */
        const uint8_t* p = training + c * channel_bytes + (channel_bytes - (uint64_t)in_samples * 4);   /* skip the oldest audio, :189-194 */
        for (uint32_t i = 0; i < count; i++) {                          /* :199-220 */
            const uint32_t s = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
            const int16_t s16 = (int16_t)(uint16_t)(s >> 16);
            p += 4 * dec;
            if (i >= count - degree) fb_samples[c][count - i - 1] = (int32_t)s;   /* initial states, newest first */
            in16[i] = (int16_t)(s16 >> kBurgDataDescaleBitCount);
        }
        int16_t burg[OHP_FLYWHEEL_DEGREE] = {0, 0, 0}, h[OHP_FLYWHEEL_DEGREE] = {0, 0, 0};
        memset(per, 0, count * sizeof(int16_t));
        memset(pef, 0, count * sizeof(int16_t));
        ohp_burgs_method(in16, count, degree, burg, h, per, pef);
        const int16_t excess = ohp_flywheel_coeff_overflow(burg, degree, kBurgOutputFormat);   /* CorrectBurgCoeffs, :333-340 */
        if (excess != 0) burg[0] = wrap16((int32_t)burg[0] - mul32(excess, 2));
        for (uint32_t i = 0; i < degree; i++)                           /* PrepareFeedbackCoeffs, :228-235 */
            fb_coeffs[c][i] = sub32(0, shl32((int32_t)burg[i], 16));
        ohp_feedback_init(&models[c], degree, kFeedbackDataDescaleBitCount, kBurgOutputFormat, kFeedbackDataFormat, 1,
                          fb_coeffs[c], fb_samples[c]);                 /* FlywheelRamper.cpp:149, 225 */
    }
    int32_t prev[OHP_FLYWHEEL_MAX_CHANNELS] = {0};
    uint32_t remaining = out_frames;
    uint8_t* o = out;
    while (remaining > 0) {                                             /* Ramp, :52-63: blocks of <= 1 ms */
        const uint32_t n = remaining > block_frames ? block_frames : remaining;
        remaining -= n;
        uint32_t hold = 0;                                              /* RenderChannels, :83-131 */
        for (uint32_t j = 0; j < n; j++) {
            for (uint32_t c = 0; c < channels; c++) {
                int32_t s;
                if (hold == 0) { s = ohp_feedback_next_sample(&models[c]); prev[c] = s; }
                else s = prev[c];
                o[0] = (uint8_t)((uint32_t)s >> 24); o[1] = (uint8_t)((uint32_t)s >> 16);
                o[2] = (uint8_t)((uint32_t)s >> 8); o[3] = (uint8_t)s;
                o += 4;
            }
            if (++hold == dec) hold = 0;
        }
    }
    free(in16); free(per); free(pef); free(fb_samples); free(fb_coeffs); free(models);
    return 0;
}
