/*
 * ohp_flywheel.h -- CPU restatement of ohPipeline's FlywheelRamper (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench legs named "cpu_baseline" may use this file; the product
 * (libohgpu.so, libohhost.so) never links or loads it.
 *
 * Follows, function by function, OpenHome/Media/FlywheelRamper.cpp of the reference (read as text; the reference
 * cannot be compiled here: ohNet headers are absent).  Pinned by the reference's own known-answer tests,
 * OpenHome/Media/Tests/TestFlywheelRamper.cpp:111-157 (FeedbackModel, Test1), :160-271 (scaling, Test2),
 * :274-520 (step / impulse / oscillator, Test3-5) and :535-612 (Burg's method, Test6), restated in
 * tests/test_oracle_flywheel_kats.py.
 *
 * Integer semantics: the reference computes in TInt16 / TInt32 with C's implicit conversions; where that overflows
 * (signed overflow is undefined in C) this file wraps in two's complement, which is what the reference's compilers do.
 */
#ifndef OHP_FLYWHEEL_H
#define OHP_FLYWHEEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHP_FLYWHEEL_DEGREE 3u            /* kDegree, FlywheelRamper.cpp:14 */
#define OHP_FLYWHEEL_MAX_CHANNELS 10u     /* kMaxChannelCount, FlywheelRamper.cpp:19 */

/* FlywheelRamper::BurgsMethod, FlywheelRamper.cpp:246-314.  per / pef are working arrays of `count` TInt16 that the
 * caller zeroes (the reference callocs them and clears them in Reset, :240-244). */
void ohp_burgs_method(const int16_t* samples, uint32_t count, uint32_t degree, int16_t* out, int16_t* h,
                      int16_t* per, int16_t* pef);

/* FlywheelRamper::DecimationFactor, FlywheelRamper.cpp:316-331 */
uint32_t ohp_flywheel_decimation_factor(uint32_t sample_rate);

/* FlywheelRamper::CoeffOverflow, FlywheelRamper.cpp:342-372 */
int16_t ohp_flywheel_coeff_overflow(const int16_t* coeffs, uint32_t count, uint32_t format);

/* FeedbackModel, FlywheelRamper.cpp:426-485.  `coeffs` and `samples` (state_count each) are borrowed; init descales
 * the samples in place exactly like FeedbackModel::Initialise. */
typedef struct {
    int32_t* coeffs;
    int32_t* samples;
    uint32_t state_count;
    uint32_t data_descale_bits;
    uint32_t coeff_format;
    int32_t  scale_shift_for_output;   /* aDataFormat + aDataDescaleBitCount - aOutputFormat */
} ohp_feedback_model;

void    ohp_feedback_init(ohp_feedback_model* m, uint32_t state_count, uint32_t data_descale_bits, uint32_t coeff_format,
                          uint32_t data_format, uint32_t output_format, int32_t* coeffs, int32_t* samples);
int32_t ohp_feedback_next_sample(ohp_feedback_model* m);

/* FlywheelRamperManager::Ramp, FlywheelRamper.cpp:44-66 with InitChannels :68-81, FlywheelRamper::Initialise :176-226
 * and RenderChannels :83-131.
 *   training : planar big-endian 32-bit audio, channel c at training + c * channel_bytes (FlywheelInput's layout,
 *              StarvationRamper.cpp:159-186); per channel the LAST in_samples * 4 bytes are used (Initialise skips
 *              older audio, :189-194)
 *   in_samples  = Jiffies::ToSamples(input jiffies, rate);  out_frames = ToSamples(output jiffies, rate);
 *   block_frames = ToSamples(kMaxOutputJiffiesBlockSize = 1 ms, rate): the sample-hold counter restarts per block
 *   out : interleaved big-endian 32-bit, out_frames * channels * 4 bytes
 * Returns 0, or -1 on a bad argument (channels outside 1..10, in_samples / decimation < degree + 1, ...). */
int ohp_flywheel_ramp(const uint8_t* training, uint64_t channel_bytes, uint32_t in_samples, uint32_t sample_rate,
                      uint32_t channels, uint32_t out_frames, uint32_t block_frames, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
