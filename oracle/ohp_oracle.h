/*
 * ohp_oracle.h -- CPU restatement of ohPipeline's PCM hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (ohpipeline_amd/, libohgpu.so)
 * must never call into it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - The reference's translation units cannot be compiled here: every one of them includes
 *     ohNet 1.40.5859 headers (projectdata/dependencies.json:2-17), an un-vendored binary
 *     dependency, and stand-ins for missing headers are not allowed.  The restatement is
 *     therefore pinned by the known-answer tests the reference's own suites hold for this
 *     path (OpenHome/Media/Tests/TestMsg.cpp SuiteRamp / SuiteMsgAudio / SuiteMsgPlayable)
 *     restated in tests/test_oracle_reference_kats.py.
 *   - Rows the reference tests pin only by property (exact ramped bytes, LE->BE with
 *     asymmetric data, 6-channel id bytes, pack/unpack helpers) are "property-pinned".
 *   - The resampler has no reference at all: PARITY UNPINNED.  Its specification is this
 *     file's ohp_src_* functions (exact integer model) checked against the fp64 model.
 *
 * All citations are file:line relative to /root/reference.
 */
#ifndef OHP_ORACLE_H
#define OHP_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (OpenHome/Media/Pipeline/Msg.h:117,193-194,258-259) ---- */
#define OHP_MAX_BYTES        9216u      /* AudioData::kMaxBytes            Msg.h:117 */
#define OHP_JIFFIES_PER_SEC  56448000u  /* Jiffies::kPerSecond             Msg.h:193 */
#define OHP_JIFFIES_PER_MS   56448u     /* Jiffies::kPerMs                 Msg.h:194 */
#define OHP_RAMP_MAX         16384u     /* Ramp::kMax = 1<<14              Msg.h:258 */
#define OHP_RAMP_MIN         0u         /* Ramp::kMin                      Msg.h:259 */
#define OHP_UNITY_ATTENUATION 256u      /* MsgAudioPcm::kUnityAttenuation  Msg.cpp:2219 */
#define OHP_RAMP_TABLE_COUNT 512u       /* kRampArrayCount                 RampArray.h:76 */

enum { OHP_ENDIAN_INVALID = 0, OHP_ENDIAN_LITTLE = 1, OHP_ENDIAN_BIG = 2 }; /* Msg.h:107-112 */
enum { OHP_RAMP_NONE = 0, OHP_RAMP_UP = 1, OHP_RAMP_DOWN = 2, OHP_RAMP_MUTE = 3 }; /* Msg.h:260-266 */

/* error codes returned where the reference would ASSERT / THROW */
#define OHP_OK                 0
#define OHP_ERR_ASSERT        (-1)   /* reference: ASSERT -> AssertionFailed */
#define OHP_ERR_SAMPLE_RATE   (-2)   /* reference: THROW(SampleRateInvalid)  Msg.cpp:472 */
#define OHP_ERR_UNSUPPORTED   (-3)   /* reference: THROW(CodecStreamFeatureUnsupported) Flac.cpp:404-407 */

/* ---- a7: ramp multiplier table (RampArray.h:7-74) ---- */
/* Generated, not transcribed: kRampArray[i] == min(32767, round(32768*(1-i/512)^2.5)),
 * evaluated exactly in integers as round(sqrt((512-i)^5 / 2^15)). */
const uint16_t* ohp_ramp_table(void);

/* ---- a1: DecodedAudio::ConstructPcm / CopyToBigEndian16/24/32 (Msg.cpp:347-408) ---- */
int ohp_construct_pcm(const uint8_t* src, uint32_t bytes, uint32_t bit_depth, int endian, uint8_t* dst);

/* ---- a2: Jiffies (Msg.cpp:424-502; Msg.h:193-237) ---- */
int      ohp_jiffies_per_sample(uint32_t sample_rate);   /* >0, or OHP_ERR_SAMPLE_RATE */
uint32_t ohp_jiffies_to_bytes(uint32_t* jiffies, uint32_t jiffies_per_sample, uint32_t channels, uint32_t bits);
uint32_t ohp_jiffies_to_bytes_sample_block(uint32_t* jiffies, uint32_t jps, uint32_t channels, uint32_t bits, uint32_t samples_per_block);
int      ohp_jiffies_round_down(uint32_t* jiffies, uint32_t sample_rate);
int      ohp_jiffies_round_up(uint32_t* jiffies, uint32_t sample_rate);
void     ohp_jiffies_round_down_nonzero_sample_block(uint32_t* jiffies, uint32_t block_jiffies);
int      ohp_jiffies_to_songcast_time(uint32_t jiffies, uint32_t sample_rate, uint32_t* out);

/* ---- a3: Ramp (Msg.h:253-286; Msg.cpp:569-807) ---- */
typedef struct {
    uint32_t start;
    uint32_t end;
    uint32_t direction;   /* OHP_RAMP_* */
    uint32_t enabled;
} ohp_ramp;

void ohp_ramp_reset(ohp_ramp* r);                                     /* Msg.cpp:582-588 */
/* returns 1 iff *split is set, 0 otherwise, OHP_ERR_ASSERT where the reference asserts */
int  ohp_ramp_set(ohp_ramp* r, uint32_t start, uint32_t fragment_size, uint32_t remaining_duration,
                  uint32_t direction, ohp_ramp* split, uint32_t* split_pos);  /* Msg.cpp:590-712 */
void ohp_ramp_set_muted(ohp_ramp* r);                                 /* Msg.cpp:714-719 */
int  ohp_ramp_validate(const ohp_ramp* r);                            /* Msg.cpp:745-782: 1 valid, 0 invalid */
int  ohp_ramp_split(ohp_ramp* r, uint32_t new_size, uint32_t current_size, ohp_ramp* remaining); /* Msg.cpp:784-807 */
uint32_t ohp_ramp_median_multiplier(const ohp_ramp* r);               /* Msg.cpp:901-920 */

/* ---- a4/a5: MsgAudio metadata model (Msg.cpp:1949-2074, 2234-2262) ---- */
typedef struct {
    uint32_t size_jiffies;     /* iSize   */
    uint32_t offset_jiffies;   /* iOffset */
    uint32_t sample_rate;
    uint32_t bit_depth;
    uint32_t channels;
    uint32_t attenuation;      /* MsgAudioPcm::iAttenuation; OHP_UNITY_ATTENUATION by default */
    uint32_t is_silence;       /* 1 for MsgSilence */
    ohp_ramp ramp;
} ohp_msg_audio;

int ohp_msg_audio_init_pcm(ohp_msg_audio* m, uint32_t data_bytes, uint32_t channels, uint32_t sample_rate, uint32_t bit_depth); /* Msg.cpp:2264-2276,2155-2168 */
int ohp_msg_audio_init_silence(ohp_msg_audio* m, uint32_t* jiffies, uint32_t sample_rate, uint32_t bit_depth, uint32_t channels); /* Msg.cpp:2547-2560 */
int ohp_msg_audio_split(ohp_msg_audio* m, uint32_t jiffies, ohp_msg_audio* remaining);           /* Msg.cpp:1949-1969 */
/* returns new current ramp value (iRamp.End()); *has_split=1 if *split was produced */
int ohp_msg_audio_set_ramp(ohp_msg_audio* m, uint32_t start, uint32_t* remaining_duration, uint32_t direction,
                           ohp_msg_audio* split, int* has_split, uint32_t* ramp_end_out);       /* Msg.cpp:1989-2046 */

typedef struct {
    uint32_t offset_bytes;
    uint32_t size_bytes;
    uint32_t jiffies;
    uint32_t sample_rate;
    uint32_t bit_depth;
    uint32_t channels;
    uint32_t attenuation;
    uint32_t is_silence;       /* EMute or MsgSilence => MsgPlayableSilence */
    ohp_ramp ramp;
} ohp_playable;

int ohp_create_playable(const ohp_msg_audio* m, ohp_playable* p);     /* Msg.cpp:2234-2262, 2466-2491 */
int ohp_playable_split(ohp_playable* p, uint32_t bytes, ohp_playable* remaining, int* has_remaining); /* Msg.cpp:2591-2624 */

/* ---- a6: MsgPlayablePcm::ApplyAttenuation (Msg.cpp:2736-2751) ---- */
int ohp_apply_attenuation(uint8_t* data, uint32_t bytes, uint32_t bit_depth, uint32_t attenuation);

/* ---- a7: RampApplicator (Msg.cpp:812-899) ---- */
int ohp_ramp_apply(const uint8_t* src, uint32_t bytes, uint32_t bit_depth, uint32_t channels,
                   uint32_t ramp_start, uint32_t ramp_end, uint8_t* dst);

/* ---- a8/a9: MsgPlayable::Read (Msg.cpp:2646-2653, 2753-2786, 2874-2893) ----
 * Reads a playable whose audio lives at audio[offset_bytes .. offset_bytes+size_bytes).
 * NOTE the reference attenuates IN PLACE (Msg.cpp:2742-2750); so does this (audio is mutable).
 * out receives the concatenation of every fragment; frag_sizes (may be NULL) receives each
 * ProcessFragment/ProcessSilence size, *n_frags their count (capacity max_frags). */
int ohp_playable_read(const ohp_playable* p, uint8_t* audio, uint8_t* out, uint32_t out_capacity,
                      uint32_t* frag_sizes, uint32_t max_frags, uint32_t* n_frags, uint32_t* out_bytes);

/* ---- a11: FlywheelInput::DoProcessFragment (StarvationRamper.cpp:117-186) ----
 * packed BE interleaved -> planar 4-byte BE left-justified.  chan_out[j] points at channel j's
 * plane; *chan_pos (array of channels) carries the append position in bytes for each plane. */
int ohp_flywheel_unpack(const uint8_t* data, uint32_t bytes, uint32_t channels, uint32_t subsample_bytes,
                        uint8_t* planes, uint32_t plane_stride_bytes, uint32_t* plane_pos);

/* ---- a12: RampGenerator::ProcessFragment (StarvationRamper.cpp:281-327) ---- */
int ohp_rampgen_pack(const uint8_t* data, uint32_t bytes, uint32_t bit_depth, uint8_t* dst, uint32_t* dst_bytes);

/* ---- a13: Sender::DoProcessFragment (Av/Songcast/Sender.cpp:351-377) ---- */
int ohp_sender_pack(const uint8_t* data, uint32_t bytes, uint32_t channels, uint32_t bytes_per_subsample,
                    uint8_t* dst, uint32_t* dst_bytes);

/* ---- a14: CodecFlac::CallbackWrite packer (Codec/Flac.cpp:379-417) ---- */
int ohp_flac_pack(const int32_t* const* planes, uint32_t channels, uint32_t first, uint32_t samples,
                  uint32_t bit_depth, uint8_t* dst, uint32_t* dst_bytes);

/* ---- a-R: polyphase sample-rate converter -- NO REFERENCE, PARITY UNPINNED ----
 * Specification (this build's own):
 *   ratio L/M = rate_out/rate_in reduced; T taps per phase; prototype length N = L*T,
 *   Kaiser(beta)-windowed sinc, pass edge f_pass, stop edge f_stop = rate_out - f_pass,
 *   cutoff (f_pass+f_stop)/2, DC gain L;  coefficients quantised to Q28 int32:
 *        c[p*T + k] = floor(h[p + k*L] * 2^28 + 0.5)
 *   output m: t = m*M, n0 = t / L, p = t % L,
 *        acc = sum_{k<T} c[p*T+k] * x[n0-k]          (x = S24 sample, x[<0] = history/zeros)
 *        y   = clamp((acc + 2^27) >> 28, -2^23, 2^23-1)
 *   The exact-integer model is the parity target (GPU must be bit-exact to it); the fp64
 *   model with UNQUANTISED coefficients is the +/-1 LSB yardstick.                        */
typedef struct {
    uint32_t L, M, T;
    double   beta, f_pass, f_stop;
    uint32_t rate_in, rate_out;
    int32_t* coef_q28;     /* [L][T]  */
    double*  coef_f64;     /* [L][T], unquantised */
    int64_t  sum_abs_max;  /* max over phases of sum |c| (must be < 2^30 for exact fp64 accumulation) */
} ohp_src;

int  ohp_src_design(ohp_src* s, uint32_t rate_in, uint32_t rate_out, uint32_t taps_per_phase, double beta, double f_pass);
void ohp_src_free(ohp_src* s);
/* number of output frames produced once in_frames_total input frames have been consumed, counting from 0 */
uint64_t ohp_src_out_frames(const ohp_src* s, uint64_t in_frames_total);
/* Process output frames [m0, m0+n_out) of one stream.  x is planar-free interleaved S24 held in int32
 * (x[frame*channels + c]), indexed from absolute input frame x_first; frames before 0 are zeros.
 * Frames needed: n0(m0)-T+1 .. n0(m0+n_out-1); the caller guarantees they lie in [x_first, x_first+x_frames)
 * or before 0. */
int ohp_src_process_i64(const ohp_src* s, const int32_t* x, int64_t x_first, uint64_t x_frames, uint32_t channels,
                        uint64_t m0, uint32_t n_out, int32_t* y);
int ohp_src_process_f64(const ohp_src* s, const int32_t* x, int64_t x_first, uint64_t x_frames, uint32_t channels,
                        uint64_t m0, uint32_t n_out, double* y);

/* ---- helpers used by the fused "resample -> ramp -> fmt" oracle ---- */
/* unpack packed PCM (LE or BE, 8/16/24/32 bit) to S24-in-int32 (left-justified to 32 then >>8) */
int ohp_unpack_s24(const uint8_t* src, uint32_t subsamples, uint32_t bit_depth, int endian, int32_t* dst);
/* pack S24-in-int32 to packed BE at bit_depth (8/16/24 truncate low bytes, 32 = 24 + zero LSB;
 * a11 o a12 composition) then optionally byte-swap to LE */
int ohp_pack_from_s24(const int32_t* src, uint32_t subsamples, uint32_t bit_depth, int endian, uint8_t* dst);
/* generic depth/endian converter on packed data = a1 (to BE) o a11 (left-justify to 4 bytes) o truncate to
 * dst depth [+ endian swap].  zero_lsb32 != 0 additionally reproduces a12's "case 32" (StarvationRamper.cpp:
 * 311-320), which writes a zero least-significant byte; without it equal depths pass through untouched, as
 * they do in the reference when no converting IPcmProcessor sits downstream. */
int ohp_convert_format(const uint8_t* src, uint32_t subsamples, uint32_t src_bits, int src_endian,
                       uint32_t dst_bits, int dst_endian, int zero_lsb32, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif /* OHP_ORACLE_H */
