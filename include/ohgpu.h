/*
 * ohgpu.h -- C ABI of the MI355X-native PCM hot path of ohPipeline.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ types, no torch types, never
 * throws.  Every entry point returns OHGPU_OK (0) or a negative OHGPU_ERR_* code; the message
 * for the most recent failure on the calling thread is available from ohgpu_last_error().
 *
 * What it stands in for (file:line relative to the reference tree, openhome/ohPipeline):
 *
 *   reference call chain (per message, on the animator thread)          replaced by
 *   ------------------------------------------------------------------  ---------------------------
 *   MsgFactory::CreateMsgAudioPcm -> DecodedAudio::ConstructPcm          src_endian in the descriptor
 *       (OpenHome/Media/Pipeline/Msg.cpp:3961-3965, 347-408)             (LE->BE fused into the load)
 *   MsgAudioPcm::CreatePlayable (Msg.cpp:2234-2262)                      host fills src_offset/n_frames
 *   MsgPlayable::Read(IPcmProcessor&) (Msg.cpp:2646-2653)                ohgpu_pcm_batch_run()
 *     MsgPlayablePcm::ApplyAttenuation (Msg.cpp:2736-2751)                 attenuation field
 *     RampApplicator::GetNextSample    (Msg.cpp:832-899)                   OHGPU_FLAG_RAMP + ramp_start/end
 *     MsgPlayableSilence::ReadBlock    (Msg.cpp:2874-2893)                 OHGPU_FLAG_SILENCE
 *   IPcmProcessor::ProcessFragment doing depth conversion                dst_bits / dst_endian
 *     FlywheelInput::AppendSubsample8/16/24/32 (StarvationRamper.cpp:117-186)
 *     RampGenerator::ProcessFragment           (StarvationRamper.cpp:281-327)
 *   "SampleRateConverter" -- NOT PRESENT in the reference  ohgpu_src_* (own spec)
 *
 * The reference binds nothing through FFI today (it is one C++ static library); INTEGRATION.md
 * shows the adapter a maintainer would add: an IPcmProcessor-shaped C++ shim over this ABI.
 *
 * Threading: an ohgpu_ctx may be used from one thread at a time (the reference's element
 * contract is the same: one puller thread per element, Msg.h:1844-1849).  Different contexts
 * are independent.  All *_run calls are asynchronous on the given stream.
 */
#ifndef OHGPU_H
#define OHGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHGPU_ABI_VERSION 1

/* ---- error codes ---- */
#define OHGPU_OK                0
#define OHGPU_ERR_INVALID      (-1)  /* bad argument / descriptor fails validation (reference: ASSERT) */
#define OHGPU_ERR_DEVICE       (-2)  /* HIP runtime error (message holds hipGetErrorString)            */
#define OHGPU_ERR_NO_DEVICE    (-3)  /* no GPU visible; the product path never falls back to the CPU   */
#define OHGPU_ERR_NOMEM        (-4)
#define OHGPU_ERR_BOUNDS       (-5)  /* descriptor addresses bytes outside the declared arenas         */
#define OHGPU_ERR_UNSUPPORTED  (-6)  /* e.g. attenuation on non-16-bit audio (Msg.cpp:2741)            */

/* ---- enums mirroring the reference ---- */
#define OHGPU_ENDIAN_LITTLE 1   /* AudioDataEndian::Little  Msg.h:107-112 */
#define OHGPU_ENDIAN_BIG    2   /* AudioDataEndian::Big */

#define OHGPU_RAMP_MAX 16384u          /* Ramp::kMax  Msg.h:258 */
#define OHGPU_UNITY_ATTENUATION 256u   /* MsgAudioPcm::kUnityAttenuation  Msg.cpp:2219 */
#define OHGPU_MAX_CHANNELS 8u          /* DecodedAudio::kMaxNumChannels   Msg.h:170 (Sender maps 10 -> handled on host) */

/* ---- descriptor flags ---- */
#define OHGPU_FLAG_RAMP       0x01u  /* Ramp::IsEnabled(): apply RampApplicator semantics (16-bit, low bytes zeroed) */
#define OHGPU_FLAG_SILENCE    0x02u  /* MsgPlayableSilence: emit zeros (+ the 6-channel id bytes of Msg.cpp:2877);
                                        src is not read */
#define OHGPU_FLAG_SRC_PLANAR32 0x08u /* resampled messages only: the source is what CodecFlac::CallbackWrite is handed (Codec/Flac.cpp:
                                         379-417) -- one plane of host-endian TInt32 per channel, sample values at src_bits depth,
                                         sign-extended -- instead of the packed bytes that callback makes of it: channel c's plane
                                         starts at src_offset + c * src_plane_stride, a frame is 4 bytes.  a14 -> a1 -> a-R in one pass */
#define OHGPU_FLAG_ZERO_LSB32 0x04u  /* RampGenerator::ProcessFragment "case 32" (StarvationRamper.cpp:311-320):
                                        when dst_bits == 32 write a zero least-significant byte */

/*
 * One MsgPlayable (Msg.cpp:2684-2696, 2722-2734) flattened for the device.  32 bytes.
 * Audio is packed, interleaved; src_offset already includes MsgPlayable::iOffset.
 * Output is packed interleaved at dst_bits, big endian unless dst_endian says otherwise.
 */
typedef struct ohgpu_msg_desc {
    uint64_t src_offset;    /* bytes from src_base to the first frame                               */
    uint64_t dst_offset;    /* bytes from dst_base to the first output frame                        */
    uint32_t n_frames;      /* sample instants ("samples" in the reference, Msg.cpp:826)            */
    uint16_t ramp_start;    /* Ramp::Start()  [0, 16384]                                            */
    uint16_t ramp_end;      /* Ramp::End()                                                          */
    uint16_t attenuation;   /* 256 = unity; other values only with src_bits == 16 (Msg.cpp:2741)    */
    uint8_t  channels;      /* 1..8                                                                 */
    uint8_t  src_bits;      /* 8, 16, 24, 32                                                        */
    uint8_t  src_endian;    /* OHGPU_ENDIAN_*                                                       */
    uint8_t  dst_bits;      /* 8, 16, 24, 32                                                        */
    uint8_t  dst_endian;    /* OHGPU_ENDIAN_*                                                       */
    uint8_t  flags;         /* OHGPU_FLAG_*                                                         */
} ohgpu_msg_desc;

/*
 * One OUTPUT message of a sample-rate-converted stream.  64 bytes.
 * The input buffer holds frames [src_frame0, src_frame0 + src_frames) of the stream starting at
 * src_offset; frames with a negative absolute index are zeros (stream start); every other frame the
 * filter needs, n0(out_frame0) - T + 1 .. n0(out_frame0 + n_frames - 1), must be present.
 * The resampler runs in the S24 domain (sources are left-justified to 24 bits first), so ramping
 * follows RampApplicator's 24-bit case and attenuation must be unity.
 */
typedef struct ohgpu_src_msg_desc {
    uint64_t src_offset;    /* bytes from src_base to input frame src_frame0                        */
    uint64_t src_frame0;    /* absolute index of the first input frame held in the buffer           */
    uint64_t src_frames;    /* number of input frames held                                          */
    uint64_t out_frame0;    /* absolute index of this message's first output frame                  */
    uint64_t dst_offset;    /* bytes from dst_base                                                  */
    uint32_t n_frames;      /* output frames in this message                                        */
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint16_t attenuation;   /* must be 256                                                          */
    uint8_t  channels;
    uint8_t  src_bits;
    uint8_t  src_endian;
    uint8_t  dst_bits;
    uint8_t  dst_endian;
    uint8_t  flags;         /* OHGPU_FLAG_RAMP | OHGPU_FLAG_ZERO_LSB32 | OHGPU_FLAG_SRC_PLANAR32    */
    uint64_t src_plane_stride; /* OHGPU_FLAG_SRC_PLANAR32: bytes between the channels' planes, multiple of 4 (else 0) */
} ohgpu_src_msg_desc;

typedef struct ohgpu_ctx   ohgpu_ctx;     /* one per GPU / per pipeline thread            */
typedef struct ohgpu_batch ohgpu_batch;   /* validated, device-resident descriptor batch  */
typedef struct ohgpu_src   ohgpu_src;     /* a designed polyphase filter on the device    */

/* ---- library / device ---- */
int         ohgpu_abi_version(void);
const char* ohgpu_last_error(void);
int         ohgpu_device_count(void);                       /* >= 0, or OHGPU_ERR_DEVICE */
int         ohgpu_init(int device, ohgpu_ctx** ctx);         /* OHGPU_ERR_NO_DEVICE when there is no GPU */
int         ohgpu_shutdown(ohgpu_ctx* ctx);
int         ohgpu_device_name(ohgpu_ctx* ctx, char* buf, size_t buf_bytes);
/* The device's PCI address, "0000:c1:00.0", lower case: a host that runs one rank per GPU finds the CPUs nearest the device under
 * /sys/bus/pci/devices/<address>/local_cpulist (bench.py pins each rank's planner and feeder threads there). */
int         ohgpu_device_pci_bus_id(ohgpu_ctx* ctx, char* buf, size_t buf_bytes);

/* ---- plumbing: device memory, streams, events (thin wrappers, so hosts need no HIP headers) ---- */
int ohgpu_malloc(ohgpu_ctx* ctx, size_t bytes, void** dptr);
int ohgpu_free(ohgpu_ctx* ctx, void* dptr);
int ohgpu_malloc_host(ohgpu_ctx* ctx, size_t bytes, void** hptr);   /* pinned */
int ohgpu_free_host(ohgpu_ctx* ctx, void* hptr);
int ohgpu_memcpy_h2d(ohgpu_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int ohgpu_memcpy_d2h(ohgpu_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int ohgpu_memset(ohgpu_ctx* ctx, void* dptr, int value, size_t bytes, void* stream);
int ohgpu_stream_create(ohgpu_ctx* ctx, void** stream);
int ohgpu_stream_destroy(ohgpu_ctx* ctx, void* stream);
int ohgpu_stream_sync(ohgpu_ctx* ctx, void* stream);                /* NULL = the context's own stream */
int ohgpu_event_create(ohgpu_ctx* ctx, void** event);
int ohgpu_event_destroy(ohgpu_ctx* ctx, void* event);
int ohgpu_event_record(ohgpu_ctx* ctx, void* event, void* stream);
int ohgpu_stream_wait_event(ohgpu_ctx* ctx, void* stream, void* event);          /* work queued on stream after this waits for event */
int ohgpu_event_elapsed_ms(ohgpu_ctx* ctx, void* start, void* stop, float* ms);  /* synchronises on stop */

/* ---- RampArray.h:7-74: the 512 Q15 multipliers the device uses (generated) ---- */
int ohgpu_ramp_table(uint16_t out[512]);

/* ---- unpack -> attenuate -> ramp -> pack over a batch of playables ---- */
/* Validates every descriptor against the arena sizes (OHGPU_ERR_BOUNDS / _INVALID / _UNSUPPORTED),
 * then keeps a device-resident copy.  descs is host memory and may be freed after the call. */
int ohgpu_pcm_batch_create(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** batch);
/* Launches the batch: src_base / dst_base are DEVICE pointers to arenas at least as large as declared. */
int ohgpu_pcm_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream);
int ohgpu_batch_destroy(ohgpu_ctx* ctx, ohgpu_batch* batch);
/* Totals recorded at creation (for throughput accounting). */
int ohgpu_batch_info(const ohgpu_batch* batch, uint64_t* n_msgs, uint64_t* in_frames, uint64_t* out_frames,
                     uint64_t* src_bytes_touched, uint64_t* dst_bytes_written);

/* Convenience for hosts that hold host buffers (a live pipeline's 5 ms cadence: the driver thread's MsgPlayable::Read of a
 * period, Msg.cpp:2646-2653, for as many playables as the caller brings): H2D, run, D2H, sync.  Of dst_host only the bytes the
 * messages' outputs cover are written (in ONE copy when the outputs tile a span of it, whatever their order).  The device arenas
 * the audio passes through belong to the context and are kept from call to call -- a steady caller allocates nothing per period
 * (ohgpu_device_allocations) -- and src_host / dst_host may be pageable or pinned (ohgpu_malloc_host: no staging copy).
 * The same holds for the three *_process_host calls below. */
int ohgpu_pcm_process_host(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes);

/* ---- the in-tree IPcmProcessor conversions that are not plain interleaved -> interleaved ---- */
#define OHGPU_FMT_UNPACK_PLANAR 1  /* FlywheelInput::DoProcessFragment + AppendSubsample8/16/24/32 (Pipeline/StarvationRamper.cpp:
                                      117-186): packed BE interleaved -> PLANAR 4-byte BE, left-justified, low bytes zero;
                                      channel c's plane starts at dst_offset + c * dst_plane_stride */
#define OHGPU_FMT_SENDER_PACK   2  /* Sender::DoProcessFragment (Av/Songcast/Sender.cpp:351-377): first two channels (from
                                      channel 8 when there are >= 10), min(bytes, 3) most significant bytes each */
#define OHGPU_FMT_FLAC_PACK     3  /* CodecFlac::CallbackWrite (Codec/Flac.cpp:379-417): planar host-endian TInt32 (channel c at
                                      src_offset + c * src_plane_stride) -> packed BE interleaved 8/16/24 bit */

typedef struct ohgpu_fmt_desc {     /* 48 bytes */
    uint64_t src_offset;
    uint64_t dst_offset;
    uint64_t src_plane_stride;      /* FLAC_PACK only   */
    uint64_t dst_plane_stride;      /* UNPACK_PLANAR only */
    uint32_t n_frames;
    uint8_t  kind;                  /* OHGPU_FMT_* */
    uint8_t  channels;              /* 1..10 (Sender is told about up to 10) */
    uint8_t  src_bits;              /* packed depth of the source (UNPACK/SENDER); 32 for FLAC_PACK's TInt32 planes */
    uint8_t  dst_bits;              /* FLAC_PACK: 8/16/24; ignored otherwise (UNPACK -> 32, SENDER -> min(src,24)) */
    uint8_t  reserved[8];
} ohgpu_fmt_desc;

int ohgpu_fmt_batch_create(ohgpu_ctx* ctx, const ohgpu_fmt_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** batch);
int ohgpu_fmt_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream);

/* ---- FlywheelRamper ----
 * Replaces FlywheelRamperManager::Ramp (OpenHome/Media/FlywheelRamper.cpp:44-66; per channel FlywheelRamper::Initialise
 * :176-226 = decimate, Burg's method :246-314, coefficient correction :333-372, and FeedbackModel::NextSample :449-487;
 * rendering with sample hold :83-131) for a batch of starving streams: one descriptor = one stream's ramp request.
 * Training audio is what FlywheelInput prepares (StarvationRamper.cpp:90-111, 159-186 = OHGPU_FMT_UNPACK_PLANAR):
 * planar big-endian 32-bit, channel c at src_offset + c * channel_bytes; of each plane the LAST in_samples * 4 bytes
 * are used (Initialise skips older audio, :189-194).  Output: interleaved big-endian 32-bit, what RenderChannels hands
 * to IPcmProcessor::ProcessFragment(buf, channels, 4) in blocks of block_frames. */
typedef struct ohgpu_flywheel_desc {     /* 48 bytes */
    uint64_t src_offset;
    uint64_t channel_bytes;         /* bytes per channel plane, >= in_samples * 4 */
    uint64_t dst_offset;            /* out_frames * channels * 4 bytes are written here */
    uint32_t in_samples;            /* Jiffies::ToSamples(training jiffies, rate); in_samples / decimation >= 4 */
    uint32_t out_frames;            /* Jiffies::ToSamples(ramp jiffies, rate) */
    uint32_t block_frames;          /* Jiffies::ToSamples(kMaxOutputJiffiesBlockSize = 1 ms, rate): the hold counter restarts per block */
    uint32_t sample_rate;           /* decides the decimation factor (FlywheelRamper.cpp:316-331); <= 384000 */
    uint32_t channels;              /* 1..10 (kMaxChannelCount) */
    uint32_t reserved;
} ohgpu_flywheel_desc;

int ohgpu_flywheel_batch_create(ohgpu_ctx* ctx, const ohgpu_flywheel_desc* descs, size_t n,
                                uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** batch);
int ohgpu_flywheel_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream);
/* Host-buffer convenience: H2D, run, D2H, sync (dst_host bytes that no request covers are preserved). */
int ohgpu_flywheel_process_host(ohgpu_ctx* ctx, const ohgpu_flywheel_desc* descs, size_t n,
                                const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes);

/* ---- Songcast sender frames ----
 * Replaces, for a batch of 5 ms packets of many streams, what Sender::SendPendingAudio (Av/Songcast/Sender.cpp:307-321)
 * and OhmSenderDriver::SendAudio (Av/Songcast/OhmSender.cpp:418-480) do per packet on the pipeline thread:
 *   MsgPlayable::Read of every pending message (attenuation, ramp; Msg.cpp:2646-2653)      the fragment's ramp fields
 *   Sender::DoProcessFragment (Sender.cpp:351-377): first two channels, <= 3 bytes each      fused into the same pass
 *   OhmMsgAudio::ReinitialiseFields + Serialise (OhmMsg.cpp:203-223, 363-413),
 *   OhmHeader::Externalise (Ohm.cpp:44-52), OhmMsgAudio::GetStreamHeader (OhmMsg.cpp:225-241)  the frame header kernel
 * Each frame is written to dst_base + dst_offset as the datagram OhmMsgAudio::SendableBuffer hands to the socket:
 *   "Ohm " 01 03 <total:2> | 32 <flags> <samples:2> <frame:4> <network ts:4> <media latency:4> <media ts:4> <sample start:8>
 *   | <samples total:8> <sample rate:4> <bit rate:4> <volume offset:2> <bit depth> <channels> 00 <codec len> <codec> | audio
 * (big endian).  Sending it, resend history and timestamping stay with the host. */
#define OHGPU_OHM_FLAG_HALT        0x01u   /* OhmMsgAudio::kFlagHalt .. kFlagResent, OhmMsg.h:67-70 */
#define OHGPU_OHM_FLAG_LOSSLESS    0x02u
#define OHGPU_OHM_FLAG_TIMESTAMPED 0x04u   /* the frame also gets kFlagTimestamped2 (OhmMsg.cpp:211) */
#define OHGPU_OHM_FLAG_RESENT      0x08u
#define OHGPU_OHM_MAX_CODEC_BYTES  29u     /* OhmMsgAudio::kMaxCodecBytes */
#define OHGPU_OHM_MAX_AUDIO_BYTES  5760u   /* OhmMsgAudio::kMaxSampleBytes */

typedef struct ohgpu_ohm_stream {   /* 64 bytes: what Sender::ProcessMsg(MsgDecodedStream*) (Sender.cpp:217-242) fixes for a stream */
    uint64_t samples_total;         /* TrackLength / Jiffies::PerSample */
    uint32_t sample_rate;
    uint32_t bit_rate;
    int16_t  volume_offset;         /* OhmSenderDriver::SetAudioFormat passes 0 */
    uint8_t  src_channels;          /* 1..10: the pipeline's channel count; the wire carries min(channels, 2), taken from
                                       channel 0, or channel 8 when there are >= 10 (Sender::FirstChannelToSend) */
    uint8_t  src_bits;              /* 8/16/24/32: the pipeline's depth; the wire carries min(bits, 24) */
    uint8_t  codec_bytes;           /* 0..29 */
    uint8_t  codec[29];
    uint8_t  src_endian;            /* 0 or OHGPU_ENDIAN_BIG: audio as DecodedAudio stores it (always big endian in the
                                       reference); OHGPU_ENDIAN_LITTLE: codec output not yet swapped (row a1 fused in) */
    uint8_t  reserved[13];
} ohgpu_ohm_stream;

typedef struct ohgpu_ohm_fragment { /* 24 bytes: one pending message's MsgPlayable read into the frame (Sender.cpp:312-316) */
    uint64_t src_offset;            /* packed big-endian interleaved audio at the stream's src_channels / src_bits */
    uint32_t n_frames;
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint16_t attenuation;           /* 256 = unity; other values only on 16-bit audio */
    uint8_t  flags;                 /* OHGPU_FLAG_RAMP | OHGPU_FLAG_SILENCE */
    uint8_t  reserved[5];
} ohgpu_ohm_fragment;

typedef struct ohgpu_ohm_frame_desc {   /* 48 bytes: one OhmSenderDriver::SendAudio */
    uint64_t dst_offset;            /* the datagram starts here; ohgpu_ohm_frame_layout gives its size */
    uint64_t sample_start;          /* iSampleStart */
    uint32_t stream;                /* index into streams[] */
    uint32_t frame;                 /* iFrame */
    uint32_t network_timestamp;
    uint32_t media_latency;         /* iLatencyOhm */
    uint32_t media_timestamp;       /* 0 from this sender (OhmMsg.cpp:217) */
    uint32_t first_fragment;        /* fragments[first_fragment .. first_fragment + n_fragments) make up the audio, in order */
    uint16_t n_fragments;
    uint8_t  flags;                 /* OHGPU_OHM_FLAG_* */
    uint8_t  reserved[5];
} ohgpu_ohm_frame_desc;

/* Sizes of a frame of `samples` sample instants: the header (8 + 50 + codec_bytes) and the whole datagram. */
int ohgpu_ohm_frame_layout(const ohgpu_ohm_stream* stream, uint32_t samples, uint32_t* header_bytes, uint32_t* frame_bytes);
/* OHGPU_ERR_INVALID where the reference asserts: more audio than OhmMsgAudio::kMaxSampleBytes in a frame (Sender.cpp:364),
 * a codec name over 29 bytes; OHGPU_ERR_UNSUPPORTED for ramped/silent/attenuated fragments of more than 8 channels. */
int ohgpu_ohm_batch_create(ohgpu_ctx* ctx, const ohgpu_ohm_stream* streams, size_t n_streams,
                           const ohgpu_ohm_frame_desc* frames, size_t n_frames,
                           const ohgpu_ohm_fragment* fragments, size_t n_fragments,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** batch);
int ohgpu_ohm_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream);
/* Host-buffer convenience: H2D, run, D2H, sync (dst_host bytes that no frame covers are preserved). */
int ohgpu_ohm_process_host(ohgpu_ctx* ctx, const ohgpu_ohm_stream* streams, size_t n_streams,
                           const ohgpu_ohm_frame_desc* frames, size_t n_frames,
                           const ohgpu_ohm_fragment* fragments, size_t n_fragments,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes);

/* ---- sample-rate converter (own specification; DESIGN.md "Resampler") ---- */
/* Host-side filter design: Kaiser-windowed sinc, Q28 coefficients, coef_q28[p*T + k] = h[p + k*L].
 * Pass coef_q28 = NULL to query L, M only.  Capacity must be >= L*T. */
int ohgpu_src_design(uint32_t rate_in, uint32_t rate_out, uint32_t taps_per_phase, double beta, double f_pass_hz,
                     int32_t* coef_q28, size_t coef_capacity, uint32_t* L, uint32_t* M);
int ohgpu_src_create(ohgpu_ctx* ctx, uint32_t L, uint32_t M, uint32_t taps_per_phase, const int32_t* coef_q28, ohgpu_src** src);
int ohgpu_src_destroy(ohgpu_ctx* ctx, ohgpu_src* src);
/* ceil(in_frames * L / M): output frames available once in_frames input frames have arrived */
uint64_t ohgpu_src_out_frames(uint32_t L, uint32_t M, uint64_t in_frames);
/* Host only, no device needed: the tables ohgpu_src_create makes for the matrix-pipe resampler kernel (24-bit stereo,
 * ohpipeline_amd/csrc/src_mfma_kernel.hip) -- the coefficients' balanced base-256 digits as padded rows
 * [4 digits][L][96] bytes, and one 288-byte record per 16-output step of a block row (A-row offsets, the accumulators'
 * initial values, the window's first 16-frame chunk).  *block_outputs = outputs per block; steps cover rows of up to
 * max_blocks_per_row blocks.  Query sizes with NULL buffers.  OHGPU_ERR_UNSUPPORTED when the filter does not fit the
 * tiling (taps_per_phase != 32, or a ratio whose 16-output tiles need more than 64 input frames).  Tests check the
 * tables against the integer model on the CPU; the kernel is checked on the device. */
int ohgpu_src_mfma_tables(uint32_t L, uint32_t M, uint32_t taps_per_phase, const int32_t* coef_q28, uint32_t max_blocks_per_row,
                          uint8_t* coef_digits, size_t coef_digits_capacity, void* steps, size_t steps_capacity,
                          size_t* coef_digits_bytes, size_t* steps_bytes, uint32_t* block_outputs);
/* ... and the tables it makes for the same kernel's half-band form (ohpipeline_amd/csrc/src_mfma_wg_kernel.hip, HB) when the
 * filter is a 2:1 decimator of 64 stored taps whose odd taps but the centre one (31) are zero: ONE image of the B operand's
 * coefficient digits, [4 digits][lane = 16 g + n][16 bytes] -- K groups g = 0..2 meet the even input frames 16 (s + g) .. + 15 of
 * step s (a row's image starts 64 frames before its block), group 3 the odd frames 16 (s + 1) .. + 15, output n taking sample n --
 * and the accumulators' initial value 32896 * sum(c) + 2^27.  *block_outputs = outputs per block (128).
 * OHGPU_ERR_UNSUPPORTED when the coefficients are not such a filter.  Host only. */
int ohgpu_src_mfma_halfband_tables(const int32_t* coef_q28 /* 64 */, uint8_t* image /* 4096 bytes */, int64_t* bias, uint32_t* block_outputs);
/* The messages of one batch may differ in layout (channels, depths, byte orders, packed or planar source): the batch is
 * planned per layout and runs one launch sequence per layout, messages of a stream in the order given.
 * What a batch keeps: its plan (a record per unit of 30-32 blocks, per ramped message and unit, per block-unaligned message end) on
 * the device and a copy of the ramp records on the host -- nothing per message, unless it is the generic kernel's batch (created
 * while ohgpu_set_kernel_variant(1) is in force, or of a layout no block kernel has: 56 bytes a message on the host, and on the
 * device from its first run). */
int ohgpu_src_batch_create(ohgpu_ctx* ctx, const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** batch);
/* A batch may be run any number of times, one launch at a time (it owns device-side work counters): launches of the same
 * batch on one stream queue behind each other; a launch on ANOTHER stream while the previous one has not finished returns
 * OHGPU_ERR_INVALID (nothing is launched).  Different batches are independent.  The same holds for
 * ohgpu_flywheel_batch_run (the batch owns Burg's workspace). */
int ohgpu_src_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream);
/* The same with two events of the caller's (ohgpu_event_create) that bracket the batch's device work on `stream`: a batch that is one
 * launch of the workgroup matrix kernel carries them ON ITS DISPATCH (hipExtLaunchKernelGGL: the dispatch's own start and end
 * timestamps, no packet more in the queue -- two ohgpu_event_record calls around every launch cost a benchmark's back-to-back launches
 * 5 us each, 1.7 % of the headline's); any other batch gets them recorded in front of its first launch and behind its last.
 * ohgpu_event_elapsed_ms(start, stop) is the batch's device time either way.  A measurement's tool (bench.py's roofline): the audio
 * is ohgpu_src_batch_run's. */
int ohgpu_src_batch_run_timed(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream,
                              void* start_event, void* stop_event);

/* ---- the same batch, the next period ----
 * A caller that brings new audio every period in the same shape -- the same streams, the same tiling into messages, a whole number
 * of blocks further on in every stream -- need not plan again:
 *   - other arena positions: pass other base pointers to ohgpu_src_batch_run (the source's 16-byte aligned);
 *   - further on in the streams: ohgpu_src_batch_advance.  Every message's out_frame0 grows by blocks * block_outputs and its
 *     src_frame0 by blocks * block_inputs (ohgpu_src_batch_block), offsets and windows as they were: the plan names no absolute
 *     position (a unit is where its rows lie in the arenas, a ramp job where its frames lie in its message), and a whole number of
 *     blocks later every message has the phase it had -- so the call checks that this is such a batch and changes nothing.
 *     OHGPU_ERR_INVALID for a batch that holds a stream's first message (its window is zeros where the next period's is history),
 *     OHGPU_ERR_UNSUPPORTED for one without a block-kernel plan;
 *   - other ramp endpoints (the flags stay: which messages are ramped is the plan's shape): ohgpu_src_batch_set_ramps, one pair per
 *     message in the batch's order (those of unramped messages are not looked at).  Rewrites the ramp jobs and the generic-kernel
 *     pieces on the device and refills the multiplier planes; waits for the batch's last launch first.  Not for a batch of several
 *     layouts.
 * Anything else -- another tiling, other flags, other streams -- is another batch.  tests/test_plan_threads.py holds the plan of a
 * shifted period to the digest of the period it was made for; tests/test_gpu_parity.py runs both calls against the oracle. */
int ohgpu_src_batch_block(const ohgpu_batch* batch, uint32_t* block_outputs, uint32_t* block_inputs);
int ohgpu_src_batch_advance(ohgpu_ctx* ctx, ohgpu_batch* batch, uint64_t blocks);
int ohgpu_src_batch_set_ramps(ohgpu_ctx* ctx, ohgpu_batch* batch, const uint16_t* ramp_start, const uint16_t* ramp_end, size_t n);

/* Host-buffer convenience (a live pipeline's 5 ms cadence): H2D, run, D2H, sync, as ohgpu_pcm_process_host.  dst_host bytes
 * that no message covers are preserved.  src_host need only hold each message's WINDOW (ohgpu_src_msg_desc: src_frame0 /
 * src_frames), not the stream's history: host/SampleRateConverter.cpp packs the windows of a period's messages back to back. */
int ohgpu_src_process_host(ohgpu_ctx* ctx, const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t n,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes);

/* What the *_process_host calls of this context have moved so far: their number, how many of them were ohgpu_src_process_host,
 * and the bytes copied to and from the device for audio (descriptors and plans not counted).  A driver's handle on "one call per
 * filter per period, windows only" (tests/cpp/test_host.cpp, bench.py's cadence.adapter). */
int ohgpu_host_transfer_stats(ohgpu_ctx* ctx, uint64_t* calls, uint64_t* src_calls, uint64_t* h2d_bytes, uint64_t* d2h_bytes);

/* How a resampler batch was planned: output frames handled by the block kernel, and the number of message pieces
 * (block-unaligned heads/tails, unsupported layouts) left to the generic kernel. */
int ohgpu_src_batch_plan(const ohgpu_batch* batch, uint64_t* block_kernel_out_frames, uint64_t* generic_pieces);

/* ... and how the block kernel's share was cut: its work units, and how many of them are "long" (every row several
 * consecutive blocks of its stream: the unit schedule of large batches, ohpipeline_amd/csrc/src_plan.cpp).  Both 0 when the
 * batch runs on the generic kernel alone. */
int ohgpu_src_batch_units(const ohgpu_batch* batch, uint64_t* units, uint64_t* long_units);

/* Planning is host work, done per batch: the descriptors are independent per message and per stream, so the library checks them
 * and cuts the streams into work units on several threads (up to 16, by the size of the batch).  `threads` caps that number for
 * every later ohgpu_*_batch_create of the process (1: the calling thread alone; 0: no cap, the default).  The plan does not depend
 * on it. */
int ohgpu_set_plan_threads(int threads);

/* The plan ohgpu_src_batch_create would make for these messages, as a 64-bit hash of its unit list, ramp jobs and generic-kernel
 * pieces -- computed on the host alone (no context, no device): what tests/test_plan_threads.py compares across thread counts.
 * coef_q28 (L * taps_per_phase of them) describes the filter as ohgpu_src_create would -- the half-band form, the matrix kernels'
 * tables, the gain bound -- so that the plan is the one a real batch gets; NULL stands for a polyphase filter of sane gain with
 * no special structure.  num_cus is the device's CU count (the long-unit schedule aims at one long unit per wave): 0 = 256.
 * `kernel` is set to the newest kernel the plan serves: 3 / 1 / 0 for src_mfma_wg_kernel / src_lean_kernel / round 1's fallback or
 * none (2 = the retired unit-per-wave matrix kernel, legacy builds only); which one runs is the run-time variant's choice among those. */
int ohgpu_src_plan_digest(uint32_t L, uint32_t M, uint32_t taps_per_phase, const ohgpu_src_msg_desc* descs, size_t n,
                          uint64_t src_arena_bytes, uint64_t dst_arena_bytes, int kernel_variant,
                          const int32_t* coef_q28, int num_cus,
                          uint64_t* digest, uint64_t* units, uint64_t* generic_pieces, int* kernel);

/* Which kernel ohgpu_src_batch_run launches for the batch's whole phase-aligned blocks under the context's current kernel
 * variant: "src_mfma_wg_kernel", "src_lean_kernel", "src_block_kernel" (round 1's: the fallback for a filter whose phase sums reach
 * 2^29, beyond the lean kernel's exact rounding -- five stereo layouts) or "src_kernel_v1" (the generic one alone; a legacy build
 * also knows "src_mfma_kernel"); a batch of several layouts names its parts' kernels, comma separated.  A measurement's label (bench.py), nothing the
 * data path depends on.  The name is written to out[0, cap) NUL-terminated (truncated if it does not fit). */
int ohgpu_src_batch_kernel_name(ohgpu_ctx* ctx, const ohgpu_batch* batch, char* out, size_t cap);

/* How many workgroups of the batch's resampler kernel the device keeps on a CU at once (hipOccupancyMaxActiveBlocksPerMultiprocessor
 * of the instantiation the batch runs, with its LDS), how many its geometry was laid out for, and its LDS bytes per workgroup.  The
 * workgroup matrix kernel is sized to the LDS: three workgroups per CU at 50.7-52.0 KB each (csrc/src_mfma_wg_kernel.hip WgGeom);
 * one granule more and the device would keep two, a third of the throughput gone without an error anywhere -- tests/test_gpu_parity.py
 * holds every layout to its design.  OHGPU_ERR_UNSUPPORTED for a batch that runs on another kernel; nothing is launched.  (A
 * diagnostic like ohgpu_measure_shader_clock: no reference interface behind it.) */
int ohgpu_src_batch_occupancy(ohgpu_ctx* ctx, const ohgpu_batch* batch, int* workgroups_per_cu, int* designed_for, uint32_t* lds_bytes);

/* The shader clock the device holds right now, in MHz: a short kernel on every CU (about 0.2 ms of dependent vector work,
 * queued on `stream` like any launch) compares the shader cycle counter (s_memtime) with the constant 100 MHz reference counter
 * (s_memrealtime) and the call waits for it.  For a benchmark to report next to a kernel time, so that a slow box can be told
 * from a slow build; it touches no audio buffer. */
int ohgpu_measure_shader_clock(ohgpu_ctx* ctx, void* stream, double* mhz);

/* How many device allocations (hipMalloc) the context has made for batches' descriptors and plan arrays since ohgpu_init.  A
 * destroyed batch's blocks stay with the context and serve the next batch of that size, so a caller in a steady state -- the
 * StarvationRamper's rescue creates three batches per starving period (reference: StarvationRamper.cpp:560-620 does that work on
 * the CPU, inside the period) -- sees this number stop growing; a test's handle on "nothing is allocated in the period". */
int ohgpu_device_allocations(ohgpu_ctx* ctx, uint64_t* count);

/* Kernel selection for A/B measurement and tests: 0 = default/best; 1 = baseline "v1" kernels (a resampled batch must be CREATED under
 * 1 to run under it: only then does it keep the generic kernel's per-message form); 3 = the default kernels with the resampler's
 * long-row unit schedule forced onto batches of any size (a resampled batch created while 3 is set cuts every run of plain units
 * into rows of three blocks -- what only a batch of thousands of units gets otherwise -- so that tests reach that path with small
 * inputs; results are identical); 4 = round 2's fp64 "lean" block kernel where round 4's matrix-pipe kernel would run, for A/B.
 * 2 and 5 select round 1's block kernel and round 4's unit-per-wave matrix kernel, which are RETIRED as selectable kernels
 * (round 5): only a legacy build takes the two values (OHGPU_LEGACY=1 python ohpipeline_amd/build.py); in the shipped library both
 * behave as 4.  (Round 1's kernel stays in the shipped library for five stereo layouts as the fallback for filters beyond the
 * lean kernel's rounding bound, whatever the variant.)  Variants 2..5 shape the plan of batches created while they are set; the variant in force when a batch is RUN
 * chooses among the kernels its plan serves (ohgpu_src_batch_kernel_name says which).  A batch planned for the workgroup kernel
 * alone (six and eight channels, the layouts only it has) is REFUSED under any other variant (OHGPU_ERR_UNSUPPORTED, nothing is
 * launched): create it under the variant it is to run under. */
int ohgpu_set_kernel_variant(ohgpu_ctx* ctx, int variant);

#ifdef __cplusplus
}
#endif
#endif /* OHGPU_H */
