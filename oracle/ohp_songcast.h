/*
 * ohp_songcast.h -- CPU restatement of ohPipeline's Songcast sender data path (TEST INFRASTRUCTURE ONLY; SURVEY.md 8f
 * row N3): 5 ms packetisation, the sender pack and the OHM audio frame.
 *
 * Only tests/, __graft_entry__.smoke() and bench legs named "cpu_baseline" may use this file; the product
 * (libohgpu.so, libohhost.so) never links or loads it.
 *
 * Follows OpenHome/Av/Songcast/{Ohm.cpp, OhmMsg.cpp, OhmSender.cpp, Sender.cpp} of the reference (read as text; the
 * reference cannot be compiled here: ohNet headers are absent).
 *
 * PARITY UNPINNED for the frame layout: the reference holds no test, fixture or golden datagram for OhmMsgAudio (its
 * only Songcast test, Av/Tests/TestSenderQueueMain.cpp, exercises the message queue, not the wire format).  What pins
 * this file instead: the writer (ohp_ohm_audio_frame) and the reader (ohp_ohm_audio_parse) restate two DIFFERENT
 * reference functions -- OhmMsgAudio::Serialise and OhmMsgAudio::Create(IReader&) -- and tests/test_oracle_songcast.py
 * checks that each field written is the field read back, plus a datagram assembled by hand from the field table in
 * OhmMsg.cpp:391-400 / :225-241.  The audio payload (ohp_sender_pack) and the message arithmetic under the packetiser
 * (ohp_msg_audio_split, ohp_create_playable) ARE pinned, by the reference's TestMsg cases restated in
 * tests/test_oracle_reference_kats.py.
 */
#ifndef OHP_SONGCAST_H
#define OHP_SONGCAST_H

#include <stdint.h>
#include "ohp_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OHP_OHM_HEADER_BYTES         8u     /* OhmHeader::kHeaderBytes, Ohm.h:30 */
#define OHP_OHM_MSG_TYPE_AUDIO       3u     /* OhmHeader::kMsgTypeAudio, Ohm.h:36 */
#define OHP_OHM_AUDIO_HEADER_BYTES   50u    /* OhmMsgAudio::kHeaderBytes, OhmMsg.h:75 (not including the codec name) */
#define OHP_OHM_MAX_CODEC_BYTES      29u    /* OhmMsgAudio::kMaxCodecBytes, OhmMsg.h:66 */
#define OHP_OHM_MAX_SAMPLE_BYTES     5760u  /* OhmMsgAudio::kMaxSampleBytes, OhmMsg.h:65 */
#define OHP_OHM_STREAM_HEADER_BYTES  88u    /* OhmMsgAudio::kStreamHeaderBytes, OhmMsg.h:72 */
#define OHP_OHM_FLAG_HALT          0x01u    /* OhmMsg.h:67-71 */
#define OHP_OHM_FLAG_LOSSLESS      0x02u
#define OHP_OHM_FLAG_TIMESTAMPED   0x04u
#define OHP_OHM_FLAG_RESENT        0x08u
#define OHP_OHM_FLAG_TIMESTAMPED2  0x10u
#define OHP_SONGCAST_PACKET_MS       5u     /* Sender::kSongcastPacketMs, Sender.h:35 */

/* OhmMsgAudio::GetStreamHeader, OhmMsg.cpp:225-241: the part of the header that is constant over a stream.
 * Returns the number of bytes written (22 + codec_bytes), or OHP_ERR_ASSERT when it does not fit. */
int ohp_ohm_stream_header(uint8_t* buf, uint32_t capacity, uint64_t samples_total, uint32_t sample_rate, uint32_t bit_rate,
                          int32_t volume_offset, uint32_t bit_depth, uint32_t channels, const uint8_t* codec, uint32_t codec_bytes);

/* One audio datagram as OhmSenderDriver::SendAudio puts it on the wire: OhmMsgAudio::ReinitialiseFields (OhmMsg.cpp:
 * 203-223: timestamped2 = timestamped, media timestamp = 0), OhmMsgAudio::Serialise (:363-413) with OhmHeader::Externalise
 * (Ohm.cpp:44-52), then SendableBuffer (:415-419).  `flags` carries OHP_OHM_FLAG_HALT / _LOSSLESS / _TIMESTAMPED / _RESENT.
 * Returns the datagram size, or OHP_ERR_ASSERT. */
int ohp_ohm_audio_frame(uint8_t* out, uint32_t capacity, uint32_t flags, uint32_t samples, uint32_t frame,
                        uint32_t network_timestamp, uint32_t media_latency, uint64_t sample_start,
                        const uint8_t* stream_header, uint32_t stream_header_bytes, const uint8_t* audio, uint32_t audio_bytes);

typedef struct {
    uint32_t msg_type, msg_bytes;          /* OhmHeader: type, iBytes - 8 */
    uint32_t halt, lossless, timestamped, timestamped2, resent;
    uint32_t samples, frame, network_timestamp, media_latency, media_timestamp;
    uint64_t sample_start, samples_total;
    uint32_t sample_rate, bit_rate;
    int32_t  volume_offset;
    uint32_t bit_depth, channels, codec_bytes;
    uint8_t  codec[32];
    uint32_t audio_offset, audio_bytes;    /* where the payload sits in the datagram */
} ohp_ohm_audio;

/* The receiving side: OhmHeader::Internalise (Ohm.cpp:22-42) then OhmMsgAudio::Create(IReader&, const OhmHeader&)
 * (OhmMsg.cpp:100-174).  OHP_ERR_ASSERT where the reference throws OhmError or asserts. */
int ohp_ohm_audio_parse(const uint8_t* datagram, uint32_t bytes, ohp_ohm_audio* out);

/* ---- Sender::ProcessAudio / SendPendingAudio (Av/Songcast/Sender.cpp:277-321) ----
 * Feeds msgs[0..n) (as they arrive after a MsgDecodedStream) through the pending-audio logic; every SendPendingAudio
 * becomes one packet = a run of fragments, each fragment the MsgPlayable created from one pending message
 * (offset/size in bytes inside message `msg`'s own audio buffer, with the ramp and attenuation the playable carries).
 * flush != 0 appends the final SendPendingAudio of ProcessMsg(MsgQuit*) (:271-275).  Audio left pending stays unsent. */
typedef struct {
    uint32_t msg;                          /* index into msgs[] */
    ohp_playable playable;
} ohp_sender_fragment;

typedef struct {
    uint32_t first_fragment, n_fragments;
} ohp_sender_packet;

int ohp_sender_packetise(const ohp_msg_audio* msgs, uint32_t n_msgs, int flush,
                         ohp_sender_fragment* fragments, uint32_t fragment_capacity, uint32_t* n_fragments,
                         ohp_sender_packet* packets, uint32_t packet_capacity, uint32_t* n_packets);

/* ---- OhmSenderDriver::SetAudioFormat / SendAudio (Av/Songcast/OhmSender.cpp:325-344, 418-480) ----
 * The per-stream counters that end up in each frame header; no timestamper (network timestamp 0, not timestamped). */
typedef struct {
    uint32_t sample_rate, bytes_per_sample, lossless, latency_ms, latency_ohm, timestamp_multiplier;
    uint64_t sample_start, samples_total;
    uint32_t frame, first_frame, send;
    uint8_t  stream_header[OHP_OHM_STREAM_HEADER_BYTES];
    uint32_t stream_header_bytes;
} ohp_ohm_driver;

void ohp_ohm_driver_init(ohp_ohm_driver* d, uint32_t latency_ms);         /* the constructor's state (OhmSender.cpp:299-318) after
                                                                             SetLatency (:544-549), SetEnabled(true), SetActive(true) */
void ohp_ohm_driver_set_track_position(ohp_ohm_driver* d, uint64_t samples_total, uint64_t sample_start); /* :551-556 */
int  ohp_ohm_driver_set_audio_format(ohp_ohm_driver* d, uint32_t sample_rate, uint32_t bit_rate, uint32_t channels,
                                     uint32_t bit_depth, uint32_t lossless, const uint8_t* codec, uint32_t codec_bytes,
                                     uint64_t sample_start);
/* Returns the datagram size (0 when SendAudio decides there is nothing to send), or OHP_ERR_ASSERT. */
int  ohp_ohm_driver_send_audio(ohp_ohm_driver* d, const uint8_t* audio, uint32_t audio_bytes, int halt,
                               uint8_t* out, uint32_t capacity);
void ohp_ohm_driver_stream_interrupted(ohp_ohm_driver* d);               /* :482-488 */

#ifdef __cplusplus
}
#endif
#endif
