/*
 * ohp_songcast.c -- CPU restatement of the Songcast sender data path.  TEST INFRASTRUCTURE ONLY (see ohp_songcast.h).
 */
#include "ohp_songcast.h"
#include <string.h>

/* WriterBinary (ohNet Stream.h): big-endian scalars appended to a bounded buffer */
typedef struct {
    uint8_t* p;
    uint32_t bytes, capacity;
    int overflow;
} wr;

static void wr_u8(wr* w, uint32_t v)
{
    if (w->bytes + 1 > w->capacity) { w->overflow = 1; return; }
    w->p[w->bytes++] = (uint8_t)v;
}
static void wr_be(wr* w, uint64_t v, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) wr_u8(w, (uint32_t)(v >> (8 * (n - 1 - i))) & 0xffu);
}
static void wr_bytes(wr* w, const uint8_t* src, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) wr_u8(w, src[i]);
}

int ohp_ohm_stream_header(uint8_t* buf, uint32_t capacity, uint64_t samples_total, uint32_t sample_rate, uint32_t bit_rate,
                          int32_t volume_offset, uint32_t bit_depth, uint32_t channels, const uint8_t* codec, uint32_t codec_bytes)
{                                                                     /* OhmMsg.cpp:225-241 */
    wr w = { buf, 0, capacity, 0 };
    if (codec_bytes > OHP_OHM_MAX_CODEC_BYTES) return OHP_ERR_ASSERT;
    wr_be(&w, samples_total, 8);
    wr_be(&w, sample_rate, 4);
    wr_be(&w, bit_rate, 4);
    wr_be(&w, (uint16_t)(int16_t)volume_offset, 2);
    wr_u8(&w, bit_depth);
    wr_u8(&w, channels);
    wr_u8(&w, 0);                                                     /* kReserved */
    wr_u8(&w, codec_bytes);
    if (codec_bytes > 0) wr_bytes(&w, codec, codec_bytes);
    return w.overflow ? OHP_ERR_ASSERT : (int)w.bytes;
}

int ohp_ohm_audio_frame(uint8_t* out, uint32_t capacity, uint32_t flags_in, uint32_t samples, uint32_t frame,
                        uint32_t network_timestamp, uint32_t media_latency, uint64_t sample_start,
                        const uint8_t* stream_header, uint32_t stream_header_bytes, const uint8_t* audio, uint32_t audio_bytes)
{
    /* ReinitialiseFields, OhmMsg.cpp:203-223: the unified buffer holds [pad | stream header | audio] with the audio at
       kStreamHeaderBytes; Serialise (:363-413) prepends the 8 + 28 bytes in front of the stream header, so the stream
       header has to leave room for them */
    if (stream_header_bytes + OHP_OHM_HEADER_BYTES + 28u > OHP_OHM_STREAM_HEADER_BYTES) return OHP_ERR_ASSERT;
    if (audio_bytes > OHP_OHM_MAX_SAMPLE_BYTES) return OHP_ERR_ASSERT;
    const uint32_t per_frame = 28;                                    /* kPerFrameBytes, :368 */
    const uint32_t additional = per_frame + stream_header_bytes + audio_bytes;        /* :371 */
    wr w = { out, 0, capacity, 0 };
    /* OhmHeader(kMsgTypeAudio, additional) then Externalise, Ohm.cpp:16-20, 44-52 */
    wr_bytes(&w, (const uint8_t*)"Ohm ", 4);
    wr_u8(&w, 1);                                                     /* kMajor */
    wr_u8(&w, OHP_OHM_MSG_TYPE_AUDIO);
    wr_be(&w, OHP_OHM_HEADER_BYTES + additional, 2);
    uint32_t flags = 0;                                               /* :385-400 */
    const uint32_t timestamped = (flags_in & OHP_OHM_FLAG_TIMESTAMPED) ? 1u : 0u;
    if (flags_in & OHP_OHM_FLAG_HALT) flags |= OHP_OHM_FLAG_HALT;
    if (flags_in & OHP_OHM_FLAG_LOSSLESS) flags |= OHP_OHM_FLAG_LOSSLESS;
    if (timestamped) flags |= OHP_OHM_FLAG_TIMESTAMPED;
    if (flags_in & OHP_OHM_FLAG_RESENT) flags |= OHP_OHM_FLAG_RESENT;
    if (timestamped) flags |= OHP_OHM_FLAG_TIMESTAMPED2;              /* iTimestamped2 = iTimestamped, :211 */
    wr_u8(&w, OHP_OHM_AUDIO_HEADER_BYTES);
    wr_u8(&w, flags);
    wr_be(&w, samples, 2);
    wr_be(&w, frame, 4);
    wr_be(&w, network_timestamp, 4);
    wr_be(&w, media_latency, 4);
    wr_be(&w, 0, 4);                                                  /* iMediaTimestamp = 0, :217 */
    wr_be(&w, sample_start, 8);
    wr_bytes(&w, stream_header, stream_header_bytes);
    wr_bytes(&w, audio, audio_bytes);
    return w.overflow ? OHP_ERR_ASSERT : (int)w.bytes;
}

/* ReaderBinary over a bounded buffer */
typedef struct {
    const uint8_t* p;
    uint32_t pos, bytes;
    int underflow;
} rd;

static uint64_t rd_be(rd* r, uint32_t n)
{
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (r->pos >= r->bytes) { r->underflow = 1; return 0; }
        v = (v << 8) | r->p[r->pos++];
    }
    return v;
}

int ohp_ohm_audio_parse(const uint8_t* datagram, uint32_t bytes, ohp_ohm_audio* out)
{
    rd r = { datagram, 0, bytes, 0 };
    memset(out, 0, sizeof(*out));
    /* OhmHeader::Internalise, Ohm.cpp:22-42 */
    if (bytes < OHP_OHM_HEADER_BYTES || memcmp(datagram, "Ohm ", 4) != 0) return OHP_ERR_ASSERT;
    r.pos = 4;
    if (rd_be(&r, 1) != 1) return OHP_ERR_ASSERT;
    out->msg_type = (uint32_t)rd_be(&r, 1);
    if (out->msg_type > 7 && out->msg_type != 255) return OHP_ERR_ASSERT;
    const uint32_t total = (uint32_t)rd_be(&r, 2);
    if (total < OHP_OHM_HEADER_BYTES) return OHP_ERR_ASSERT;
    out->msg_bytes = total - OHP_OHM_HEADER_BYTES;
    if (out->msg_type != OHP_OHM_MSG_TYPE_AUDIO) return OHP_ERR_ASSERT;      /* OhmMsg.cpp:103-104 */
    /* OhmMsgAudio::Create(IReader&, const OhmHeader&), OhmMsg.cpp:100-174: 50 header bytes, the last one the codec length */
    if (bytes < OHP_OHM_HEADER_BYTES + OHP_OHM_AUDIO_HEADER_BYTES) return OHP_ERR_ASSERT;
    if (rd_be(&r, 1) != OHP_OHM_AUDIO_HEADER_BYTES) return OHP_ERR_ASSERT;   /* :131-132 */
    const uint32_t flags = (uint32_t)rd_be(&r, 1);
    out->halt = (flags & OHP_OHM_FLAG_HALT) != 0;
    out->lossless = (flags & OHP_OHM_FLAG_LOSSLESS) != 0;
    out->timestamped = (flags & OHP_OHM_FLAG_TIMESTAMPED) != 0;
    out->timestamped2 = (flags & OHP_OHM_FLAG_TIMESTAMPED2) != 0;
    out->resent = (flags & OHP_OHM_FLAG_RESENT) != 0;
    out->samples = (uint32_t)rd_be(&r, 2);
    out->frame = (uint32_t)rd_be(&r, 4);
    out->network_timestamp = (uint32_t)rd_be(&r, 4);
    out->media_latency = (uint32_t)rd_be(&r, 4);
    out->media_timestamp = (uint32_t)rd_be(&r, 4);
    out->sample_start = rd_be(&r, 8);
    out->samples_total = rd_be(&r, 8);
    out->sample_rate = (uint32_t)rd_be(&r, 4);
    out->bit_rate = (uint32_t)rd_be(&r, 4);
    out->volume_offset = (int16_t)(uint16_t)rd_be(&r, 2);
    out->bit_depth = (uint32_t)rd_be(&r, 1);
    out->channels = (uint32_t)rd_be(&r, 1);
    if (rd_be(&r, 1) != 0) return OHP_ERR_ASSERT;                            /* reserved, :166-167 */
    out->codec_bytes = (uint32_t)rd_be(&r, 1);                               /* the header's last byte, :115 */
    if (out->codec_bytes > OHP_OHM_MAX_CODEC_BYTES) return OHP_ERR_ASSERT;   /* ReadReplace into Bws<kMaxCodecBytes>, :120 */
    for (uint32_t i = 0; i < out->codec_bytes; i++) out->codec[i] = (uint8_t)rd_be(&r, 1);
    if (out->msg_bytes < OHP_OHM_AUDIO_HEADER_BYTES + out->codec_bytes) return OHP_ERR_ASSERT;
    out->audio_bytes = out->msg_bytes - OHP_OHM_AUDIO_HEADER_BYTES - out->codec_bytes;     /* :169 */
    out->audio_offset = r.pos;
    if (r.underflow || out->audio_offset + out->audio_bytes > bytes) return OHP_ERR_ASSERT;
    if (out->audio_bytes > OHP_OHM_MAX_SAMPLE_BYTES) return OHP_ERR_ASSERT;
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * Sender::ProcessAudio / SendPendingAudio (Sender.cpp:277-321)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    ohp_sender_fragment* fragments;
    uint32_t fragment_capacity, n_fragments;
    ohp_sender_packet* packets;
    uint32_t packet_capacity, n_packets;
    ohp_msg_audio pending[64];
    uint32_t pending_msg[64];
    uint32_t n_pending;
} packetiser;

static int send_pending(packetiser* s)
{                                                                     /* Sender.cpp:307-321 */
    if (s->n_packets >= s->packet_capacity) return OHP_ERR_ASSERT;
    ohp_sender_packet* pk = &s->packets[s->n_packets++];
    pk->first_fragment = s->n_fragments;
    pk->n_fragments = s->n_pending;
    for (uint32_t i = 0; i < s->n_pending; i++) {
        if (s->n_fragments >= s->fragment_capacity) return OHP_ERR_ASSERT;
        ohp_sender_fragment* f = &s->fragments[s->n_fragments++];
        f->msg = s->pending_msg[i];
        const int err = ohp_create_playable(&s->pending[i], &f->playable);         /* PlayableCreator, :494-510 */
        if (err < 0) return err;
    }
    s->n_pending = 0;
    return OHP_OK;
}

static int push_pending(packetiser* s, const ohp_msg_audio* m, uint32_t msg)
{
    if (s->n_pending >= 64) return OHP_ERR_ASSERT;
    s->pending[s->n_pending] = *m;
    s->pending_msg[s->n_pending] = msg;
    s->n_pending++;
    return OHP_OK;
}

int ohp_sender_packetise(const ohp_msg_audio* msgs, uint32_t n_msgs, int flush,
                         ohp_sender_fragment* fragments, uint32_t fragment_capacity, uint32_t* n_fragments,
                         ohp_sender_packet* packets, uint32_t packet_capacity, uint32_t* n_packets)
{
    const uint32_t kPacketJiffies = OHP_JIFFIES_PER_MS * OHP_SONGCAST_PACKET_MS;     /* Sender.h:36 */
    packetiser s;
    memset(&s, 0, sizeof(s));
    s.fragments = fragments; s.fragment_capacity = fragment_capacity;
    s.packets = packets; s.packet_capacity = packet_capacity;
    int err;
    for (uint32_t i = 0; i < n_msgs; i++) {                           /* ProcessAudio, :277-305 */
        uint32_t jiffies = 0;
        for (uint32_t k = 0; k < s.n_pending; k++) jiffies += s.pending[k].size_jiffies;
        uint32_t new_jiffies = jiffies + msgs[i].size_jiffies;
        if (new_jiffies < kPacketJiffies) {
            if ((err = push_pending(&s, &msgs[i], i)) < 0) return err;
            continue;
        }
        ohp_msg_audio msg = msgs[i];
        ohp_msg_audio remaining;
        int has_remaining;
        memset(&remaining, 0, sizeof(remaining));
        do {
            has_remaining = (new_jiffies != kPacketJiffies);
            if (has_remaining) {
                if ((err = ohp_msg_audio_split(&msg, kPacketJiffies - jiffies, &remaining)) < 0) return err;
            }
            if ((err = push_pending(&s, &msg, i)) < 0) return err;
            if ((err = send_pending(&s)) < 0) return err;
            msg = remaining;
            jiffies = 0;
            new_jiffies = has_remaining ? remaining.size_jiffies : 0;
        } while (has_remaining && new_jiffies >= kPacketJiffies);
        if (has_remaining) {
            if ((err = push_pending(&s, &remaining, i)) < 0) return err;
        }
    }
    if (flush) {
        if ((err = send_pending(&s)) < 0) return err;                 /* ProcessMsg(MsgQuit*), :271-275 */
    }
    *n_fragments = s.n_fragments;
    *n_packets = s.n_packets;
    return OHP_OK;
}

/* ------------------------------------------------------------------------------------------
 * OhmSenderDriver (OhmSender.cpp:299-344, 418-488, 544-556)
 * ---------------------------------------------------------------------------------------- */
void ohp_ohm_driver_init(ohp_ohm_driver* d, uint32_t latency_ms)
{
    memset(d, 0, sizeof(*d));
    d->first_frame = 1;
    d->latency_ms = latency_ms;                                       /* SetLatency: iLatencyOhm stays 0 until a format is set */
    d->send = 1;                                                      /* SetEnabled(true) + SetActive(true), :490-525 */
}

void ohp_ohm_driver_set_track_position(ohp_ohm_driver* d, uint64_t samples_total, uint64_t sample_start)
{
    d->samples_total = samples_total;
    d->sample_start = sample_start;
}

int ohp_ohm_driver_set_audio_format(ohp_ohm_driver* d, uint32_t sample_rate, uint32_t bit_rate, uint32_t channels,
                                    uint32_t bit_depth, uint32_t lossless, const uint8_t* codec, uint32_t codec_bytes,
                                    uint64_t sample_start)
{                                                                     /* :325-344 */
    uint32_t ticks = 0;
    const int err = ohp_jiffies_to_songcast_time(OHP_JIFFIES_PER_SEC, sample_rate, &ticks);   /* = SongcastTicksPerSecond */
    if (err < 0) return err;
    d->sample_rate = sample_rate;
    d->timestamp_multiplier = ticks;
    d->latency_ohm = (uint32_t)(d->latency_ms * d->timestamp_multiplier) / 1000u;             /* UpdateLatencyOhm, :320-323 (TUint arithmetic) */
    d->bytes_per_sample = channels * bit_depth / 8;
    d->lossless = lossless;
    d->sample_start = sample_start;
    const int n = ohp_ohm_stream_header(d->stream_header, sizeof(d->stream_header), d->samples_total, sample_rate, bit_rate,
                                        0, bit_depth, channels, codec, codec_bytes);
    if (n < 0) return n;
    d->stream_header_bytes = (uint32_t)n;
    return OHP_OK;
}

int ohp_ohm_driver_send_audio(ohp_ohm_driver* d, const uint8_t* audio, uint32_t audio_bytes, int halt,
                              uint8_t* out, uint32_t capacity)
{                                                                     /* :418-480 */
    const uint32_t samples = d->bytes_per_sample == 0 ? 0 : audio_bytes / d->bytes_per_sample;
    if (!d->send) {
        d->sample_start += samples;
        return 0;
    }
    if (d->sample_rate == 0 || (samples == 0 && !halt)) return 0;
    if (d->first_frame) d->first_frame = 0;                           /* no timestamper: never timestamped */
    uint32_t flags = 0;
    if (halt) flags |= OHP_OHM_FLAG_HALT;
    if (d->lossless) flags |= OHP_OHM_FLAG_LOSSLESS;
    const int n = ohp_ohm_audio_frame(out, capacity, flags, samples, d->frame, 0, d->latency_ohm, d->sample_start,
                                      d->stream_header, d->stream_header_bytes, audio, audio_bytes);
    if (n < 0) return n;
    d->sample_start += samples;
    d->frame++;
    return n;
}

void ohp_ohm_driver_stream_interrupted(ohp_ohm_driver* d)
{
    d->frame += 250;
}
