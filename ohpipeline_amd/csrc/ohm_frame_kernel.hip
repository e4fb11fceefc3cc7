// ohm_frame_kernel.hip -- Songcast sender frames: the C ABI's ohgpu_ohm_* entry points.
//
// A frame = header + audio.  The audio of every fragment is an ordinary message for the kernels that already exist:
//   mono / stereo streams: one PCM message (ramp, attenuation, 32 -> 24 bit) written straight into the frame -- the
//     Sender's pack of such a stream IS the depth conversion (Sender.cpp:351-377 keeps min(bytes, 3) leading bytes);
//   wider streams: the Sender pack's channel select (fmt_line_kernel), after a PCM pass into a scratch arena when the
//     fragment is ramped, attenuated or silent (what MsgPlayable::Read would have applied first, Msg.cpp:2753-2786).
// What is new here is the header: 36 per-frame bytes (OhmHeader + the per-frame part of OhmMsgAudio::Serialise,
// OhmMsg.cpp:363-413) and the per-stream 22 + codec bytes (GetStreamHeader, :225-241).  A mono / stereo frame's header is
// assembled on the host when the batch is created and travels as the PREFIX of the frame's first audio message: the wave
// that writes that audio writes the header in front of it (pcm_line_kernel), one launch for the whole batch.  Every other
// frame's header (wider streams, frames without audio) is written by ohm_header_kernel, 16 lanes per frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

static constexpr uint32_t kPerFrameHeader = 36;        // OhmHeader::kHeaderBytes (8) + kPerFrameBytes (28), OhmMsg.cpp:368
static constexpr uint32_t kStreamFixed = 22;           // GetStreamHeader without the codec name
static constexpr uint32_t kLanesPerFrame = 16;
static constexpr uint32_t kFramesPerGroup = 4;         // frames each 16-lane group writes, their loads issued together

// 16 lanes per frame, one (unaligned) dword store each per 64 header bytes: the record already holds the 36 per-frame
// bytes in wire order and the stream record the rest, so this is a two-source copy.  A header that is not a whole
// number of dwords ends in byte stores.  The kernel is latency bound (record -> stream record -> store), hence the
// header size travelling in the frame record and several frames in flight per lane.
__global__ void __launch_bounds__(256)
ohm_header_kernel(const OhmFrameRec* __restrict__ frames, uint32_t n_frames, const uint8_t* __restrict__ streams,
                  uint8_t* __restrict__ dst)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t group = t / kLanesPerFrame, lane = t % kLanesPerFrame;
    uint32_t v[kFramesPerGroup][2], hb[kFramesPerGroup];
    uint64_t off[kFramesPerGroup];
#pragma unroll
    for (uint32_t k = 0; k < kFramesPerGroup; k++) {
        const uint32_t f = group * kFramesPerGroup + k;
        hb[k] = 0;
        if (f >= n_frames) continue;
        const OhmFrameRec& r = frames[f];
        const uint32_t sh = r.stream_and_bytes;
        hb[k] = sh >> 24;
        off[k] = r.dst_off;
        const uint8_t* s = streams + (size_t)(sh & 0xffffffu) * 64;
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t w = lane + i * kLanesPerFrame;
            v[k][i] = 0;
            if (w * 4 < hb[k]) v[k][i] = (w < kPerFrameHeader / 4) ? r.w[w] : *(const uint32_t*)(s + (w * 4 - kPerFrameHeader));
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < kFramesPerGroup; k++) {
        uint8_t* out = dst + off[k];
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            const uint32_t w = lane + i * kLanesPerFrame;
            if (w * 4 + 4 <= hb[k]) {
                __builtin_memcpy(out + w * 4, &v[k][i], 4);
            } else if (w * 4 < hb[k]) {
                for (uint32_t b = w * 4; b < hb[k]; b++) out[b] = (uint8_t)(v[k][i] >> (8u * (b & 3u)));
            }
        }
    }
}

// Ramped / attenuated fragments of streams with more than two channels: what MsgPlayablePcm::ReadBlock (attenuation, then
// RampApplicator; Msg.cpp:2736-2786) and Sender::DoProcessFragment (two channels, <= 3 leading bytes; Sender.cpp:351-377)
// do to the two channels that go on the wire, in one pass: a wave per fragment, a lane per wire subsample.
__global__ void __launch_bounds__(256)
ohm_select_ramp_kernel(const OhmSelRec* __restrict__ recs, uint32_t n_recs, const uint16_t* __restrict__ ramp_table,
                       const uint8_t* __restrict__ src, uint64_t src_arena_bytes, uint8_t* __restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63, waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n_recs; i += waves) {
        const OhmSelRec r = recs[i];
        const uint32_t sb = r.sb, wb = sb < 3 ? sb : 3;
        const int32_t total = (int32_t)r.ramp_start - (int32_t)r.ramp_end;
        for (uint32_t q = lane; q < r.n_frames * 2; q += 64) {
            const uint32_t f = q >> 1, c = r.first_ch + (q & 1);
            const uint64_t at = r.src_off + ((uint64_t)f * r.channels + c) * sb;
            const uint8_t* p = src + at;
            uint32_t w = 0;                                              // left-justified, most significant byte first
            if (at + 4 <= src_arena_bytes) {                             // one (unaligned) dword load; the bytes beyond the subsample are dropped
                uint32_t v;
                __builtin_memcpy(&v, p, 4);
                w = r.little ? (v << (32 - 8 * sb)) : (__builtin_bswap32(v) & (0xffffffffu << (32 - 8 * sb)));
            } else {
                for (uint32_t k = 0; k < sb; k++) w |= (uint32_t)p[r.little ? sb - 1 - k : k] << (24 - 8 * k);
            }
            if (r.attenuation != OHGPU_UNITY_ATTENUATION) w = attenuate_word(w, r.attenuation);
            if (r.flags & OHGPU_FLAG_RAMP)
                w = ramp_word(w, ramp_table[ramp_index_magic(r.ramp_start, total, f, r.n_frames, r.m_n1, r.s_n1)], sb, r.channels, c);
            uint8_t* o = dst + r.dst_off + (size_t)q * wb;
            for (uint32_t k = 0; k < wb; k++) o[k] = (uint8_t)(w >> (24 - 8 * k));
        }
    }
}

static inline void put_be(uint8_t* p, uint64_t v, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * (n - 1 - i)));
}

void free_ohm(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    OhmPlan& p = b->ohm;
    ohgpu_batch_destroy(ctx, p.direct);
    ohgpu_batch_destroy(ctx, p.select);
    ohgpu_batch_destroy(ctx, p.stage);
    ohgpu_batch_destroy(ctx, p.select_staged);
    if (p.d_scratch) hipFree(p.d_scratch);
    if (p.d_selr) hipFree(p.d_selr);
    if (p.d_frames) hipFree(p.d_frames);
    if (p.d_streams) hipFree(p.d_streams);
    p = OhmPlan();
}

static int check_stream(const ohgpu_ohm_stream& s, size_t i)
{
    if (s.src_channels < 1 || s.src_channels > 10)
        return set_error(OHGPU_ERR_INVALID, "ohm stream %zu: channels %u outside 1..10", i, s.src_channels);
    if (!(s.src_bits == 8 || s.src_bits == 16 || s.src_bits == 24 || s.src_bits == 32))
        return set_error(OHGPU_ERR_INVALID, "ohm stream %zu: bit depth %u (must be 8/16/24/32)", i, s.src_bits);
    if (!(s.src_endian == 0 || s.src_endian == OHGPU_ENDIAN_BIG || s.src_endian == OHGPU_ENDIAN_LITTLE))
        return set_error(OHGPU_ERR_INVALID, "ohm stream %zu: endian %u", i, s.src_endian);
    if (s.codec_bytes > OHGPU_OHM_MAX_CODEC_BYTES)                     // Bws<kMaxCodecBytes>, OhmMsg.h:66
        return set_error(OHGPU_ERR_INVALID, "ohm stream %zu: codec name of %u bytes (limit 29)", i, s.codec_bytes);
    return OHGPU_OK;
}

static inline uint32_t wire_channels(const ohgpu_ohm_stream& s) { return s.src_channels < 2 ? s.src_channels : 2; }
static inline uint32_t wire_bytes(const ohgpu_ohm_stream& s) { return s.src_bits / 8 < 3 ? s.src_bits / 8 : 3; }

}  // namespace ohgpu

using namespace ohgpu;

extern "C" {

int ohgpu_ohm_frame_layout(const ohgpu_ohm_stream* stream, uint32_t samples, uint32_t* header_bytes, uint32_t* frame_bytes)
{
    if (!stream) return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_frame_layout: null stream");
    const int err = check_stream(*stream, 0);
    if (err != OHGPU_OK) return err;
    const uint32_t hb = kPerFrameHeader + kStreamFixed + stream->codec_bytes;
    const uint64_t audio = (uint64_t)samples * wire_channels(*stream) * wire_bytes(*stream);
    if (audio > OHGPU_OHM_MAX_AUDIO_BYTES)
        return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_frame_layout: %u samples need %llu audio bytes (OhmMsgAudio::kMaxSampleBytes is 5760)",
                         samples, (unsigned long long)audio);
    if (header_bytes) *header_bytes = hb;
    if (frame_bytes) *frame_bytes = hb + (uint32_t)audio;
    return OHGPU_OK;
}

int ohgpu_ohm_batch_create(ohgpu_ctx* ctx, const ohgpu_ohm_stream* streams, size_t n_streams,
                           const ohgpu_ohm_frame_desc* frames, size_t n_frames,
                           const ohgpu_ohm_fragment* fragments, size_t n_fragments,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** out)
{
    if (!ctx) return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_create: null context");
    OHGPU_HIP_TRY(hipSetDevice(ctx->device));
    if (!out || (n_streams && !streams) || (n_frames && !frames) || (n_fragments && !fragments))
        return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_create: null argument");
    *out = nullptr;
    if (n_frames > 0x03ffffffull || n_fragments > 0xffffffffull || n_streams > 0x00ffffffull)
        return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_create: too many descriptors");
    for (size_t i = 0; i < n_streams; i++) {
        const int err = check_stream(streams[i], i);
        if (err != OHGPU_OK) return err;
    }

    std::vector<OhmFrameRec> recs(n_frames);
    std::vector<uint8_t> stream_recs(n_streams * 64, 0);
    for (size_t i = 0; i < n_streams; i++) {                          // OhmMsgAudio::GetStreamHeader, OhmMsg.cpp:225-241
        const ohgpu_ohm_stream& s = streams[i];
        uint8_t* p = &stream_recs[i * 64];
        put_be(p, s.samples_total, 8);
        put_be(p + 8, s.sample_rate, 4);
        put_be(p + 12, s.bit_rate, 4);
        put_be(p + 16, (uint16_t)s.volume_offset, 2);
        p[18] = (uint8_t)(s.src_bits < 24 ? s.src_bits : 24);          // Sender.cpp:226
        p[19] = (uint8_t)wire_channels(s);                             // Sender.cpp:235
        p[20] = 0;                                                     // kReserved
        p[21] = s.codec_bytes;
        memcpy(p + 22, s.codec, s.codec_bytes);
        p[63] = (uint8_t)(kStreamFixed + s.codec_bytes);
    }

    std::vector<ohgpu_msg_desc> direct, stage;
    std::vector<MsgPrefix> direct_prefix;                              // per direct message: the frame header it carries (bytes == 0: none)
    std::vector<uint8_t> blob;                                         // those headers, each padded to whole dwords
    std::vector<uint8_t> folded(n_frames, 0);
    std::vector<ohgpu_fmt_desc> select, select_staged;
    std::vector<OhmSelRec> selr;
    uint64_t scratch_bytes = 0, in_frames = 0, src_touched = 0, dst_written = 0;
    for (size_t f = 0; f < n_frames; f++) {
        const ohgpu_ohm_frame_desc& fr = frames[f];
        if (fr.stream >= n_streams) return set_error(OHGPU_ERR_INVALID, "ohm frame %zu: stream %u of %zu", f, fr.stream, n_streams);
        if (fr.flags & ~(OHGPU_OHM_FLAG_HALT | OHGPU_OHM_FLAG_LOSSLESS | OHGPU_OHM_FLAG_TIMESTAMPED | OHGPU_OHM_FLAG_RESENT))
            return set_error(OHGPU_ERR_INVALID, "ohm frame %zu: unknown flag bits 0x%x", f, fr.flags);
        if ((uint64_t)fr.first_fragment + fr.n_fragments > n_fragments)
            return set_error(OHGPU_ERR_INVALID, "ohm frame %zu: fragments [%u, +%u) of %zu", f, fr.first_fragment, fr.n_fragments, n_fragments);
        const ohgpu_ohm_stream& s = streams[fr.stream];
        const uint32_t ch = s.src_channels, wch = wire_channels(s), wb = wire_bytes(s);
        const uint32_t header_bytes = kPerFrameHeader + kStreamFixed + s.codec_bytes;
        uint64_t samples = 0;
        for (uint32_t g = 0; g < fr.n_fragments; g++) samples += fragments[fr.first_fragment + g].n_frames;
        const uint64_t audio_bytes = samples * wch * wb;
        if (audio_bytes > OHGPU_OHM_MAX_AUDIO_BYTES)                   // ASSERT(iAudioBuf->BytesRemaining() >= totalBytesToCopy), Sender.cpp:364
            return set_error(OHGPU_ERR_INVALID, "ohm frame %zu: %llu audio bytes (OhmMsgAudio::kMaxSampleBytes is 5760)", f, (unsigned long long)audio_bytes);
        const uint64_t frame_bytes = header_bytes + audio_bytes;
        if (fr.dst_offset > dst_arena_bytes || frame_bytes > dst_arena_bytes - fr.dst_offset)
            return set_error(OHGPU_ERR_BOUNDS, "ohm frame %zu: writes [%llu, +%llu) beyond the %llu-byte destination arena", f,
                             (unsigned long long)fr.dst_offset, (unsigned long long)frame_bytes, (unsigned long long)dst_arena_bytes);
        // ---- header: OhmHeader::Externalise (Ohm.cpp:44-52) then OhmMsgAudio::Serialise (OhmMsg.cpp:385-411) ----
        OhmFrameRec& r = recs[f];
        uint32_t flags = fr.flags;
        if (flags & OHGPU_OHM_FLAG_TIMESTAMPED) flags |= 0x10u;        // kFlagTimestamped2: iTimestamped2 = iTimestamped, OhmMsg.cpp:211
        r.dst_off = fr.dst_offset;
        r.stream_and_bytes = fr.stream | (header_bytes << 24);
        uint8_t* h = (uint8_t*)r.w;                                    // the 36 bytes in wire order
        memcpy(h, "Ohm ", 4);
        h[4] = 1;                                                      // kMajor
        h[5] = 3;                                                      // kMsgTypeAudio
        put_be(h + 6, frame_bytes, 2);                                 // iBytes = kHeaderBytes + the rest
        h[8] = 50;                                                     // OhmMsgAudio::kHeaderBytes
        h[9] = (uint8_t)flags;
        put_be(h + 10, samples, 2);
        put_be(h + 12, fr.frame, 4);
        put_be(h + 16, fr.network_timestamp, 4);
        put_be(h + 20, fr.media_latency, 4);
        put_be(h + 24, fr.media_timestamp, 4);
        put_be(h + 28, fr.sample_start, 8);
        // ---- audio: one message per fragment ----
        uint64_t at = fr.dst_offset + header_bytes;
        for (uint32_t g = 0; g < fr.n_fragments; g++) {
            const ohgpu_ohm_fragment& fg = fragments[fr.first_fragment + g];
            const size_t gi = (size_t)fr.first_fragment + g;
            if (fg.flags & ~(OHGPU_FLAG_RAMP | OHGPU_FLAG_SILENCE))
                return set_error(OHGPU_ERR_INVALID, "ohm fragment %zu: unknown flag bits 0x%x", gi, fg.flags);
            const bool little = s.src_endian == OHGPU_ENDIAN_LITTLE && s.src_bits > 8;
            const bool plain = !(fg.flags & (OHGPU_FLAG_RAMP | OHGPU_FLAG_SILENCE)) && fg.attenuation == OHGPU_UNITY_ATTENUATION && !little;
            const uint64_t src_bytes = (uint64_t)fg.n_frames * ch * (s.src_bits / 8);
            ohgpu_msg_desc m;
            memset(&m, 0, sizeof(m));
            m.src_offset = fg.src_offset;
            m.n_frames = fg.n_frames;
            m.ramp_start = fg.ramp_start;
            m.ramp_end = fg.ramp_end;
            m.attenuation = fg.attenuation;
            m.channels = (uint8_t)ch;
            m.src_bits = s.src_bits;
            m.src_endian = little ? OHGPU_ENDIAN_LITTLE : OHGPU_ENDIAN_BIG;
            m.dst_endian = OHGPU_ENDIAN_BIG;
            m.flags = fg.flags;
            ohgpu_fmt_desc p;
            memset(&p, 0, sizeof(p));
            p.dst_offset = at;
            p.n_frames = fg.n_frames;
            p.kind = OHGPU_FMT_SENDER_PACK;
            p.channels = (uint8_t)ch;
            p.src_bits = s.src_bits;
            if (fg.n_frames == 0) {
                // nothing to write
            } else if (ch <= 2) {
                m.dst_offset = at;
                m.dst_bits = (uint8_t)(wb * 8);
                direct.push_back(m);
                // The frame's first audio is this message's: its wave writes the header too (the bytes right before `at`), so a
                // batch of mono / stereo streams is ONE launch.  The header as it goes on the wire: the record's 36 bytes, then
                // the stream's (what ohm_header_kernel assembles from the two records for every other frame).
                MsgPrefix pre = {0, 0};
                if (at == fr.dst_offset + header_bytes) {
                    pre.off = (uint32_t)blob.size(); pre.bytes = header_bytes;
                    blob.insert(blob.end(), h, h + kPerFrameHeader);
                    blob.insert(blob.end(), &stream_recs[(size_t)fr.stream * 64], &stream_recs[(size_t)fr.stream * 64] + (header_bytes - kPerFrameHeader));
                    blob.resize((blob.size() + 3) & ~(size_t)3, 0);
                    folded[f] = 1;
                }
                direct_prefix.push_back(pre);
            } else if (plain) {
                p.src_offset = fg.src_offset;
                select.push_back(p);
            } else if (!(fg.flags & OHGPU_FLAG_SILENCE)) {
                // ramp / attenuation on a wider stream: one pass over the two channels that go on the wire
                if (ch > OHGPU_MAX_CHANNELS)
                    return set_error(OHGPU_ERR_UNSUPPORTED, "ohm fragment %zu: ramp / attenuation on %u channels (MsgPlayable carries at most 8)", gi, ch);
                if (fg.ramp_start > OHGPU_RAMP_MAX || fg.ramp_end > OHGPU_RAMP_MAX)
                    return set_error(OHGPU_ERR_INVALID, "ohm fragment %zu: ramp [%u..%u] beyond Ramp::kMax", gi, fg.ramp_start, fg.ramp_end);
                if ((fg.flags & OHGPU_FLAG_RAMP) && fg.n_frames > 131071u)      // i*iTotalRamp is TInt arithmetic (Msg.cpp:835)
                    return set_error(OHGPU_ERR_INVALID, "ohm fragment %zu: ramped fragment of %u frames overflows the reference's TInt ramp product", gi, fg.n_frames);
                if (fg.attenuation != OHGPU_UNITY_ATTENUATION && s.src_bits != 16)  // ASSERT(iBitDepth == 16), Msg.cpp:2741
                    return set_error(OHGPU_ERR_UNSUPPORTED, "ohm fragment %zu: attenuation %u on %u-bit audio (16-bit only)", gi, fg.attenuation, s.src_bits);
                if (fg.src_offset > src_arena_bytes || src_bytes > src_arena_bytes - fg.src_offset)
                    return set_error(OHGPU_ERR_BOUNDS, "ohm fragment %zu: reads [%llu, +%llu) beyond the %llu-byte source arena", gi,
                                     (unsigned long long)fg.src_offset, (unsigned long long)src_bytes, (unsigned long long)src_arena_bytes);
                OhmSelRec sr;
                memset(&sr, 0, sizeof(sr));
                sr.src_off = fg.src_offset; sr.dst_off = at; sr.n_frames = fg.n_frames;
                sr.ramp_start = fg.ramp_start; sr.ramp_end = fg.ramp_end; sr.attenuation = fg.attenuation;
                sr.channels = (uint8_t)ch; sr.sb = (uint8_t)(s.src_bits / 8); sr.first_ch = (uint8_t)(ch < 10 ? 0 : 8);
                sr.flags = fg.flags; sr.little = little ? 1 : 0;
                uint32_t sh = 0;
                magic_u31(fg.n_frames > 1 ? fg.n_frames - 1 : 1, &sr.m_n1, &sh);
                sr.s_n1 = (uint8_t)sh;
                selr.push_back(sr);
            } else {
                if (ch > OHGPU_MAX_CHANNELS)
                    return set_error(OHGPU_ERR_UNSUPPORTED, "ohm fragment %zu: silence on %u channels (MsgPlayable carries at most 8)", gi, ch);
                m.dst_offset = scratch_bytes;
                m.dst_bits = s.src_bits;
                stage.push_back(m);
                p.src_offset = scratch_bytes;
                select_staged.push_back(p);
                scratch_bytes += (src_bytes + 63) & ~63ull;
            }
            in_frames += fg.n_frames;
            if (!(fg.flags & OHGPU_FLAG_SILENCE)) src_touched += src_bytes;
            at += (uint64_t)fg.n_frames * wch * wb;
        }
        dst_written += frame_bytes;
    }

    ohgpu_batch* b = new (std::nothrow) ohgpu_batch();
    if (!b) return set_error(OHGPU_ERR_NOMEM, "ohgpu_ohm_batch_create: out of host memory");
    b->kind = kBatchOhm;
    b->n = n_frames;
    b->src_arena_bytes = src_arena_bytes;
    b->dst_arena_bytes = dst_arena_bytes;
    b->in_frames = b->out_frames = in_frames;
    b->src_bytes_touched = src_touched;
    b->dst_bytes_written = dst_written;
    OhmPlan& plan = b->ohm;
    plan.n_frames = (uint32_t)n_frames;
    int err = OHGPU_OK;
    auto dev_copy = [&](void** d, const void* h, size_t bytes) -> int {
        if (bytes == 0) return OHGPU_OK;
        hipError_t e = hipMalloc(d, bytes);
        if (e == hipErrorOutOfMemory) return set_error(OHGPU_ERR_NOMEM, "ohgpu_ohm_batch_create: out of device memory");
        OHGPU_HIP_TRY(e);
        if (h) OHGPU_HIP_TRY(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return OHGPU_OK;
    };
    if (blob.size() > 0xffffff00ull) {                                  // (prefix offsets are 32-bit: such a batch keeps the header kernel)
        blob.clear();
        std::fill(folded.begin(), folded.end(), 0);
    }
    if (!direct.empty())
        err = pcm_batch_create_prefixed(ctx, direct.data(), direct.size(), src_arena_bytes, dst_arena_bytes,
                                        blob.empty() ? nullptr : direct_prefix.data(), blob.data(), blob.size(), &plan.direct);
    // frames whose header no audio pass writes first; the rest follow, for the runs in which the line kernel is not the one used
    const bool folds = plan.direct && plan.direct->line.prefixed;
    std::vector<OhmFrameRec> ordered;
    ordered.reserve(n_frames);
    for (size_t f = 0; f < n_frames; f++) if (!(folds && folded[f])) ordered.push_back(recs[f]);
    plan.n_unfolded = (uint32_t)ordered.size();
    for (size_t f = 0; f < n_frames; f++) if (folds && folded[f]) ordered.push_back(recs[f]);
    recs.swap(ordered);
    if (err == OHGPU_OK && !select.empty())
        err = ohgpu_fmt_batch_create(ctx, select.data(), select.size(), src_arena_bytes, dst_arena_bytes, &plan.select);
    if (err == OHGPU_OK && !stage.empty())
        err = ohgpu_pcm_batch_create(ctx, stage.data(), stage.size(), src_arena_bytes, scratch_bytes, &plan.stage);
    if (err == OHGPU_OK && !select_staged.empty())
        err = ohgpu_fmt_batch_create(ctx, select_staged.data(), select_staged.size(), scratch_bytes, dst_arena_bytes, &plan.select_staged);
    if (err == OHGPU_OK) err = dev_copy(&plan.d_scratch, nullptr, scratch_bytes);
    if (err == OHGPU_OK) err = dev_copy(&plan.d_selr, selr.data(), selr.size() * sizeof(OhmSelRec));
    plan.n_selr = (uint32_t)selr.size();
    if (err == OHGPU_OK) err = dev_copy(&plan.d_frames, recs.data(), recs.size() * sizeof(OhmFrameRec));
    if (err == OHGPU_OK) err = dev_copy(&plan.d_streams, stream_recs.data(), stream_recs.size());
    if (err != OHGPU_OK) { free_ohm(ctx, b); delete b; return err; }
    *out = b;
    return OHGPU_OK;
}

int ohgpu_ohm_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream)
{
    if (!ctx) return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_run: null context");
    OHGPU_HIP_TRY(hipSetDevice(ctx->device));
    if (!batch || batch->kind != kBatchOhm) return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_run: not a Songcast frame batch");
    if (batch->n == 0) return OHGPU_OK;
    if (!dst_base || (!src_base && batch->src_bytes_touched)) return set_error(OHGPU_ERR_INVALID, "ohgpu_ohm_batch_run: null arena pointer");
    const OhmPlan& p = batch->ohm;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int err = OHGPU_OK;
    if (p.direct) err = ohgpu_pcm_batch_run(ctx, p.direct, src_base, dst_base, s);
    if (err == OHGPU_OK && p.select) err = ohgpu_fmt_batch_run(ctx, p.select, src_base, dst_base, s);
    if (err == OHGPU_OK && p.stage) err = ohgpu_pcm_batch_run(ctx, p.stage, src_base, p.d_scratch, s);
    if (err == OHGPU_OK && p.select_staged) err = ohgpu_fmt_batch_run(ctx, p.select_staged, p.d_scratch, dst_base, s);
    if (err != OHGPU_OK) return err;
    if (p.n_selr) {
        const uint32_t blocks = (p.n_selr + 3) / 4 < 4096u ? (p.n_selr + 3) / 4 : 4096u;
        hipLaunchKernelGGL(ohm_select_ramp_kernel, dim3(blocks), dim3(256), 0, s, (const OhmSelRec*)p.d_selr, p.n_selr,
                           (const uint16_t*)ctx->d_ramp_table, (const uint8_t*)src_base, batch->src_arena_bytes, (uint8_t*)dst_base);
        OHGPU_HIP_TRY(hipGetLastError());
    }
    // headers: those the direct pass has not written (all of them when the generic kernel ran it, ohgpu_set_kernel_variant(1))
    const uint32_t n_headers = ctx->variant == 1 ? p.n_frames : p.n_unfolded;
    if (n_headers) {
        const uint32_t threads = 256, frames_per_block = threads / kLanesPerFrame * kFramesPerGroup;
        const uint32_t blocks = (n_headers + frames_per_block - 1) / frames_per_block;
        hipLaunchKernelGGL(ohm_header_kernel, dim3(blocks), dim3(threads), 0, s,
                           (const OhmFrameRec*)p.d_frames, n_headers, (const uint8_t*)p.d_streams, (uint8_t*)dst_base);
        OHGPU_HIP_TRY(hipGetLastError());
    }
    return OHGPU_OK;
}

int ohgpu_ohm_process_host(ohgpu_ctx* ctx, const ohgpu_ohm_stream* streams, size_t n_streams,
                           const ohgpu_ohm_frame_desc* frames, size_t n_frames,
                           const ohgpu_ohm_fragment* fragments, size_t n_fragments,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes)
{
    ohgpu_batch* b = nullptr;
    int err = ohgpu_ohm_batch_create(ctx, streams, n_streams, frames, n_frames, fragments, n_fragments, src_bytes, dst_bytes, &b);
    if (err != OHGPU_OK) return err;
    void *d_src = nullptr, *d_dst = nullptr;
    err = ohgpu_malloc(ctx, src_bytes, &d_src);
    if (err == OHGPU_OK) err = ohgpu_malloc(ctx, dst_bytes, &d_dst);
    if (err == OHGPU_OK && src_bytes) err = ohgpu_memcpy_h2d(ctx, d_src, src_host, src_bytes, nullptr);
    if (err == OHGPU_OK && dst_bytes) err = ohgpu_memcpy_h2d(ctx, d_dst, dst_host, dst_bytes, nullptr);   // bytes no frame covers stay as given
    if (err == OHGPU_OK) err = ohgpu_ohm_batch_run(ctx, b, d_src, d_dst, nullptr);
    if (err == OHGPU_OK && dst_bytes) err = ohgpu_memcpy_d2h(ctx, dst_host, d_dst, dst_bytes, nullptr);
    if (err == OHGPU_OK) err = ohgpu_stream_sync(ctx, nullptr);
    if (d_src) ohgpu_free(ctx, d_src);
    if (d_dst) ohgpu_free(ctx, d_dst);
    ohgpu_batch_destroy(ctx, b);
    return err;
}

}  // extern "C"
