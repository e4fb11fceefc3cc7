// ohm_frame_kernel.hip -- Songcast sender frames: the C ABI's ohgpu_ohm_* entry points.
//
// A frame = header + audio.  The audio of every fragment is an ordinary message for the kernels that already exist:
//   mono / stereo streams: one PCM message (ramp, attenuation, 32 -> 24 bit) written straight into the frame -- the
//     Sender's pack of such a stream IS the depth conversion (Sender.cpp:351-377 keeps min(bytes, 3) leading bytes);
//   wider streams: ohm_wide_kernel -- the Sender pack's channel select with the playable's attenuation and ramp, one pass
//     over the two channels that go on the wire; silent fragments (MsgPlayableSilence's channel-id bytes, Msg.cpp:2874-2893)
//     take a PCM pass into a scratch arena and fmt_line_kernel's select.
// What is new here is the header: 36 per-frame bytes (OhmHeader + the per-frame part of OhmMsgAudio::Serialise,
// OhmMsg.cpp:363-413) and the per-stream 22 + codec bytes (GetStreamHeader, :225-241).  A mono / stereo frame's header is
// assembled on the host when the batch is created and travels as the PREFIX of the frame's first audio message: the wave
// that writes that audio writes the header in front of it (pcm_line_kernel), one launch for the whole batch; ohm_wide_kernel
// does the same for the wider streams' frames.  Every other frame's header (frames without audio, frames that begin with
// silence on a wider stream) is written by ohm_header_kernel, 16 lanes per frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

template <uint32_t N> struct GroupOut;
template <> struct GroupOut<1> { typedef uint32_t type __attribute__((ext_vector_type(1))); };
template <> struct GroupOut<2> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct GroupOut<3> { typedef uint32_t type __attribute__((ext_vector_type(3))); };

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

// Audible fragments of streams with more than two channels: what MsgPlayablePcm::ReadBlock (attenuation, then
// RampApplicator; Msg.cpp:2736-2786) and Sender::DoProcessFragment (two channels from FirstChannelToSend, <= 3 leading bytes
// each; Sender.cpp:351-377) do to the two channels that go on the wire, in one pass -- and the frame's header in front of
// them when the fragment is the frame's first audio (the prefix scheme of pcm_line_kernel).  A wave per fragment; a lane
// = TWO frames: one 8-byte load per frame (the two wire subsamples lie next to each other; any alignment, scalar base +
// lane offset), 4 x WB output bytes in ONE store.  256 frames' loads -- four per lane -- are issued before any is used.
//   plain:   the wire bytes are a byte shuffle of the loaded ones, one v_perm_b32 per subsample;
//   ramped:  RampApplicator reads the top 16 bits and writes two bytes back over zeros: one v_perm_b32 lifts them into the
//            top half of a register, one 24-bit multiply by twice the Q15 multiplier, one v_perm_b32 takes the product's
//            two bytes (pcm_line_kernel's ramped group path; one multiplier look-up per frame);
//   attenuated (16-bit audio, RAOP): pcm_device.h's general expressions.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
ohm_wide_kernel(const OhmSelRec* __restrict__ recs, uint32_t n_recs, const uint16_t* __restrict__ ramp_table,
                const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint8_t* __restrict__ prefix)
{
    __shared__ uint16_t s_ramp2[kRampTableCount];                       // twice the multiplier
    __shared__ uint16_t s_ramp[kRampTableCount];
    for (uint32_t i = threadIdx.x; i < kRampTableCount; i += 256) { s_ramp[i] = ramp_table[i]; s_ramp2[i] = (uint16_t)(2u * ramp_table[i]); }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, waves = gridDim.x * 4;     // (launched with 256 threads: literal, so that the record index stays scalar)
    typedef uint32_t V2 __attribute__((ext_vector_type(2)));
    for (uint32_t i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); i < n_recs; i += waves) {
        const OhmSelRec r = recs[i];
        // (the record's byte fields come through a vector load: back to scalar registers, the loads' base has to be one)
        const uint32_t sb = __builtin_amdgcn_readfirstlane(r.sb), ch = __builtin_amdgcn_readfirstlane(r.channels), n = r.n_frames;
        const uint32_t first_ch = __builtin_amdgcn_readfirstlane(r.first_ch);
        const uint8_t* const sp = src + r.src_off + first_ch * sb;                  // frame f's two wire subsamples start at sp + f * ch * sb
        uint8_t* const dp = dst + r.dst_off;
        const uint32_t fb = ch * sb;
        const bool ramp = (r.flags & OHGPU_FLAG_RAMP) != 0, atten = r.attenuation != OHGPU_UNITY_ATTENUATION;
        const int32_t total = (int32_t)r.ramp_start - (int32_t)r.ramp_end;
        // Loads are issued from asm statements and waited for by a later one: between the two the destination register must
        // not be copied, and a branch or a loop entry in between makes the compiler do exactly that ("the rule
        // behind the hand-counted waits").  So every load here is UNCONDITIONAL -- lanes with nothing to fetch read a clamped,
        // valid address and ignore the result -- and load ... wait is always straight-line code.
        const uint32_t hb = r.prefix_bytes, safe = r.safe_frames;
        const uint32_t hpos = hb == 0 ? 0u : (lane * 4 + 4 <= hb ? lane * 4 : hb - 4);
        const uint8_t* const hbase = hb ? prefix + r.prefix_off : (const uint8_t*)ramp_table;  // (the ramp table is always there to be read)
        const uint8_t* const lbase = safe ? sp : (const uint8_t*)ramp_table;
        auto load_frame = [&](uint32_t f, V2& v) __attribute__((always_inline)) {   // in flight until the next s_waitcnt
            const uint32_t off = safe ? (f < safe ? f : safe - 1) * fb : 0u;
            asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(v) : "v"(off), "s"(lbase) : "memory");
        };
        auto load_frame_at_the_arena_end = [&](uint32_t f, V2& v) __attribute__((always_inline)) {   // (after the wait) the 8 bytes would
            if (f >= safe && f < n) {                                    // cross the arena's end: the subsamples' own bytes, one by one
                uint32_t lo = 0, hi = 0;
                for (uint32_t k = 0; k < 2 * sb; k++) {
                    const uint32_t byte = sp[(size_t)f * fb + k];
                    if (k < 4) lo |= byte << (8 * k); else hi |= byte << (8 * (k - 4));
                }
                v[0] = lo; v[1] = hi;
            }
        };
        // ---- the header and the first 256 frames ----
        uint32_t hv = 0;
        V2 a0 = {}, a1 = {}, b0 = {}, b1 = {};
        asm volatile("global_load_dword %0, %1, %2" : "+v"(hv) : "v"(hpos), "s"(hbase) : "memory");
        load_frame(2 * lane, a0); load_frame(2 * lane + 1, a1);
        load_frame(2 * (lane + 64), b0); load_frame(2 * (lane + 64) + 1, b1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(hv) : : "memory");
        if (lane * 4 < hb) {
            uint8_t* const obase = dp - hb;
            asm volatile("global_store_dword %0, %1, %2" : : "v"(hpos), "v"(hv), "s"(obase) : "memory");
        }
        auto body = [&](auto wb_tag) __attribute__((always_inline)) {
            constexpr uint32_t WB = decltype(wb_tag)::value;
            typedef typename GroupOut<WB>::type Out;
            // selectors over {d1, d0}: subsample k's wire bytes in memory order (plain), its two top bytes into the top half (ramp)
            uint32_t sel_plain[2], sel_top[2];
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                uint32_t a = 0, b = 0;
                const uint32_t o = k * sb;
#pragma unroll
                for (uint32_t m = 0; m < 4; m++) a |= (m < WB ? (r.little ? o + sb - 1 - m : o + m) : 0x0cu) << (8 * m);
                b = ((r.little ? o + sb - 1 : o) << 24) | ((sb > 1 ? (r.little ? o + sb - 2 : o + 1) : 0x0cu) << 16) | 0x0c0cu;
                sel_plain[k] = a; sel_top[k] = b;
            }
            const uint32_t abs_total = (uint32_t)(total < 0 ? -total : total) & 0x1ffffu;
            const uint32_t neg_mask = total < 0 ? 0xffffffffu : 0u, ramp_base = (uint32_t)r.ramp_start + neg_mask;
            const bool fast_ramp = ramp && !atten && r.m_n1 != 0 && n <= 32768u;
            auto mult2 = [&](uint32_t f) __attribute__((always_inline)) -> uint32_t {       // pcm_line_kernel's ramp_mult2
                const uint32_t mag = __umulhi((f & 0xffffu) * abs_total, r.m_n1) >> r.s_n1;
                const uint32_t ramp16 = (ramp_base - (mag ^ neg_mask)) & 0xffffu;
                const uint32_t idx = (kRampMax + (1u << 4) - ramp16) >> 5;
                return s_ramp2[idx < kRampTableCount - 1 ? idx : kRampTableCount - 1];
            };
            // one frame -> its two subsamples' wire bytes (memory order, first byte low)
            auto frame_bytes = [&](uint32_t f, const V2& d, uint32_t& v0, uint32_t& v1) __attribute__((always_inline)) {
                if (fast_ramp) {
                    const uint32_t m2 = mult2(f);
                    const uint32_t p0 = (uint32_t)(((int32_t)__builtin_amdgcn_perm(d[1], d[0], sel_top[0]) >> 16) * (int32_t)m2);
                    const uint32_t p1 = (uint32_t)(((int32_t)__builtin_amdgcn_perm(d[1], d[0], sel_top[1]) >> 16) * (int32_t)m2);
                    constexpr uint32_t take = WB == 1 ? 0x0c0c0c03u : 0x0c0c0203u;          // product bytes 3, 2, then zeros
                    v0 = __builtin_amdgcn_perm(0u, p0, take);
                    v1 = __builtin_amdgcn_perm(0u, p1, take);
                } else if (!ramp && !atten) {
                    v0 = __builtin_amdgcn_perm(d[1], d[0], sel_plain[0]);
                    v1 = __builtin_amdgcn_perm(d[1], d[0], sel_plain[1]);
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 2; k++) {
                        // left-justified big-endian word of subsample k
                        uint32_t w = 0;
                        const uint64_t both = ((uint64_t)d[1] << 32) | d[0];
                        for (uint32_t m = 0; m < sb; m++) w |= (uint32_t)((both >> (8 * (k * sb + (r.little ? sb - 1 - m : m)))) & 0xffu) << (24 - 8 * m);
                        if (atten) w = attenuate_word(w, r.attenuation);
                        if (ramp) w = ramp_word(w, s_ramp[ramp_index_magic(r.ramp_start, total, f, n, r.m_n1, r.s_n1)], sb, ch, first_ch + k);
                        const uint32_t v = __builtin_bswap32(w) & (0xffffffffu >> (32 - 8 * WB));
                        if (k == 0) v0 = v; else v1 = v;
                    }
                }
            };
            auto put_pair = [&](uint32_t g, const V2& da, const V2& db) __attribute__((always_inline)) {   // frames 2g, 2g + 1
                uint32_t v[4] = {0, 0, 0, 0};
                frame_bytes(2 * g, da, v[0], v[1]);
                const bool second = 2 * g + 1 < n;
                if (second) frame_bytes(2 * g + 1, db, v[2], v[3]);
                uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t o = k * WB;
                    ow[o >> 2] |= v[k] << (8 * (o & 3));
                    if ((o & 3) + WB > 4) ow[(o >> 2) + 1] |= v[k] >> (32 - 8 * (o & 3));
                }
                const uint32_t at = g * (4 * WB);
                if (second) {
                    Out out;
#pragma unroll
                    for (uint32_t j = 0; j < WB; j++) out[j] = ow[j];
                    if constexpr (WB == 1) asm volatile("global_store_dword %0, %1, %2" : : "v"(at), "v"(ow[0]), "s"(dp) : "memory");
                    else if constexpr (WB == 2) asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(at), "v"(out), "s"(dp) : "memory");
                    else asm volatile("global_store_dwordx3 %0, %1, %2" : : "v"(at), "v"(out), "s"(dp) : "memory");
                } else {                                                 // an odd fragment's last frame
                    for (uint32_t bq = 0; bq < 2 * WB; bq++) dp[(size_t)at + bq] = (uint8_t)(ow[bq >> 2] >> (8 * (bq & 3)));
                }
            };
            auto put_four = [&](uint32_t g0, V2& fa0, V2& fa1, V2& fb0, V2& fb1) __attribute__((always_inline)) {
                const uint32_t ga = g0 + lane, gb = ga + 64;
                if (safe < n) {                                          // (wave-uniform: the arena's last fragment only)
                    load_frame_at_the_arena_end(2 * ga, fa0); load_frame_at_the_arena_end(2 * ga + 1, fa1);
                    load_frame_at_the_arena_end(2 * gb, fb0); load_frame_at_the_arena_end(2 * gb + 1, fb1);
                }
                if (2 * ga < n) put_pair(ga, fa0, fa1);
                if (2 * gb < n) put_pair(gb, fb0, fb1);
            };
            put_four(0, a0, a1, b0, b1);
            for (uint32_t g0 = 128; 2 * g0 < n; g0 += 128) {             // fragments of more than 256 frames
                const uint32_t ga = g0 + lane, gb = ga + 64;
                V2 c0 = {}, c1 = {}, d0 = {}, d1 = {};
                load_frame(2 * ga, c0); load_frame(2 * ga + 1, c1);
                load_frame(2 * gb, d0); load_frame(2 * gb + 1, d1);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1) : : "memory");
                put_four(g0, c0, c1, d0, d1);
            }
        };
        const uint32_t wb = sb < 3 ? sb : 3;
        if (wb == 3) body(std::integral_constant<uint32_t, 3>{});
        else if (wb == 2) body(std::integral_constant<uint32_t, 2>{});
        else body(std::integral_constant<uint32_t, 1>{});
    }
}

// A plain record (no ramp, unity attenuation, no header); the callers add what their fragment carries.
OhmSelRec wide_record(uint64_t src_off, uint64_t dst_off, uint32_t n_frames, uint32_t channels, uint32_t sb, bool little, uint64_t src_arena_bytes)
{
    OhmSelRec sr;
    memset(&sr, 0, sizeof(sr));
    sr.src_off = src_off; sr.dst_off = dst_off; sr.n_frames = n_frames;
    sr.ramp_start = sr.ramp_end = OHGPU_RAMP_MAX; sr.attenuation = OHGPU_UNITY_ATTENUATION;
    sr.channels = (uint8_t)channels; sr.sb = (uint8_t)sb; sr.first_ch = (uint8_t)(channels < 10 ? 0 : 8);   // Sender::FirstChannelToSend, Sender.cpp:351-354
    sr.little = little ? 1 : 0;
    uint32_t sh = 0;
    magic_u31(n_frames > 1 ? n_frames - 1 : 1, &sr.m_n1, &sh);
    sr.s_n1 = (uint8_t)sh;
    // frames whose 8-byte read (their two wire subsamples and what follows) ends inside the arena
    const uint64_t first = src_off + (uint64_t)sr.first_ch * sb, fbytes = (uint64_t)channels * sb;
    sr.safe_frames = first + 8 > src_arena_bytes ? 0u : (uint32_t)std::min<uint64_t>(n_frames, (src_arena_bytes - 8 - first) / fbytes + 1);
    return sr;
}

hipError_t launch_ohm_wide(const ohgpu_ctx* ctx, const void* d_recs, uint32_t n_recs, const uint8_t* src, uint8_t* dst, const uint8_t* prefix, hipStream_t s)
{
    if (n_recs == 0) return hipSuccess;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
#ifndef OHGPU_LINE_OHM_GROUPS_PER_CU
#define OHGPU_LINE_OHM_GROUPS_PER_CU 4                                                            // (round 5: four workgroups per CU serve a streaming kernel better than eight: six-channel frames 0.293-0.311 -> 0.276-0.290 ms)
#endif
    const uint32_t blocks = (n_recs + 3) / 4 < cus * OHGPU_LINE_OHM_GROUPS_PER_CU ? (n_recs + 3) / 4 : cus * OHGPU_LINE_OHM_GROUPS_PER_CU;
    hipLaunchKernelGGL(ohm_wide_kernel, dim3(blocks), dim3(256), 0, s, (const OhmSelRec*)d_recs, n_recs,
                       (const uint16_t*)ctx->d_ramp_table, src, dst, prefix);
    return hipGetLastError();
}

static inline void put_be(uint8_t* p, uint64_t v, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * (n - 1 - i)));
}

void free_ohm(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    OhmPlan& p = b->ohm;
    ohgpu_batch_destroy(ctx, p.direct);
    ohgpu_batch_destroy(ctx, p.stage);
    ohgpu_batch_destroy(ctx, p.select_staged);
    if (p.d_scratch) hipFree(p.d_scratch);
    if (p.d_selr) hipFree(p.d_selr);
    if (p.d_wide_prefix) hipFree(p.d_wide_prefix);
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
    std::vector<uint8_t> wide_blob;                                    // the same for the wider streams' frames (ohm_wide_kernel)
    std::vector<uint8_t> folded(n_frames, 0);                          // 1: the header rides with `direct`, 2: with ohm_wide_kernel
    std::vector<ohgpu_fmt_desc> select_staged;
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
            } else if (!(fg.flags & OHGPU_FLAG_SILENCE)) {
                // a wider stream's audible fragment: one pass over the two channels that go on the wire
                if (!plain && ch > OHGPU_MAX_CHANNELS)
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
                if (src_bytes > 0xffffffffull)
                    return set_error(OHGPU_ERR_INVALID, "ohm fragment %zu: %llu source bytes (the limit is 4 GiB - 1)", gi, (unsigned long long)src_bytes);
                OhmSelRec sr = wide_record(fg.src_offset, at, fg.n_frames, ch, s.src_bits / 8, little, src_arena_bytes);
                sr.ramp_start = fg.ramp_start; sr.ramp_end = fg.ramp_end; sr.attenuation = fg.attenuation;
                sr.flags = fg.flags;
                if (at == fr.dst_offset + header_bytes && wide_blob.size() < 0xffffff00ull) {   // the frame's first audio: the header rides along
                    sr.prefix_off = (uint32_t)wide_blob.size(); sr.prefix_bytes = (uint8_t)header_bytes;
                    wide_blob.insert(wide_blob.end(), h, h + kPerFrameHeader);
                    wide_blob.insert(wide_blob.end(), &stream_recs[(size_t)fr.stream * 64], &stream_recs[(size_t)fr.stream * 64] + (header_bytes - kPerFrameHeader));
                    wide_blob.resize((wide_blob.size() + 3) & ~(size_t)3, 0);
                    folded[f] = 2;
                }
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
        for (auto& x : folded) if (x == 1) x = 0;
    }
    if (!direct.empty())
        err = pcm_batch_create_prefixed(ctx, direct.data(), direct.size(), src_arena_bytes, dst_arena_bytes,
                                        blob.empty() ? nullptr : direct_prefix.data(), blob.data(), blob.size(), &plan.direct);
    // ohm_header_kernel's list: first the frames whose header no audio pass writes, then those `direct` writes (the header kernel
    // takes them too in the runs in which the generic kernel does `direct`'s work); ohm_wide_kernel always writes its own.
    const bool folds = plan.direct && plan.direct->line.prefixed;
    std::vector<OhmFrameRec> ordered;
    ordered.reserve(n_frames);
    for (size_t f = 0; f < n_frames; f++) if (folded[f] == 0 || (folded[f] == 1 && !folds)) ordered.push_back(recs[f]);
    plan.n_unfolded = (uint32_t)ordered.size();
    for (size_t f = 0; f < n_frames; f++) if (folded[f] == 1 && folds) ordered.push_back(recs[f]);
    plan.n_unfolded_generic = (uint32_t)ordered.size();
    recs.swap(ordered);
    if (err == OHGPU_OK && !stage.empty())
        err = ohgpu_pcm_batch_create(ctx, stage.data(), stage.size(), src_arena_bytes, scratch_bytes, &plan.stage);
    if (err == OHGPU_OK && !select_staged.empty())
        err = ohgpu_fmt_batch_create(ctx, select_staged.data(), select_staged.size(), scratch_bytes, dst_arena_bytes, &plan.select_staged);
    if (err == OHGPU_OK) err = dev_copy(&plan.d_scratch, nullptr, scratch_bytes);
    if (err == OHGPU_OK) err = dev_copy(&plan.d_selr, selr.data(), selr.size() * sizeof(OhmSelRec));
    plan.n_selr = (uint32_t)selr.size();
    if (err == OHGPU_OK) err = dev_copy(&plan.d_wide_prefix, wide_blob.data(), wide_blob.size());
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
    if (err == OHGPU_OK && p.stage) err = ohgpu_pcm_batch_run(ctx, p.stage, src_base, p.d_scratch, s);
    if (err == OHGPU_OK && p.select_staged) err = ohgpu_fmt_batch_run(ctx, p.select_staged, p.d_scratch, dst_base, s);
    if (err != OHGPU_OK) return err;
    OHGPU_HIP_TRY(launch_ohm_wide(ctx, p.d_selr, p.n_selr, (const uint8_t*)src_base, (uint8_t*)dst_base, (const uint8_t*)p.d_wide_prefix, s));
    // headers: those the direct pass has not written (all of them when the generic kernel ran it, ohgpu_set_kernel_variant(1))
    const uint32_t n_headers = ctx->variant == 1 ? p.n_unfolded_generic : p.n_unfolded;
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
    // (a frame's datagram = header + the audio of its fragments: the bytes the call hands back; bytes no frame covers stay as given)
    std::vector<std::pair<uint64_t, uint64_t>> out(n_frames);
    for (size_t i = 0; i < n_frames && err == OHGPU_OK; i++) {
        uint32_t samples = 0, header = 0, total = 0;
        for (uint32_t k = 0; k < frames[i].n_fragments; k++) samples += fragments[frames[i].first_fragment + k].n_frames;
        err = ohgpu_ohm_frame_layout(&streams[frames[i].stream], samples, &header, &total);
        out[i] = {frames[i].dst_offset, total};
    }
    if (err == OHGPU_OK)
        err = ohgpu::host_roundtrip(ctx, src_host, src_bytes, dst_host, dst_bytes, out,
                                    [&](const void* d_src, void* d_dst) { return ohgpu_ohm_batch_run(ctx, b, d_src, d_dst, nullptr); });
    ohgpu_batch_destroy(ctx, b);
    return err;
}

}  // extern "C"
