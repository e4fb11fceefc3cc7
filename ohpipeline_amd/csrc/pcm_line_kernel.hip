// pcm_line_kernel.hip -- the tuned kernel of the PCM message path: unpack ->
// attenuate -> ramp | silence -> pack, any depth / byte order / channel count / alignment, line-coalesced.
//
// HBM-bound byte work (about 0.5 integer op per byte), so the design is about memory instructions, not arithmetic.
// Two paths share the chunk records and the launch:
//   REGISTERS (16/24/32-bit audio on both sides: pcm_line_kernel<SB, DB>, one launch per depth pair of the batch).  A
//     chunk is a message or a run of a stream's plain messages; lane = a GROUP of four subsamples -- one unaligned
//     8/12/16-byte load on a scalar base, one store; a trip of the wave's loop issues the first 128 groups of TWO chunks
//     before it touches either.  Plain: a byte shuffle.  Ramped: shuffle -> 24-bit multiply -> shuffle (RampApplicator
//     uses a subsample's top 16 bits and writes two bytes back), two multiplier look-ups per group.  Attenuated and the
//     rest: pcm_device.h's expressions on the same groups.  A chunk may carry a PREFIX (a Songcast frame's header) that its
//     wave writes in front of the audio.
//   STAGED (8-bit audio, silence: pcm_line_kernel<0, 0> and generic_chunk), described first:
//   * A message is cut into CHUNKS of <= 512 subsamples (host, at batch creation), each described by one 64-byte
//     record that a single scalar load fetches; one wave owns a chunk at a time.
//   * In: the aligned 16-byte pieces that cover the chunk's source bytes go straight to LDS
//     (global_load_lds_dwordx4, 64 pieces per instruction).  A piece always overlaps bytes the message owns, so it
//     never leaves the page those bytes live in, whatever the arena's end looks like.
//   * Lane = aligned destination dword.  For each of the 1..4 subsamples whose bytes fall into it, the lane reads
//     the two aligned LDS words that hold the source bytes, builds the reference's left-justified big-endian word,
//     applies attenuation / ramp / silence with pcm_device.h's expressions and keeps the destination bytes in MEMORY
//     order; it funnels them together and stores 4 bytes: 64 lanes write 256 contiguous bytes.  A subsample that
//     straddles two dwords is transformed twice (cheaper than a second trip through LDS).  Only a chunk's first and
//     last dword can be partial; those are written byte by byte (another message may own the other bytes).
//   * The two integer divisions of the ramp (frame = subsample / channels, ramp = i * total / (frames - 1)) use
//     per-message multipliers computed on the host (exact for operands < 2^31, which validation guarantees).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* global_ptr_t;

constexpr uint32_t kChunkSub = 512;                                   // subsamples per chunk
constexpr uint32_t kLineWaves = 4;                                    // waves per workgroup
constexpr uint32_t kGroupChunkSub = 4096;                             // subsamples per merged group-path chunk
constexpr uint32_t kInBytes = ((15 + kChunkSub * 4 + 15) / 16) * 16 + 16;   // <= 3 staging instructions of 64 pieces

// ---- group path helpers: 4*N bytes at any byte address <-> an N-register vector (inline asm: the access is ONE
// instruction whatever the alignment).  A load's result must not be touched before the caller's s_waitcnt vmcnt(0), which
// names the vectors as "+v"; only then are they taken apart. ----
template <int N> struct GroupVec;
template <> struct GroupVec<2> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct GroupVec<3> { typedef uint32_t type __attribute__((ext_vector_type(3))); };
template <> struct GroupVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4))); };

template <int N>
__device__ __forceinline__ void group_load(typename GroupVec<N>::type& v, const uint8_t* base, uint32_t off)
{
    // (wave-uniform 64-bit base in scalar registers + a 32-bit lane offset: no 64-bit address arithmetic per lane)
    if constexpr (N == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    else if constexpr (N == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void group_put(uint8_t* base, uint32_t off, const typename GroupVec<N>::type& o)
{
    if constexpr (N == 2) asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(off), "v"(o), "s"(base) : "memory");
    else if constexpr (N == 3) asm volatile("global_store_dwordx3 %0, %1, %2 nt" : : "v"(off), "v"(o), "s"(base) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(off), "v"(o), "s"(base) : "memory");
}
template <int NI, int NO>
__device__ __forceinline__ void group_store(uint8_t* base, uint32_t off, const typename GroupVec<NI>::type& in, const uint32_t (&sel)[4][2])
{
    const uint32_t i0 = in[0], i1 = in[1], i2 = NI > 2 ? in[2] : 0u, i3 = NI > 3 ? in[3 < NI ? 3 : 0] : 0u;
    typename GroupVec<NO>::type o;
#pragma unroll
    for (int j = 0; j < NO; j++)
        o[j] = __builtin_amdgcn_perm(i1, i0, sel[j][0]) | (NI > 2 ? __builtin_amdgcn_perm(i3, i2, sel[j][1]) : 0u);
    group_put<NO>(base, off, o);
}

// ---- selectors of the ramped group path (xform_chunk's common case).  RampApplicator reads a subsample's top 16 bits and
// writes those two bytes back with zeros below them (Msg.cpp:840-895), so a ramped subsample is: ONE v_perm_b32 that lifts
// its two most significant bytes (either byte order) out of the loaded dwords into the top half of a register, one 24-bit
// multiply by twice the Q15 multiplier (bits 31..16 of the product are (s16 * mult) >> 15), and its share of the ONE
// v_perm_b32 per destination dword that picks those two product bytes (either byte order) and zeros. ----
constexpr uint32_t ramp_in_sel(int sb, int k, bool le)
{
    const int bi = (k * sb) & 3;                                      // first byte of the subsample in {iw[d + 1], iw[d]}, d = k * sb / 4
    const int hi = le ? bi + sb - 1 : bi, lo = le ? bi + sb - 2 : bi + 1;
    return ((uint32_t)hi << 24) | ((uint32_t)lo << 16) | 0x0c0cu;
}
constexpr uint32_t ramp_out_sel(int db, int j, bool le)               // destination dword j over the products {r[first + 1], r[first]}, first = 4j / db
{
    uint32_t sel = 0;
    const int first = (4 * j) / db;
    for (int t = 0; t < 4; t++) {
        const int B = 4 * j + t, qi = B / db, m = B % db;
        const int pos = le ? db - 1 - m : m;                           // 0 = most significant byte of the subsample
        uint32_t code = pos == 0 ? 3u : (pos == 1 ? 2u : 0x0cu);
        if (code != 0x0cu && qi != first) code += 4u;
        sel |= code << (8 * t);
    }
    return sel;
}

// SB / DB: bytes per source / destination subsample when the whole batch has one layout (immediates instead of
// scalar registers in every shift and multiply), 0 = read them from each chunk's record.
// (Every instantiation is held to 64 vector registers -- eight waves per SIMD.  With the group accesses on scalar bases the
// depth-changing ones need 57..65 by themselves; holding them costs scalar spills into vector lanes and still wins, same box:
// nothing on plain batches, 3..14 % when every message is ramped.  Round 1's 74-register S32 -> S24 lost 7 % when forced.)
template <int SB, int DB>
__global__ __launch_bounds__(kLineWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void pcm_line_kernel(const PcmChunk* __restrict__ chunks, const uint32_t n_chunks,
                                                                  const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                  const uint16_t* __restrict__ ramp_table, const uint8_t* __restrict__ prefix)
{
    __shared__ uint16_t s_ramp[kRampTableCount];
    __shared__ uint16_t s_ramp2[kRampTableCount];                       // twice the multiplier (<= 0xfffe): the ramped group path's
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kLineWaves][2][kInBytes];

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < kRampTableCount; i += blockDim.x) { s_ramp[i] = ramp_table[i]; s_ramp2[i] = (uint16_t)(2u * ramp_table[i]); }
    __syncthreads();

    // Software pipeline over the wave's chunks (chunk, chunk + stride, ...): while chunk i is processed, the input of
    // chunk i+1 is on its way into the other LDS buffer and the record of chunk i+2 is being fetched (a record is one
    // 64-byte scalar load and holds everything a chunk needs).  Returns the number of staging instructions issued.
    const uint32_t stride = gridDim.x * kLineWaves;
    uint32_t chunk = blockIdx.x * kLineWaves + wave;
    if (chunk >= n_chunks) return;
    auto is_plain = [](const PcmChunk& c) __attribute__((always_inline)) -> bool {
        return !(c.flags & (kChunkRamp | kChunkSilence)) && c.attenuation == OHGPU_UNITY_ATTENUATION;
    };
    // A chunk's prefix (a Songcast frame's header in front of the frame's first audio): 4..255 bytes of the batch's blob to just
    // before the chunk's destination, a dword per lane at any alignment; when the size is not a multiple of four the last
    // lane's dword is the prefix's LAST four bytes (it overlaps its neighbour's with the same bytes).  Scalar base + lane
    // offset on both sides: two vector registers per chunk.  The load is in flight until the caller's s_waitcnt.
    auto prefix_pos = [&](const PcmChunk& c) __attribute__((always_inline)) -> uint32_t {
        const uint32_t hb = c.prefix_bytes;
        return lane * 4 + 4 <= hb ? lane * 4 : hb - 4;
    };
    auto prefix_load = [&](const PcmChunk& c, bool on, uint32_t pos, uint32_t& v) __attribute__((always_inline)) {
        const uint8_t* const base = prefix + c.prefix_off;
        if (on && lane * 4 < c.prefix_bytes) asm volatile("global_load_dword %0, %1, %2" : "+v"(v) : "v"(pos), "s"(base) : "memory");
    };
    auto prefix_store = [&](const PcmChunk& c, bool on, uint32_t pos, uint32_t v) __attribute__((always_inline)) {
        uint8_t* const base = dst + (c.dst_off - c.prefix_bytes);
        if (on && lane * 4 < c.prefix_bytes) asm volatile("global_store_dword %0, %1, %2" : : "v"(pos), "v"(v), "s"(base) : "memory");
    };
    auto stage_in = [&](const PcmChunk& c, uint32_t buf) __attribute__((always_inline)) -> uint32_t {
        if (c.flags & kChunkSilence) return 0u;
        if (SB != 0) return 0u;                                        // a uniform batch works in registers (group paths): nothing to stage
        const uint32_t head = (uint32_t)((uint64_t)(uintptr_t)src + c.src_off) & 15u;        // first source byte's place in its piece
        const uint8_t* base = src + (c.src_off - head);
        const uint32_t n_pieces = (head + c.nq * (SB ? SB : c.sb) + 15u) >> 4;
        uint32_t issued = 0;
        for (uint32_t p0 = 0; p0 < n_pieces; p0 += 64, issued++) {
            if (p0 + lane < n_pieces)                                   // LDS destination = wave-uniform base + lane * 16
                __builtin_amdgcn_global_load_lds((global_ptr_t)(base + (size_t)(p0 + lane) * 16),
                                                 (lds_ptr_t)(&s_in[wave][buf][0] + p0 * 16), 16, 0, 0);
        }
        return issued;
    };
    // ---- one chunk through the general path: lane = aligned destination dword.  `in`: the chunk's staged source bytes (the
    // aligned 16-byte pieces that cover them), not read for a silent chunk ----
    auto generic_chunk = [&](const PcmChunk& ck, const __attribute__((address_space(3))) uint8_t* in) __attribute__((always_inline)) {
        const uint32_t ch = ck.channels, sb = SB ? SB : ck.sb, db = DB ? DB : ck.db;
        const bool src_le = (ck.flags & kChunkSrcLe) != 0;
        const bool dst_le = (ck.flags & kChunkDstLe) != 0;
        const bool ramp = (ck.flags & kChunkRamp) != 0;
        const bool silence = (ck.flags & kChunkSilence) != 0;
        const bool zero_lsb = (ck.flags & kChunkZeroLsb) != 0;
        const bool atten = ck.attenuation != OHGPU_UNITY_ATTENUATION;
        const uint32_t q0 = ck.q0, nq = ck.nq;
        const uint32_t head = (uint32_t)((uint64_t)(uintptr_t)src + ck.src_off) & 15u;

        // ---- lane = aligned destination dword: transform the 1..4 subsamples whose bytes fall into it, funnel, store ----
        const int32_t total = (int32_t)((uint32_t)ck.ramp_start - (uint32_t)ck.ramp_end);
        const uint32_t keep = db == 4 ? (zero_lsb ? 0xffffff00u : 0xffffffffu) : ~(0xffffffffu >> (8 * db));   // the db top bytes
        // subsample q of the chunk -> its destination bytes in memory order, first byte lowest (0 past the end)
        auto subsample = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t {
            if (q >= nq) return 0u;
            uint32_t w;
            const uint32_t sub = q0 + q;                                // subsample index inside the message
            if (silence) {
                w = silence_word((uint64_t)sub * sb, sb, ch);
            } else {
                const uint32_t off = head + q * sb;
                const __attribute__((address_space(3))) uint32_t* a = (const __attribute__((address_space(3))) uint32_t*)(in + (off & ~3u));
                const uint32_t raw = __builtin_amdgcn_alignbyte(a[1], a[0], off & 3u);      // the 4 bytes that start at the subsample
                // left-justified big-endian word (load_be_word): BE bytes are already in order, LE bytes reversed
                w = src_le ? (raw << (32 - 8 * sb)) : (__builtin_bswap32(raw) & ~(sb == 4 ? 0u : (0xffffffffu >> (8 * sb))));
                if (atten) w = attenuate_word(w, ck.attenuation);
                if (ramp) {
                    const uint32_t frame = udiv_magic(sub, ck.m_ch, ck.s_ch);
                    const uint32_t mult = s_ramp[ramp_index_magic(ck.ramp_start, total, frame, ck.n_frames, ck.m_n1, ck.s_n1)];
                    w = ramp_word(w, mult, sb, ch, sub - frame * ch);
                }
            }
            w &= keep;                                                  // store_word: the db top bytes ...
            return dst_le ? (w >> (32 - 8 * db)) : __builtin_bswap32(w);   // ... in the requested byte order
        };
        // The common case -- no attenuation, ramp or silence -- is a fixed byte shuffle of the four bytes that start at
        // the subsample: ONE v_perm_b32 with the selector the host put into the record (depth conversion, both byte
        // orders, the zeroed low byte of 32-bit output).  Subsamples past the chunk's end read bytes that exist in the
        // staging buffer and only ever land in byte positions the edge path does not store.
        auto subsample_plain = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t {
            const uint32_t off = head + __umul24(q, sb);   // q <= 514, sb <= 4
            const __attribute__((address_space(3))) uint32_t* a = (const __attribute__((address_space(3))) uint32_t*)(in + (off & ~3u));
            return __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(a[1], a[0], off & 3u), ck.plain_sel);
        };
        const uint32_t dhead = (uint32_t)((uint64_t)(uintptr_t)dst + ck.dst_off) & 3u;         // first destination byte's place in its dword
        const uint32_t len = nq * db;
        const uint32_t n_dw = (dhead + len + 3u) >> 2;
        uint8_t* const obase = dst + (ck.dst_off - dhead);             // (pointer arithmetic on the argument keeps the stores global, not flat)
        auto emit = [&](auto&& sub_fn) __attribute__((always_inline)) {
            for (uint32_t kb = 0; kb < n_dw; kb += 64) {
                const uint32_t k = kb + lane;
                if (k < n_dw) {
                    const int32_t pos = (int32_t)(k * 4) - (int32_t)dhead;  // stream position of the dword's first byte
                    const uint32_t upos = pos < 0 ? 0u : (uint32_t)pos;
                    uint32_t v;                                             // stream bytes [upos, upos + 4)
                    if (db == 3) {
                        const uint32_t qa = __umulhi(upos, 0xAAAAAAABu) >> 1, o = upos - __umul24(qa, 3u);
                        const uint32_t v0 = sub_fn(qa), v1 = sub_fn(qa + 1);
                        // bytes {v0: 0..3, v1: 4..7}; stream = v0.b0 v0.b1 v0.b2 v1.b0 v1.b1 v1.b2
                        const uint32_t sel = o == 0 ? 0x04020100u : (o == 1 ? 0x05040201u : 0x06050402u);
                        v = __builtin_amdgcn_perm(v1, v0, sel);
                    } else if (db == 4) {
                        const uint32_t qa = upos >> 2, o = upos & 3;
                        v = sub_fn(qa);
                        if (dhead != 0) v = __builtin_amdgcn_alignbyte(sub_fn(qa + 1), v, o);
                    } else if (db == 2) {
                        const uint32_t qa = upos >> 1, o = upos & 1;
                        v = sub_fn(qa) | (sub_fn(qa + 1) << 16);
                        if (dhead & 1) v = __builtin_amdgcn_alignbyte(sub_fn(qa + 2), v, o);
                    } else {
                        v = sub_fn(upos) | (sub_fn(upos + 1) << 8) | (sub_fn(upos + 2) << 16) | (sub_fn(upos + 3) << 24);
                    }
                    if (pos >= 0 && (uint32_t)pos + 4 <= len) {
                        __builtin_nontemporal_store(v, (uint32_t*)(obase + (size_t)k * 4));   // written once, never read here
                    } else {                                                // first / last dword of the chunk: only its own bytes
                        const uint32_t skip = pos < 0 ? (uint32_t)(-pos) : 0u;   // v starts at stream byte upos = pos + skip
                        for (uint32_t b = skip; b < 4; b++) {
                            if ((uint32_t)(pos + (int32_t)b) < len) obase[(size_t)k * 4 + b] = (uint8_t)(v >> (8 * (b - skip)));
                        }
                    }
                }
            }
        };
        if (!(ramp || silence || atten)) emit(subsample_plain); else emit(subsample);
    };
    if constexpr (SB != 0 && DB != 0) {
        // ================= uniform batch: registers only, TWO chunks per trip =================
        // Lane = GROUP of four subsamples: 4*SB source bytes (one unaligned 8/12/16-byte load: unaligned wide accesses run at
        // full rate, tools/micro/unaligned_store.hip) -> 4*DB destination bytes (one store).  A chunk is a message (or a run of
        // plain messages): at most a few hundred groups, so a wave that takes one chunk at a time has two loads per lane in
        // flight and then waits out the whole HBM latency -- round 1's kernel moved 24 KB per CU per latency, 3 TB/s of reads.
        // Every trip now loads the first 128 groups of BOTH its chunks before it touches either, and the records of the next
        // trip's chunks are fetched before this trip's are processed.  (Measured and dropped, same box: pulling the next trip's
        // lines into the L2 ahead of time through the LDS DMA path, -7 % on ramped stereo; computing every group's ramp
        // multipliers -- two or three per group whatever the channel count -- before the wait, -14 %: both cost registers,
        // and with them waves.)
        typedef typename GroupVec<SB>::type Vec;
        uint32_t grp_sel[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};    // plain path: selector of destination dword j over source dwords {2pr+1, 2pr}
        uint32_t grp_for = 0xffffffffu;                               // the per-subsample selector they were derived from
        auto is_silence = [](const PcmChunk& c) __attribute__((always_inline)) -> bool { return (c.flags & kChunkSilence) != 0; };
        auto head_of = [&](uint32_t i) __attribute__((always_inline)) -> PcmChunkHead { return *(const PcmChunkHead*)&chunks[i]; };
        auto load_head = [&](const PcmChunkHead& c, bool on, Vec& a, Vec& b) __attribute__((always_inline)) {
            if (!on || (c.flags & kChunkSilence)) return;
            const uint32_t n_grp = c.nq >> 2;
            const uint8_t* const sp = src + c.src_off;
            if (lane < n_grp) group_load<SB>(a, sp, lane * (4 * SB));
            if (64 + lane < n_grp) group_load<SB>(b, sp, (64 + lane) * (4 * SB));
        };
        // ---- plain chunk: a fixed byte shuffle (per destination dword: one v_perm_b32 per pair of source dwords, selectors
        // derived from the record's per-subsample selector when the layout changes) ----
        auto plain_chunk = [&](const PcmChunk& ck, Vec& in_a, Vec& in_b) __attribute__((always_inline)) {
            if (ck.plain_sel != grp_for) {
                grp_for = ck.plain_sel;
#pragma unroll
                for (int j = 0; j < DB; j++)
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {
                        uint32_t sel = 0;
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int B = 4 * j + t, qi = B / DB, m = B % DB;
                            const uint32_t sbyte = (ck.plain_sel >> (8 * m)) & 0xffu;
                            const uint32_t sidx = (uint32_t)(qi * SB) + sbyte;              // source byte of the group
                            uint32_t code = 0x0c;                                            // zero
                            if (sbyte != 0x0c && (sidx >> 3) == (uint32_t)pr) code = sidx & 7u;   // byte of {I[2pr+1], I[2pr]}
                            sel |= code << (8 * t);
                        }
                        grp_sel[j][pr] = sel;
                    }
            }
            const uint32_t n_grp = ck.nq >> 2;
            const uint8_t* const sp = src + ck.src_off;
            uint8_t* const dp = dst + ck.dst_off;
            for (uint32_t g0 = 0; g0 < n_grp; g0 += 128) {
                const uint32_t ga = g0 + lane, gb = g0 + 64 + lane;
                if (g0 != 0) {                                          // (the first 128 groups were loaded by the trip)
                    if (ga < n_grp) group_load<SB>(in_a, sp, ga * (4 * SB));
                    if (gb < n_grp) group_load<SB>(in_b, sp, gb * (4 * SB));
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(in_a), "+v"(in_b) : : "memory");
                }
                if (ga < n_grp) group_store<SB, DB>(dp, ga * (4 * DB), in_a, grp_sel);
                if (gb < n_grp) group_store<SB, DB>(dp, gb * (4 * DB), in_b, grp_sel);
            }
            // the chunk's last 1..3 subsamples: byte by byte
            const uint32_t tail0 = n_grp * 4, tail_bytes = (ck.nq - tail0) * DB;
            if (lane < tail_bytes) {
                const uint32_t q = tail0 + lane / DB, m = lane % DB;
                const uint32_t sbyte = (ck.plain_sel >> (8 * m)) & 0xffu;
                dp[(size_t)q * DB + m] = sbyte == 0x0c ? (uint8_t)0 : sp[(size_t)q * SB + sbyte];
            }
        };
        // ---- attenuated and / or ramped chunk: the same groups of four subsamples, each subsample taken out of the loaded
        // registers (static positions), run through pcm_device.h's expressions and put back at its (static) place in the
        // destination registers ----
        auto xform_chunk = [&](const PcmChunk& ck, Vec& in_a, Vec& in_b) __attribute__((always_inline)) {
            const bool src_le = (ck.flags & kChunkSrcLe) != 0, dst_le = (ck.flags & kChunkDstLe) != 0;
            const bool ramp = (ck.flags & kChunkRamp) != 0, atten = ck.attenuation != OHGPU_UNITY_ATTENUATION;
            const uint32_t keep = DB == 4 ? ((ck.flags & kChunkZeroLsb) ? 0xffffff00u : 0xffffffffu) : ~(0xffffffffu >> (8 * (DB & 3)));
            const int32_t total = (int32_t)((uint32_t)ck.ramp_start - (uint32_t)ck.ramp_end);
            auto transform = [&](uint32_t raw, uint32_t sub) __attribute__((always_inline)) -> uint32_t {
                uint32_t w = src_le ? (raw << (32 - 8 * SB)) : (__builtin_bswap32(raw) & ~(SB == 4 ? 0u : (0xffffffffu >> (8 * (SB & 3)))));
                if (atten) w = attenuate_word(w, ck.attenuation);
                if (ramp) {
                    const uint32_t frame = udiv_magic(sub, ck.m_ch, ck.s_ch);
                    const uint32_t mult = s_ramp[ramp_index_magic(ck.ramp_start, total, frame, ck.n_frames, ck.m_n1, ck.s_n1)];
                    w = ramp_word(w, mult, SB, ck.channels, sub - frame * ck.channels);
                }
                w &= keep;
                return dst_le ? (w >> (32 - 8 * DB)) : __builtin_bswap32(w);      // destination bytes in memory order, first byte low
            };
            const uint32_t n_grp = ck.nq >> 2;
            const uint8_t* const sp = src + ck.src_off;
            uint8_t* const dp = dst + ck.dst_off;
            const bool stereo_even = ck.channels == 2 && (ck.q0 & 1) == 0;     // a group = two whole frames: two ramp look-ups, not four
            auto ramp_mult = [&](uint32_t frame) __attribute__((always_inline)) -> uint32_t {
                return s_ramp[ramp_index_magic(ck.ramp_start, total, frame, ck.n_frames, ck.m_n1, ck.s_n1)];
            };
            auto do_group = [&](uint32_t g, const Vec& in) __attribute__((always_inline)) {
                uint32_t iw[5] = {in[0], in[1], SB > 2 ? in[2 < SB ? 2 : 0] : 0u, SB > 3 ? in[3 < SB ? 3 : 0] : 0u, 0u};
                uint32_t ow[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int off = k * SB, o = k * DB;
                    const uint32_t raw = (off & 3) ? __builtin_amdgcn_alignbyte(iw[(off >> 2) + 1], iw[off >> 2], off & 3) : iw[off >> 2];
                    uint32_t w = src_le ? (raw << (32 - 8 * SB)) : (__builtin_bswap32(raw) & ~(SB == 4 ? 0u : (0xffffffffu >> (8 * (SB & 3)))));
                    if (atten) w = attenuate_word(w, ck.attenuation);
                    if (ramp) {
                        const uint32_t sub = ck.q0 + 4 * g + k, frame = udiv_magic(sub, ck.m_ch, ck.s_ch);
                        w = ramp_word(w, ramp_mult(frame), SB, ck.channels, sub - frame * ck.channels);
                    }
                    w &= keep;
                    const uint32_t v = dst_le ? (w >> (32 - 8 * DB)) : __builtin_bswap32(w);   // destination bytes in memory order
                    ow[o >> 2] |= v << (8 * (o & 3));
                    if ((o & 3) + DB > 4) ow[(o >> 2) + 1] |= v >> (32 - 8 * (o & 3));
                }
                typename GroupVec<DB>::type out;
#pragma unroll
                for (int j = 0; j < DB; j++) out[j] = ow[j];
                group_put<DB>(dp, g * (4 * DB), out);
            };
            // The common ramped chunk -- no attenuation, two bytes or more either side, a group inside two frames (stereo from an
            // even subsample: frames f, f + 1; three channels or more: frame f up to subsample kb, f + 1 from it) -- without the
            // general expressions: see ramp_in_sel.  (32-bit six-channel audio gets its channel byte from the general path.)
            // ramp_index_magic for a message of 3..32768 frames (no special cases: the multiplier exists and frame x |total| is
            // below 2^31; the quotient's sign is the ramp's, the same for the whole message), then TWICE the Q15 multiplier
            const uint32_t abs_total = (uint32_t)(total < 0 ? -total : total) & 0x1ffffu;
            const uint32_t neg_mask = total < 0 ? 0xffffffffu : 0u, ramp_base = (uint32_t)ck.ramp_start + neg_mask;   // start - (+-mag) = (start + m) - (mag ^ m)
            auto ramp_mult2 = [&](uint32_t frame) __attribute__((always_inline)) -> uint32_t {
                const uint32_t mag = __umulhi((frame & 0xffffu) * abs_total, ck.m_n1) >> ck.s_n1;     // (masks: operand ranges the compiler can see select the full-rate 24-bit multiplies)
                const uint32_t ramp16 = (ramp_base - (mag ^ neg_mask)) & 0xffffu;
                const uint32_t idx = (kRampMax + (1u << 4) - ramp16) >> 5;
                return s_ramp2[idx < kRampTableCount - 1 ? idx : kRampTableCount - 1];
            };
            auto ramped_group = [&](auto stereo_tag, uint32_t g, const Vec& in) __attribute__((always_inline)) {
                constexpr bool STEREO = decltype(stereo_tag)::value;
                constexpr int S = SB < 2 ? 2 : SB, D = DB < 2 ? 2 : DB;       // (never instantiated for one-byte subsamples)
                const uint32_t iw[5] = {in[0], in[1], SB > 2 ? in[2 < SB ? 2 : 0] : 0u, SB > 3 ? in[3 < SB ? 3 : 0] : 0u, 0u};
                const uint32_t sub0 = ck.q0 + 4 * g;
                uint32_t f0, kb = 2;
                if constexpr (STEREO) {
                    f0 = sub0 >> 1;
                } else {
                    f0 = __umulhi(sub0, ck.m_ch) >> ck.s_ch;              // (three channels or more: the multiplier exists)
                    kb = ck.channels - (sub0 - (f0 & 0xffffu) * (uint32_t)ck.channels);
                }
                const uint32_t m0 = ramp_mult2(f0), m1 = ramp_mult2(f0 + 1);
                uint32_t r[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = (k * S) >> 2;
                    const uint32_t w = __builtin_amdgcn_perm(iw[d + 1], iw[d], src_le ? ramp_in_sel(S, k, true) : ramp_in_sel(S, k, false));
                    const uint32_t mk = STEREO ? (k < 2 ? m0 : m1) : ((uint32_t)k >= kb ? m1 : m0);
                    r[k] = (uint32_t)(((int32_t)w >> 16) * (int32_t)mk);
                }
                typename GroupVec<DB>::type out;
#pragma unroll
                for (int j = 0; j < DB; j++) {
                    const int first = (4 * j) / D;
                    out[j] = __builtin_amdgcn_perm(r[first + 1], r[first], dst_le ? ramp_out_sel(D, j, true) : ramp_out_sel(D, j, false));
                }
                group_put<DB>(dp, g * (4 * DB), out);
            };
            auto all_groups = [&](auto&& one) __attribute__((always_inline)) {
                for (uint32_t g0 = 0; g0 < n_grp; g0 += 128) {
                    const uint32_t ga = g0 + lane, gb = g0 + 64 + lane;
                    if (g0 != 0) {                                      // (the first 128 groups were loaded by the trip)
                        if (ga < n_grp) group_load<SB>(in_a, sp, ga * (4 * SB));
                        if (gb < n_grp) group_load<SB>(in_b, sp, gb * (4 * SB));
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(in_a), "+v"(in_b) : : "memory");
                    }
                    if (ga < n_grp) one(ga, in_a);
                    if (gb < n_grp) one(gb, in_b);
                }
            };
            const bool two_frames = stereo_even || ck.channels >= 3;
            if (SB >= 2 && DB >= 2 && ramp && !atten && two_frames && !(SB == 4 && ck.channels == 6) && ck.m_n1 != 0 && ck.n_frames <= 32768u) {
                if (stereo_even) all_groups([&](uint32_t g, const Vec& in) __attribute__((always_inline)) { ramped_group(std::true_type{}, g, in); });
                else all_groups([&](uint32_t g, const Vec& in) __attribute__((always_inline)) { ramped_group(std::false_type{}, g, in); });
            } else {
                all_groups(do_group);
            }
            const uint32_t tail0 = n_grp * 4;                   // the chunk's last 1..3 subsamples, one lane each
            if (lane < ck.nq - tail0) {
                const uint32_t q = tail0 + lane;
                uint32_t raw = 0;
#pragma unroll
                for (int bq = 0; bq < SB; bq++) raw |= (uint32_t)sp[(size_t)q * SB + bq] << (8 * bq);
                const uint32_t v = transform(raw, ck.q0 + q);
#pragma unroll
                for (int bq = 0; bq < DB; bq++) dp[(size_t)q * DB + bq] = (uint8_t)(v >> (8 * bq));
            }
        };
        // (Scalar registers are the scarce thing here: two whole records per chunk in flight -- this trip's and the next one's --
        // were 64 of them and spilled into vector lanes inside the loops.  A trip needs 16 bytes of a record to issue its
        // loads; those are fetched one trip ahead, the whole records while the audio is on its way.)
        // (a trip's two chunks are NEIGHBOURS in the list -- usually in memory too: the line two messages or two datagrams share
        // is then written by one wave within one trip, not by two waves at different times)
        chunk *= 2;
        if (chunk >= n_chunks) return;
        PcmChunkHead h0 = head_of(chunk);
        bool has1 = chunk + 1 < n_chunks;
        PcmChunkHead h1 = head_of(has1 ? chunk + 1 : chunk);
        while (true) {
            Vec a0 = {}, b0 = {}, a1 = {}, b1 = {};
            load_head(h0, true, a0, b0);
            load_head(h1, has1, a1, b1);
            const PcmChunk c0 = chunks[chunk];
            const PcmChunk c1 = chunks[has1 ? chunk + 1 : chunk];
            const uint32_t next = chunk + 2 * stride;
            const bool more = next < n_chunks, more1 = next + 1 < n_chunks;
            const PcmChunkHead n0 = head_of(more ? next : chunk);
            const PcmChunkHead n1 = head_of(more1 ? next + 1 : chunk);
            uint32_t p0 = 0, p1 = 0;                                    // the chunks' prefixes (none: no lane takes part)
            const uint32_t pp0 = prefix_pos(c0), pp1 = prefix_pos(c1);
            prefix_load(c0, true, pp0, p0);
            prefix_load(c1, has1, pp1, p1);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(p0), "+v"(p1) : : "memory");
            prefix_store(c0, true, pp0, p0);
            prefix_store(c1, has1, pp1, p1);
            if (is_silence(c0)) generic_chunk(c0, nullptr);
            else if (is_plain(c0)) plain_chunk(c0, a0, b0);
            else xform_chunk(c0, a0, b0);
            if (has1) {
                if (is_silence(c1)) generic_chunk(c1, nullptr);
                else if (is_plain(c1)) plain_chunk(c1, a1, b1);
                else xform_chunk(c1, a1, b1);
            }
            if (!more) break;
            chunk = next;
            h0 = n0; h1 = n1; has1 = more1;
        }
        return;
    }
    PcmChunk ck = chunks[chunk];
    PcmChunk nx = ck;
    bool has_nx = chunk + stride < n_chunks;
    if (has_nx) nx = chunks[chunk + stride];
    stage_in(ck, 0);
    uint32_t buf = 0;
    while (true) {
        // fetch the record after next, start the next chunk's input, then wait for this chunk's input only
        const bool has_nn = has_nx && chunk + 2 * stride < n_chunks;
        PcmChunk nn = nx;
        if (has_nn) nn = chunks[chunk + 2 * stride];
        const uint32_t k_nx = has_nx ? stage_in(nx, buf ^ 1) : 0u;
        if (k_nx == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (k_nx == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (k_nx == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (ck.prefix_bytes) {
            uint32_t pv = 0;
            const uint32_t pp = prefix_pos(ck);
            prefix_load(ck, true, pp, pv);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv) : : "memory");
            prefix_store(ck, true, pp, pv);
        }
        generic_chunk(ck, (const __attribute__((address_space(3))) uint8_t*)&s_in[wave][buf][0]);
        if (!has_nx) break;
        chunk += stride;
        ck = nx; nx = nn; has_nx = has_nn;
        buf ^= 1;
    }
}

// ---- host side: chunk list and per-message multipliers ----
void free_pcm_line(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    if (b->line.d_chunks) ctx_dev_free(ctx, b->line.d_chunks);
    if (b->line.d_prefix) ctx_dev_free(ctx, b->line.d_prefix);
    b->line = PcmLinePlan();
}

int plan_pcm_line(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_msg_desc* descs, size_t n,
                  const MsgPrefix* prefixes, const uint8_t* blob, size_t blob_bytes)
{
    b->line = PcmLinePlan();
    // How many of a stream's consecutive plain messages share a chunk: enough chunks for every wave of the launch to take a few
    // dozen (the launch ends on whole chunks), no more than eight messages each (a chunk's record is read once; one message
    // per chunk made 512 000 stereo messages 6 % slower than two, eight per chunk 5 % slower -- tools/bench_pcm.py, same box).
    size_t n_plain = 0;
    for (size_t i = 0; i < n; i++)
        n_plain += !(descs[i].flags & (OHGPU_FLAG_RAMP | OHGPU_FLAG_SILENCE)) && descs[i].attenuation == OHGPU_UNITY_ATTENUATION;
    const size_t waves = (size_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * 8 * kLineWaves;
    const uint32_t merge_msgs = (uint32_t)std::min<size_t>(8, std::max<size_t>(1, (n_plain + 16 * waves) / (32 * waves)));
    // One chunk list per layout: messages of 16/24/32-bit audio on both sides go to the list of their (source, destination)
    // depth pair and run the instantiation that has those depths as immediates and works in registers (whatever their channel
    // counts and byte orders: those are per chunk); everything else -- 8-bit audio on either side -- to list 0, the general
    // staged path.  A batch is as many launches as it has non-empty lists: one 8-bit stream no longer sends 2047 others down
    // the slow path.
    std::vector<PcmChunk> lists[kLineLists];
    std::vector<PcmChunk>* last_list = nullptr;                         // the previous message's
    bool mergeable = false;                                             // ... and its last chunk is a group-path chunk
    for (size_t i = 0; i < n; i++) {
        const ohgpu_msg_desc& d = descs[i];
        const uint64_t n_sub = (uint64_t)d.n_frames * d.channels;
        if (n_sub > 0xffffffffull) return OHGPU_OK;                     // the generic kernel takes such a batch
        if ((d.flags & OHGPU_FLAG_RAMP) && n_sub >= (1ull << 31)) return OHGPU_OK;
        PcmChunk c;
        memset(&c, 0, sizeof(c));
        const uint32_t sb = d.src_bits / 8, db = d.dst_bits / 8;
        c.n_frames = d.n_frames;
        c.ramp_start = d.ramp_start; c.ramp_end = d.ramp_end;
        c.attenuation = d.attenuation;
        c.channels = d.channels; c.sb = (uint8_t)sb; c.db = (uint8_t)db;
        c.flags = (uint8_t)(((d.flags & OHGPU_FLAG_RAMP) ? kChunkRamp : 0) | ((d.flags & OHGPU_FLAG_SILENCE) ? kChunkSilence : 0) |
                            (((d.flags & OHGPU_FLAG_ZERO_LSB32) && db == 4) ? kChunkZeroLsb : 0) |
                            ((d.src_endian == OHGPU_ENDIAN_LITTLE && sb > 1) ? kChunkSrcLe : 0) |
                            ((d.dst_endian == OHGPU_ENDIAN_LITTLE) ? kChunkDstLe : 0));
        // plain path: memory byte m of the destination subsample <- byte of the 4 source bytes (0x0c = zero)
        for (uint32_t m = 0; m < 4; m++) {
            uint32_t selb = 0x0c;
            if (m < db) {
                const uint32_t j = (c.flags & kChunkDstLe) ? db - 1 - m : m;               // index from the most significant byte
                if (j < sb && !((c.flags & kChunkZeroLsb) && j == 3)) selb = (c.flags & kChunkSrcLe) ? sb - 1 - j : j;
            }
            c.plain_sel |= selb << (8 * m);
        }
        const bool has_prefix = prefixes && prefixes[i].bytes != 0 && n_sub > 0;     // (goes with the message's FIRST chunk)
        if (has_prefix) {
            if (prefixes[i].bytes < 4u || prefixes[i].bytes > 255u || (prefixes[i].off & 3u) || (uint64_t)prefixes[i].off + ((prefixes[i].bytes + 3u) & ~3u) > blob_bytes ||
                prefixes[i].bytes > d.dst_offset)
                return set_error(OHGPU_ERR_INVALID, "message %zu: prefix [%u, +%u) of a %zu-byte blob", i, prefixes[i].off, prefixes[i].bytes, blob_bytes);
            c.prefix_off = prefixes[i].off; c.prefix_bytes = (uint8_t)prefixes[i].bytes;
        }
        uint32_t sh;
        magic_u31(d.channels, &c.m_ch, &sh); c.s_ch = (uint8_t)sh;
        magic_u31(d.n_frames > 1 ? d.n_frames - 1 : 1, &c.m_n1, &sh); c.s_n1 = (uint8_t)sh;
        // Plain messages of a uniform batch run the group path, which needs no staging buffer: such a message is one chunk,
        // and it is appended to the previous chunk when it continues it in both arenas (a stream's consecutive messages).
        const bool fast_layout = sb >= 2 && sb <= 4 && db >= 2 && db <= 4;
        std::vector<PcmChunk>& chunks = lists[fast_layout ? 1 + (sb - 2) * 3 + (db - 2) : 0];
        if (&chunks != last_list) mergeable = false;
        last_list = &chunks;
        const bool registers_only = fast_layout && !(d.flags & OHGPU_FLAG_SILENCE);
        const bool group_path = registers_only && !(d.flags & OHGPU_FLAG_RAMP) && d.attenuation == OHGPU_UNITY_ATTENUATION;
        if (registers_only && !group_path && n_sub > 0) {                // ramped / attenuated: one chunk per message, no staging
            c.q0 = 0; c.nq = (uint32_t)n_sub; c.src_off = d.src_offset; c.dst_off = d.dst_offset;
            chunks.push_back(c);
            mergeable = false;
            continue;
        }
        if (group_path && n_sub > 0) {
            // A stream's consecutive plain messages are one run of bytes: they are appended to the open chunk, and the run is cut
            // into chunks of about merge_msgs messages -- NOT at the message boundaries but where the destination is 128-byte
            // aligned (and, if possible, a whole number of groups in): a boundary inside a line makes two waves write parts of
            // it at different times.  48 kHz messages (1440 bytes of S24 stereo) end on 64-byte boundaries anyway; 44.1 kHz
            // ones (1320 bytes) do not, and ran 15 % slower until the cuts moved (64-byte cuts: +13 %, 128-byte: another 1..4 %).
            const uint32_t target = (uint32_t)std::min<uint64_t>(kGroupChunkSub, (uint64_t)merge_msgs * n_sub);
            bool appended = false;
            if (!chunks.empty() && mergeable && !has_prefix) {
                PcmChunk& p = chunks.back();
                if (p.flags == c.flags && p.plain_sel == c.plain_sel && (uint64_t)p.nq + n_sub <= 0xffffffffull &&
                    p.src_off + (uint64_t)p.nq * sb == d.src_offset && p.dst_off + (uint64_t)p.nq * db == d.dst_offset) {
                    p.nq += (uint32_t)n_sub;
                    appended = true;
                }
            }
            if (!appended) {
                c.q0 = 0; c.nq = (uint32_t)n_sub; c.src_off = d.src_offset; c.dst_off = d.dst_offset;
                chunks.push_back(c);
                mergeable = true;
            }
            while (chunks.back().nq >= target + target / 4) {           // (what stays open is less than a chunk and a quarter)
                PcmChunk rest = chunks.back();
                uint32_t cut = target & ~3u, second_best = 0;
                for (uint32_t q = target; q + 256 > target && q > target / 2; q--) {
                    if ((rest.dst_off + (uint64_t)q * db) % 128 != 0) continue;
                    if (q % 4 == 0) { second_best = q; break; }
                    if (!second_best) second_best = q;
                }
                if (second_best) cut = second_best;
                if (cut == 0) break;
                chunks.back().nq = cut;
                rest.src_off += (uint64_t)cut * sb; rest.dst_off += (uint64_t)cut * db; rest.nq -= cut; rest.prefix_bytes = 0;
                chunks.push_back(rest);
            }
            continue;
        }
        mergeable = false;
        for (uint64_t q0 = 0; q0 < n_sub; q0 += kChunkSub) {
            c.q0 = (uint32_t)q0;
            c.nq = (uint32_t)(n_sub - q0 < kChunkSub ? n_sub - q0 : kChunkSub);
            c.src_off = d.src_offset + q0 * sb;
            c.dst_off = d.dst_offset + q0 * db;
            chunks.push_back(c);
            c.prefix_bytes = 0;
        }
    }
    std::vector<PcmChunk> all;
    for (uint32_t k = 0; k < kLineLists; k++) {
        b->line.list_first[k] = (uint32_t)all.size();
        b->line.list_count[k] = (uint32_t)lists[k].size();
        // (how much of the list is arithmetic -- ramped or attenuated chunks -- by subsamples: decides the launch's occupancy)
        uint64_t sub_all = 0, sub_heavy = 0;
        for (const PcmChunk& c : lists[k]) {
            sub_all += c.nq;
            if ((c.flags & kChunkRamp) || c.attenuation != OHGPU_UNITY_ATTENUATION) sub_heavy += c.nq;
        }
        b->line.list_heavy[k] = sub_all ? (uint8_t)((sub_heavy * 100 + sub_all / 2) / sub_all) : 0;
        all.insert(all.end(), lists[k].begin(), lists[k].end());
        if (all.size() > 0xffffffffull) return OHGPU_OK;
    }
    if (all.empty()) return OHGPU_OK;
    hipError_t e = ctx_dev_alloc(ctx, &b->line.d_chunks, all.size() * sizeof(PcmChunk));
    if (e == hipSuccess) e = hipMemcpy(b->line.d_chunks, all.data(), all.size() * sizeof(PcmChunk), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        free_pcm_line(ctx, b);
        return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "chunk plan upload: %s", hipGetErrorString(e));
    }
    if (prefixes && blob_bytes) {
        e = ctx_dev_alloc(ctx, &b->line.d_prefix, blob_bytes);
        if (e == hipSuccess) e = hipMemcpy(b->line.d_prefix, blob, blob_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            free_pcm_line(ctx, b);
            return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "prefix blob upload: %s", hipGetErrorString(e));
        }
        b->line.prefixed = true;
    }
    b->line.n_chunks = (uint32_t)all.size();
    b->line.enabled = true;
    return OHGPU_OK;
}

template <int SB, int DB>
static hipError_t launch_line(const ohgpu_ctx* ctx, const ohgpu_batch* b, uint32_t first, uint32_t count, uint32_t heavy_percent, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t grid = (count + kLineWaves - 1) / kLineWaves;
    // Workgroups per CU.  A launch of byte shuffles is a streaming copy, and the memory system serves a copy best from FEW, long-lived
    // workgroups (tools/micro/run_copy.hip: the same bytes at 5.6 TB/s from two to four workgroups per CU, 4.6-4.9 from eight to
    // thirty-two): four per CU took the default mix (5 % of the messages ramped) from 0.2877 to 0.2723 ms, same box, turn and turn
    // about (three: 0.2872, two: 0.366).  A launch whose chunks are mostly ramped or attenuated has arithmetic to hide behind other
    // waves' loads: every message ramped 0.2949 at eight per CU, 0.2877 at six, 0.3167 at four.  And four is the better number only
    // for the LARGE plain launch: by chunks (same box, gpurun_out/r5/exp_line3.log, exp_line4.log, ms at 4 / 5 / 6 / 8 per CU) 128 k
    // 0.0631 / 0.0603 / 0.0613 / 0.0616; 256 k 0.158 / 0.143 / 0.135 / 0.135 -- Songcast frames of 256 k stereo messages 0.164 /
    // - / 0.150 / 0.152, the sender's S32 -> S24 pack 0.168 / - / 0.156 / 0.160 --; 512 k 0.261 / 0.262 / 0.265 / 0.279, S32 -> S24
    // 0.310 / 0.312 / 0.323 / 0.335; 1 M within the noise.  Round 5's profiles of the Songcast and sender-pack benches are what
    // showed it (6 % behind round 3's on a launch size the occupancy had not been tried on).
#ifdef OHGPU_LINE_GROUPS_PER_CU
    const uint32_t per_cu = OHGPU_LINE_GROUPS_PER_CU;
#else
    const uint32_t per_cu = heavy_percent >= 30u || count < 400000u ? 6u : 4u;
#endif
    if (grid > cus * per_cu) grid = cus * per_cu;
    hipLaunchKernelGGL((pcm_line_kernel<SB, DB>), dim3(grid), dim3(kLineWaves * 64), 0, s,
                       (const PcmChunk*)b->line.d_chunks + first, count, src, dst, ctx->d_ramp_table, (const uint8_t*)b->line.d_prefix);
    return hipGetLastError();
}

hipError_t launch_pcm_line(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    for (uint32_t k = 0; k < kLineLists; k++) {
        const uint32_t first = b->line.list_first[k], count = b->line.list_count[k];
        if (count == 0) continue;
        hipError_t e = hipSuccess;
        switch (k) {
        case 0: e = launch_line<0, 0>(ctx, b, first, count, b->line.list_heavy[k], src, dst, s); break;
#define X(S, D) case 1 + (S - 2) * 3 + (D - 2): e = launch_line<S, D>(ctx, b, first, count, b->line.list_heavy[k], src, dst, s); break;
        X(2, 2) X(2, 3) X(2, 4) X(3, 2) X(3, 3) X(3, 4) X(4, 2) X(4, 3) X(4, 4)
#undef X
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace ohgpu
