// fmt_line_kernel.hip -- line-coalesced kernel of the layout-changing processors:
//   a11  FlywheelInput::DoProcessFragment (StarvationRamper.cpp:117-186): packed BE interleaved -> planar BE 32-bit
//   a13  Sender::DoProcessFragment (Av/Songcast/Sender.cpp:351-377): first two channels, at most 3 MSBs each
//   a14  CodecFlac::CallbackWrite (Codec/Flac.cpp:379-417): planar host-endian TInt32 -> packed BE interleaved 8/16/24
// All three are "gather source subsample f(q), shuffle its bytes, write destination subsample q", so they share the PCM
// line kernel's structure (pcm_line_kernel.hip): a descriptor is cut into CHUNKS on the host, each described by one
// 64-byte record; a wave stages the chunk's source bytes in LDS with 16-byte global->LDS loads (one run, or one run per
// plane for a14), then lane = aligned destination dword: it reads the aligned LDS words that hold each contributing
// subsample, shuffles them with ONE v_perm_b32 (selector from the record), funnels and stores 4 bytes; only a chunk's
// first and last dword are written byte by byte.  The source subsample of destination subsample q is
// (q / A) * B + C + (q % A) * D, with (A, B, C, D) from the record (q / A by an exact multiplier).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ohgpu_internal.h"

namespace ohgpu {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* global_ptr_t;

constexpr uint32_t kFmtWaves = 4;
constexpr uint32_t kFmtInBytes = 2304;                                 // staged source bytes per chunk (incl. alignment slack)
constexpr uint32_t kFmtChunkSub = 512;                                 // destination subsamples per chunk

__global__ __launch_bounds__(kFmtWaves * 64) void fmt_line_kernel(const FmtChunk* __restrict__ chunks, const uint32_t n_chunks,
                                                                 const uint8_t* __restrict__ src, uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kFmtWaves][kFmtInBytes];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __attribute__((address_space(3))) uint8_t* in = (const __attribute__((address_space(3))) uint8_t*)&s_in[wave][0];
    for (uint32_t chunk = blockIdx.x * kFmtWaves + wave; chunk < n_chunks; chunk += gridDim.x * kFmtWaves) {
        const FmtChunk ck = chunks[chunk];
        const uint32_t sb = ck.sb, db = ck.db, nq = ck.nq;
        // ---- in: every run's aligned 16-byte pieces -> LDS (a piece always overlaps bytes the descriptor owns) ----
        const uint32_t head = (uint32_t)((uint64_t)(uintptr_t)src + ck.src_off) & 15u;   // the same for every run (host checks)
        const uint32_t n_pieces = (head + ck.run_bytes + 15u) >> 4;
        for (uint32_t r = 0; r < ck.n_runs; r++) {
            const uint8_t* base = src + (ck.src_off + (uint64_t)r * ck.run_src_stride - head);
            for (uint32_t p0 = 0; p0 < n_pieces; p0 += 64) {
                if (p0 + lane < n_pieces)                               // LDS destination = wave-uniform base + lane * 16
                    __builtin_amdgcn_global_load_lds((global_ptr_t)(base + (size_t)(p0 + lane) * 16),
                                                     (lds_ptr_t)(&s_in[wave][0] + r * ck.run_lds_stride + p0 * 16), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // destination subsample q -> its bytes in memory order (first byte lowest).  Past the chunk's end the lane
        // reads bytes that exist in the staging buffer; they only land in byte positions the edge path does not store.
        auto subsample = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t {
            const uint32_t i = ck.m_a ? (__umulhi(q, ck.m_a) >> ck.s_a) : q;            // q / A
            const uint32_t idx = __umul24(i, ck.map_b) + ck.map_c + __umul24(q - __umul24(i, ck.map_a), ck.map_d);
            const uint32_t off = head + __umul24(idx, sb);
            const __attribute__((address_space(3))) uint32_t* a = (const __attribute__((address_space(3))) uint32_t*)(in + (off & ~3u));
            return __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(a[1], a[0], off & 3u), ck.sel);
        };
        const uint32_t dhead = (uint32_t)((uint64_t)(uintptr_t)dst + ck.dst_off) & 3u;
        const uint32_t len = nq * db;
        const uint32_t n_dw = (dhead + len + 3u) >> 2;
        uint8_t* const obase = dst + (ck.dst_off - dhead);
        for (uint32_t kb = 0; kb < n_dw; kb += 64) {
            const uint32_t k = kb + lane;
            if (k < n_dw) {
                const int32_t pos = (int32_t)(k * 4) - (int32_t)dhead;  // stream position of the dword's first byte
                const uint32_t upos = pos < 0 ? 0u : (uint32_t)pos;
                uint32_t v;                                             // stream bytes [upos, upos + 4)
                if (db == 3) {
                    const uint32_t qa = __umulhi(upos, 0xAAAAAAABu) >> 1, o = upos - __umul24(qa, 3u);
                    const uint32_t v0 = subsample(qa), v1 = subsample(qa + 1);
                    const uint32_t sel = o == 0 ? 0x04020100u : (o == 1 ? 0x05040201u : 0x06050402u);
                    v = __builtin_amdgcn_perm(v1, v0, sel);
                } else if (db == 4) {
                    const uint32_t qa = upos >> 2, o = upos & 3;
                    v = subsample(qa);
                    if (dhead != 0) v = __builtin_amdgcn_alignbyte(subsample(qa + 1), v, o);
                } else if (db == 2) {
                    const uint32_t qa = upos >> 1, o = upos & 1;
                    v = subsample(qa) | (subsample(qa + 1) << 16);
                    if (dhead & 1) v = __builtin_amdgcn_alignbyte(subsample(qa + 2), v, o);
                } else {
                    v = subsample(upos) | (subsample(upos + 1) << 8) | (subsample(upos + 2) << 16) | (subsample(upos + 3) << 24);
                }
                if (pos >= 0 && (uint32_t)pos + 4 <= len) {
                    __builtin_nontemporal_store(v, (uint32_t*)(obase + (size_t)k * 4));       // written once, never read here
                } else {                                                // first / last dword of the chunk: only its own bytes
                    const uint32_t skip = pos < 0 ? (uint32_t)(-pos) : 0u;
                    for (uint32_t b = skip; b < 4; b++) {
                        if ((uint32_t)(pos + (int32_t)b) < len) obase[(size_t)k * 4 + b] = (uint8_t)(v >> (8 * (b - skip)));
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the staging buffer is reused by the next chunk
    }
}

// ---- register-only kernels for the two stereo cases that dominate in practice (the generic kernel above serves the
// rest): no LDS, one or two wide loads per lane, byte permutes with COMPILE-TIME selectors, wide stores.  A chunk is a
// whole descriptor here; record fields: src_off, dst_off, nq = frames, run_src_stride = the plane stride. ----
typedef uint32_t fv2 __attribute__((ext_vector_type(2)));
typedef uint32_t fv3 __attribute__((ext_vector_type(3)));
typedef uint32_t fv4 __attribute__((ext_vector_type(4)));

// selector of v_perm_b32 {hi (bytes 4-7), lo (bytes 0-3)} that takes `n` bytes from byte `first` on, low to high, zero above
static constexpr uint32_t sel_bytes(int first, int n)
{
    uint32_t s = 0;
    for (int t = 0; t < 4; t++) s |= (uint32_t)(t < n ? first + t : 0x0c) << (8 * t);
    return s;
}

// a11, stereo: lane = four frames = 8*SB interleaved source bytes -> four left-justified big-endian words per plane
template <int SB>
__global__ __launch_bounds__(256) void unpack_stereo_kernel(const FmtChunk* __restrict__ chunks, const uint32_t n_chunks,
                                                            const uint8_t* __restrict__ src, uint8_t* __restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t chunk = wave; chunk < n_chunks; chunk += gridDim.x * 4) {
        const FmtChunk ck = chunks[chunk];
        const uint8_t* sp = src + ck.src_off;
        uint8_t* p0 = dst + ck.dst_off;
        uint8_t* p1 = p0 + ck.run_src_stride;
        const uint32_t n_grp = ck.nq >> 2;
        for (uint32_t g = lane; g < n_grp; g += 64) {
            uint32_t in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint8_t* a = sp + (size_t)g * (8 * SB);
            if constexpr (SB == 2) { fv4 v; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory"); in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w; }
            else if constexpr (SB == 3) { fv3 v, w; asm volatile("global_load_dwordx3 %0, %2, off\n\tglobal_load_dwordx3 %1, %2, off offset:12\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory"); in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = w.x; in[4] = w.y; in[5] = w.z; }
            else { fv4 v, w; asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory"); in[0] = v.x; in[1] = v.y; in[2] = v.z; in[3] = v.w; in[4] = w.x; in[5] = w.y; in[6] = w.z; in[7] = w.w; }
            fv4 o0, o1;
#pragma unroll
            for (int f = 0; f < 4; f++) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int sidx = (2 * f + c) * SB, d = sidx >> 2;                 // first source byte, its dword
                    const uint32_t w = __builtin_amdgcn_perm(in[d + 1 < 8 ? d + 1 : 7], in[d], sel_bytes(sidx & 3, SB));
                    if (c == 0) o0[f] = w; else o1[f] = w;
                }
            }
            asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p0 + (size_t)g * 16), "v"(o0) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p1 + (size_t)g * 16), "v"(o1) : "memory");
        }
        const uint32_t f0 = n_grp * 4;                                                // the last 1..3 frames, byte by byte
        if (lane < (ck.nq - f0) * 8) {
            const uint32_t f = f0 + lane / 8, c = (lane >> 2) & 1, b = lane & 3;
            (c ? p1 : p0)[(size_t)f * 4 + b] = b < (uint32_t)SB ? sp[((size_t)f * 2 + c) * SB + b] : (uint8_t)0;
        }
    }
}

// a14, stereo: lane = two frames = two TInt32 of each plane -> four DB-byte big-endian subsamples, interleaved
template <int DB>
__global__ __launch_bounds__(256) void flac_stereo_kernel(const FmtChunk* __restrict__ chunks, const uint32_t n_chunks,
                                                          const uint8_t* __restrict__ src, uint8_t* __restrict__ dst)
{
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t chunk = wave; chunk < n_chunks; chunk += gridDim.x * 4) {
        const FmtChunk ck = chunks[chunk];
        const uint8_t* s0 = src + ck.src_off;
        const uint8_t* s1 = s0 + ck.run_src_stride;
        uint8_t* dp = dst + ck.dst_off;
        const uint32_t n_grp = ck.nq >> 1;
        for (uint32_t g = lane; g < n_grp; g += 64) {
            fv2 a, b;
            asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b) : "v"(s0 + (size_t)g * 8), "v"(s1 + (size_t)g * 8) : "memory");
            const uint32_t x[4] = {a.x, b.x, a.y, b.y};                               // frame 0: ch 0, ch 1; frame 1: ch 0, ch 1
            uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int o = k * DB;
                const uint32_t v = __builtin_bswap32(x[k]) >> (8 * (4 - DB));        // the low DB bytes, most significant first
                ow[o >> 2] |= v << (8 * (o & 3));
                if ((o & 3) + DB > 4) ow[(o >> 2) + 1] |= v >> (32 - 8 * (o & 3));
            }
            uint8_t* op = dp + (size_t)g * (4 * DB);
            if constexpr (DB == 1) asm volatile("global_store_dword %0, %1, off nt" : : "v"(op), "v"(ow[0]) : "memory");
            else if constexpr (DB == 2) { fv2 o = {ow[0], ow[1]}; asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(op), "v"(o) : "memory"); }
            else { fv3 o = {ow[0], ow[1], ow[2]}; asm volatile("global_store_dwordx3 %0, %1, off nt" : : "v"(op), "v"(o) : "memory"); }
        }
        if ((ck.nq & 1) && lane < 2 * DB) {                                           // an odd last frame, byte by byte
            const uint32_t f = ck.nq - 1, c = lane / DB, b = lane % DB;
            const uint32_t xv = *(const uint32_t*)((c ? s1 : s0) + (size_t)f * 4);
            dp[((size_t)f * 2 + c) * DB + b] = (uint8_t)(xv >> (8 * (DB - 1 - b)));
        }
    }
}

// ---- host side ----
void free_fmt_line(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    if (b->fmtline.d_chunks) ctx_dev_free(ctx, b->fmtline.d_chunks);
    if (b->fmtline.d_wide) ctx_dev_free(ctx, b->fmtline.d_wide);
    b->fmtline = FmtLinePlan();
}

int plan_fmt_line(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_fmt_desc* descs, size_t n)
{
    b->fmtline = FmtLinePlan();
    // A batch of Songcast packs that drop no channel (mono / stereo senders) is a plain depth conversion, big-endian to
    // big-endian: exactly the PCM line kernel's register-only group path.
    {
        bool as_pcm = n > 0;
        for (size_t i = 0; i < n && as_pcm; i++)
            as_pcm = descs[i].kind == OHGPU_FMT_SENDER_PACK && descs[i].channels <= 2 && descs[i].src_bits >= 16 &&
                     descs[i].src_bits == descs[0].src_bits && descs[i].channels == descs[0].channels;
        if (as_pcm) {
            std::vector<ohgpu_msg_desc> msgs(n);
            const uint8_t sbits = descs[0].src_bits, dbits = sbits < 24 ? sbits : 24;
            for (size_t i = 0; i < n; i++) {
                ohgpu_msg_desc& m = msgs[i];
                memset(&m, 0, sizeof(m));
                m.src_offset = descs[i].src_offset; m.dst_offset = descs[i].dst_offset; m.n_frames = descs[i].n_frames;
                m.ramp_start = m.ramp_end = OHGPU_RAMP_MAX; m.attenuation = OHGPU_UNITY_ATTENUATION;
                m.channels = descs[i].channels; m.src_bits = sbits; m.dst_bits = dbits;
                m.src_endian = m.dst_endian = OHGPU_ENDIAN_BIG;
            }
            b->uniform = true;
            b->channels = descs[0].channels; b->src_bits = sbits; b->dst_bits = dbits;
            b->src_endian = b->dst_endian = OHGPU_ENDIAN_BIG;
            return plan_pcm_line(ctx, b, msgs.data(), n);                // -> b->line; ohgpu_fmt_batch_run launches it
        }
    }
    // A batch of Songcast packs that all drop channels (streams wider than stereo) is the plain case of ohm_wide_kernel: a lane
    // per two frames, one 8-byte load per frame, no staging (csrc/ohm_frame_kernel.hip).
    {
        bool wide = n > 0;
        for (size_t i = 0; i < n && wide; i++)
            wide = descs[i].kind == OHGPU_FMT_SENDER_PACK && descs[i].channels > 2 &&
                   (uint64_t)descs[i].n_frames * descs[i].channels * (descs[i].src_bits / 8) <= 0xffffffffull;
        if (wide) {
            std::vector<OhmSelRec> recs;
            for (size_t i = 0; i < n; i++)
                if (descs[i].n_frames)
                    recs.push_back(wide_record(descs[i].src_offset, descs[i].dst_offset, descs[i].n_frames, descs[i].channels, descs[i].src_bits / 8,
                                               false, b->src_arena_bytes));
            if (!recs.empty()) {
                hipError_t e = ctx_dev_alloc(ctx, &b->fmtline.d_wide, recs.size() * sizeof(OhmSelRec));
                if (e == hipSuccess) e = hipMemcpy(b->fmtline.d_wide, recs.data(), recs.size() * sizeof(OhmSelRec), hipMemcpyHostToDevice);
                if (e != hipSuccess) {
                    free_fmt_line(ctx, b);
                    return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "record upload: %s", hipGetErrorString(e));
                }
                b->fmtline.n_wide = (uint32_t)recs.size();
                return OHGPU_OK;
            }
        }
    }
    // Uniform stereo batches of a11 (16/24/32-bit) or a14 take the register-only kernels: one record per descriptor.
    {
        const uint8_t kind = n ? descs[0].kind : 0;
        bool grp = n > 0 && (kind == OHGPU_FMT_UNPACK_PLANAR || kind == OHGPU_FMT_FLAC_PACK);
        for (size_t i = 0; i < n && grp; i++) {
            const ohgpu_fmt_desc& d = descs[i];
            grp = d.kind == kind && d.channels == 2 && d.src_bits == descs[0].src_bits && d.dst_bits == descs[0].dst_bits &&
                  (kind == OHGPU_FMT_FLAC_PACK || d.src_bits >= 16);
        }
        if (grp) {
            std::vector<FmtChunk> recs;
            for (size_t i = 0; i < n; i++) {
                const ohgpu_fmt_desc& d = descs[i];
                if (d.n_frames == 0) continue;
                FmtChunk c;
                memset(&c, 0, sizeof(c));
                c.src_off = d.src_offset; c.dst_off = d.dst_offset; c.nq = d.n_frames;
                c.run_src_stride = kind == OHGPU_FMT_FLAC_PACK ? d.src_plane_stride : d.dst_plane_stride;
                recs.push_back(c);
            }
            if (recs.empty()) return OHGPU_OK;
            hipError_t e = ctx_dev_alloc(ctx, &b->fmtline.d_chunks, recs.size() * sizeof(FmtChunk));
            if (e == hipSuccess) e = hipMemcpy(b->fmtline.d_chunks, recs.data(), recs.size() * sizeof(FmtChunk), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                free_fmt_line(ctx, b);
                return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "record upload: %s", hipGetErrorString(e));
            }
            b->fmtline.n_chunks = (uint32_t)recs.size();
            b->fmtline.group_kind = kind;
            b->fmtline.group_bytes = (uint8_t)((kind == OHGPU_FMT_FLAC_PACK ? descs[0].dst_bits : descs[0].src_bits) / 8);
            b->fmtline.enabled = true;
            return OHGPU_OK;
        }
    }
    std::vector<FmtChunk> chunks;
    const uint32_t budget = kFmtInBytes - 32;                           // 15 bytes of head, rounding up to pieces, the +4 over-read
    for (size_t i = 0; i < n; i++) {
        const ohgpu_fmt_desc& d = descs[i];
        if (d.n_frames == 0) continue;
        const uint32_t ch = d.channels, sb = d.src_bits / 8;
        FmtChunk c;
        memset(&c, 0, sizeof(c));
        uint32_t frames_per_chunk, sub_per_frame, sh;
        if (d.kind == OHGPU_FMT_UNPACK_PLANAR) {                        // one chunk list per plane: q = frame
            c.sb = (uint8_t)sb; c.db = 4; c.n_runs = 1;
            c.map_a = 1; c.map_b = (uint16_t)ch; c.map_d = 0;
            for (uint32_t m = 0; m < 4; m++) c.sel |= (m < sb ? m : 0x0cu) << (8 * m);     // left-justified, low bytes zero
            sub_per_frame = 1;
            frames_per_chunk = budget / (ch * sb) > 4 ? std::min<uint32_t>(kFmtChunkSub, budget / (ch * sb) - 4) : 0;   // (the lanes past the end read up to 3 frames on)
        } else if (d.kind == OHGPU_FMT_SENDER_PACK) {                   // q = frame * out_ch + channel
            const uint32_t out_ch = ch < 2 ? ch : 2, db = sb < 3 ? sb : 3;
            c.sb = (uint8_t)sb; c.db = (uint8_t)db; c.n_runs = 1;
            c.map_a = (uint16_t)out_ch; c.map_b = (uint16_t)ch; c.map_c = (uint16_t)(ch < 10 ? 0u : 8u); c.map_d = 1;
            for (uint32_t m = 0; m < 4; m++) c.sel |= (m < db ? m : 0x0cu) << (8 * m);     // the db most significant bytes
            sub_per_frame = out_ch;
            frames_per_chunk = budget / (ch * sb) > 4 ? std::min<uint32_t>(kFmtChunkSub / out_ch, budget / (ch * sb) - 4) : 0;
        } else {                                                        // FLAC: q = frame * ch + channel, one staged run per plane
            const uint32_t db = d.dst_bits / 8;
            if (d.src_plane_stride % 16 != 0 && ch > 1) return OHGPU_OK;   // planes would sit differently in their pieces: generic kernel
            c.sb = 4; c.db = (uint8_t)db; c.n_runs = (uint16_t)ch;
            c.map_a = (uint16_t)ch; c.map_b = 1; c.map_d = 0;                      // map_d set per chunk (run_lds_stride / 4)
            for (uint32_t m = 0; m < 4; m++) c.sel |= (m < db ? db - 1 - m : 0x0cu) << (8 * m);   // BE bytes of the low db bytes of a LE word
            sub_per_frame = ch;
            frames_per_chunk = budget / ch > 80 ? std::min<uint32_t>(kFmtChunkSub / ch, (budget / ch - 64) / 4) : 0;   // ch runs of <= 4f + 62 bytes
        }
        if (frames_per_chunk == 0) return OHGPU_OK;                     // a frame does not fit the staging buffer: generic kernel
        magic_u31(c.map_a, &c.m_a, &sh); c.s_a = (uint8_t)sh;
        const uint32_t planes = d.kind == OHGPU_FMT_UNPACK_PLANAR ? ch : 1;
        for (uint32_t p = 0; p < planes; p++) {
            for (uint32_t f0 = 0; f0 < d.n_frames; f0 += frames_per_chunk) {
                const uint32_t f = std::min<uint32_t>(frames_per_chunk, d.n_frames - f0);
                FmtChunk k = c;
                k.nq = f * sub_per_frame;
                if (d.kind == OHGPU_FMT_UNPACK_PLANAR) {
                    k.map_c = (uint16_t)p;
                    k.src_off = d.src_offset + (uint64_t)f0 * ch * sb;
                    k.run_bytes = f * ch * sb;
                    k.dst_off = d.dst_offset + (uint64_t)p * d.dst_plane_stride + (uint64_t)f0 * 4;
                } else if (d.kind == OHGPU_FMT_SENDER_PACK) {
                    k.src_off = d.src_offset + (uint64_t)f0 * ch * sb;
                    k.run_bytes = f * ch * sb;
                    k.dst_off = d.dst_offset + (uint64_t)f0 * c.map_a * c.db;
                } else {
                    k.src_off = d.src_offset + (uint64_t)f0 * 4;
                    k.run_bytes = f * 4;
                    k.run_src_stride = d.src_plane_stride;
                    k.run_lds_stride = (uint16_t)(((15 + f * 4 + 15) / 16) * 16 + 16);
                    k.map_d = (uint16_t)(k.run_lds_stride / 4);
                    k.dst_off = d.dst_offset + (uint64_t)f0 * ch * c.db;
                }
                chunks.push_back(k);
            }
        }
    }
    if (chunks.empty() || chunks.size() > 0xffffffffull) return OHGPU_OK;
    hipError_t e = ctx_dev_alloc(ctx, &b->fmtline.d_chunks, chunks.size() * sizeof(FmtChunk));
    if (e == hipSuccess) e = hipMemcpy(b->fmtline.d_chunks, chunks.data(), chunks.size() * sizeof(FmtChunk), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        free_fmt_line(ctx, b);
        return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "chunk plan upload: %s", hipGetErrorString(e));
    }
    b->fmtline.n_chunks = (uint32_t)chunks.size();
    b->fmtline.enabled = true;
    return OHGPU_OK;
}

hipError_t launch_fmt_line(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t grid = (b->fmtline.n_chunks + kFmtWaves - 1) / kFmtWaves;
    // Workgroups per CU, as for pcm_line_kernel (round 5): the register-only stereo kernels are streaming copies with a byte shuffle
    // in the middle, and the memory system serves those best from few workgroups -- a11's unpack 0.160-0.167 ms at eight per CU,
    // 0.147-0.152 at four (0.66-0.69 -> 0.72-0.75 of 8 TB/s); a14's pack 0.170 -> 0.163 at six; the staged generic kernel keeps eight.
#ifdef OHGPU_LINE_FMT_GROUPS_PER_CU
    const uint32_t per_cu = OHGPU_LINE_FMT_GROUPS_PER_CU;
#else
    const uint32_t per_cu = b->fmtline.group_kind == OHGPU_FMT_UNPACK_PLANAR ? 4u : (b->fmtline.group_kind == OHGPU_FMT_FLAC_PACK ? 6u : 8u);
#endif
    if (grid > cus * per_cu) grid = cus * per_cu;
    const FmtChunk* recs = (const FmtChunk*)b->fmtline.d_chunks;
    const uint32_t nr = b->fmtline.n_chunks;
    if (b->fmtline.group_kind == OHGPU_FMT_UNPACK_PLANAR) {
        if (b->fmtline.group_bytes == 2) hipLaunchKernelGGL(unpack_stereo_kernel<2>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        else if (b->fmtline.group_bytes == 3) hipLaunchKernelGGL(unpack_stereo_kernel<3>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        else hipLaunchKernelGGL(unpack_stereo_kernel<4>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        return hipGetLastError();
    }
    if (b->fmtline.group_kind == OHGPU_FMT_FLAC_PACK) {
        if (b->fmtline.group_bytes == 1) hipLaunchKernelGGL(flac_stereo_kernel<1>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        else if (b->fmtline.group_bytes == 2) hipLaunchKernelGGL(flac_stereo_kernel<2>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        else hipLaunchKernelGGL(flac_stereo_kernel<3>, dim3(grid), dim3(256), 0, s, recs, nr, src, dst);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fmt_line_kernel, dim3(grid), dim3(kFmtWaves * 64), 0, s,
                       (const FmtChunk*)b->fmtline.d_chunks, b->fmtline.n_chunks, src, dst);
    return hipGetLastError();
}

}  // namespace ohgpu
