// src_block_kernel.hip -- the tuned resample -> ramp -> pack kernel ("block kernel") for mono/stereo streams.
//
// Mapping (DESIGN.md "Resampler kernel"):
//   * A stream's output is cut into BLOCKS of L_blk frames that start where the polyphase phase is 0
//     (L_blk is a multiple of L), so every block walks the same phase sequence.
//   * One lane owns one block (all CPL channels of it).  All 64 lanes of a wave are therefore at the SAME phase at
//     the same instruction: the coefficient address is wave-uniform, coefficients come through the scalar cache
//     into SGPRs and each tap is ONE v_fma_f64 (SGPR coefficient, VGPR sample, VGPR accumulator).
//   * The lane keeps its T-frame sliding window in registers as exact integer-valued doubles.  The advance loop
//     is unrolled T times so that the circular window is indexed statically (slot = advance mod T).
//   * Input is staged through LDS by direct global->LDS loads (16 B per lane, two buffers, eight advances per
//     stage).  Rounded outputs go to a lane-private LDS ring; at each stage boundary a lane turns every complete
//     group of 8 (or 16) outputs into packed bytes (ramp + depth/endian conversion, pcm_device.h's code) and
//     writes them with aligned 16-byte stores.
//   * Accumulation is fp64 FMA on integer-valued operands with |sum| < 2^53: exact, hence bit-identical to the
//     integer model regardless of order.  No MFMA: this is a 1-D filter.
// Formats are template parameters (the per-advance unpack sits in the unrolled hot path); layouts without an
// instantiation run on the generic kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* global_ptr_t;
typedef const __attribute__((address_space(4))) double* const_f64_ptr_t;   // constant address space: scalar loads

// calls f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a compile-time unrolled loop
template <typename F, int... S>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, S...>)
{
    (f(std::integral_constant<int, S>{}), ...);
}

// ---- explicit scalar-cache loads (the compiler does not see them: every use is fenced by coef_wait) ----
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// eight consecutive coefficients (64 bytes = one scalar-cache line) into 16 SGPRs
__device__ __forceinline__ void coef_load8(u32x16& q, const_f64_ptr_t p)
{
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(q) : "s"(p));
}
// all outstanding scalar loads (and LDS operations) have landed; q is usable afterwards
__device__ __forceinline__ void coef_wait(u32x16& q)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q));
}
__device__ __forceinline__ double coef_get(const u32x16& q, int k)
{
    return __hiloint2double((int)q[2 * k + 1], (int)q[2 * k]);
}

// one frame (CPL subsamples of SB bytes, packed; 2-byte aligned when CPL*SB is even) from LDS -> S24 integers
template <int CPL, int SB, bool LE>
__device__ __forceinline__ void lds_load_frame(const __attribute__((address_space(3))) uint8_t* fp, int32_t (&x)[CPL])
{
    typedef const __attribute__((address_space(3))) uint16_t* lds_u16_t;
    if constexpr (CPL == 2 && SB == 3) {
        // three 16-bit reads: h0 = b0 b1, h1 = b2 b3, h2 = b4 b5 (little-endian halves)
        const uint32_t h0 = *(lds_u16_t)(fp), h1 = *(lds_u16_t)(fp + 2), h2 = *(lds_u16_t)(fp + 4);
        if constexpr (LE) {
            x[0] = ((int32_t)((h0 | (h1 << 16)) << 8)) >> 8;                       // b2 b1 b0
            x[1] = ((int32_t)(((h1 >> 8) | (h2 << 8)) << 8)) >> 8;                 // b5 b4 b3
        } else {
            x[0] = ((int32_t)__builtin_bswap32(h0 | (h1 << 16))) >> 8;             // b0 b1 b2 (b0 = MSB)
            x[1] = ((int32_t)__builtin_bswap32((h1 >> 8) | (h2 << 8))) >> 8;       // b3 b4 b5
        }
    } else if constexpr (SB == 2) {
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const uint32_t h = *(lds_u16_t)(fp + 2 * c);                            // b0 | b1 << 8
            const uint32_t v = LE ? h : (((h & 0xffu) << 8) | (h >> 8));
            x[c] = ((int32_t)(v << 16)) >> 8;                                      // left-justify to 32, then S24
        }
    } else {
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < SB; b++) w |= (uint32_t)fp[c * SB + (LE ? SB - 1 - b : b)] << (24 - 8 * b);
            x[c] = ((int32_t)w) >> 8;
        }
    }
}

template <int T, int CPL, int SB, bool SRC_LE, int DB, bool DST_LE>
__global__ __launch_bounds__(256, 2)
void src_block_kernel(const SrcSeg* __restrict__ segs, const SegMsg* __restrict__ msgs, const SrcWork* __restrict__ work,
                      const double* __restrict__ coef, const uint16_t* __restrict__ ramp_table,
                      const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                      const uint64_t src_arena_bytes, const int L, const int M, const uint32_t L_blk, const uint32_t M_blk)
{
    constexpr int OC = (CPL * DB * 8) % 16 == 0 ? 8 : 16;     // outputs per store group: OC*CPL*DB is a multiple of 16
    constexpr int OC_LOG2 = OC == 8 ? 3 : 4;
    constexpr int RING = 2 * OC;                              // ring entries per lane; the planner checks OC-1 + outputs/stage <= RING
    constexpr int GROUP_DWORDS = OC * CPL * DB / 4;
    constexpr int FB_SRC = CPL * SB, FB_DST = CPL * DB;
    constexpr int IN_BLOCKS = ((8 * FB_SRC + 15 + 15) / 16) | 1;   // 16-byte pieces per staged row, odd (bank spread)
    constexpr int IN_STRIDE = IN_BLOCKS * 16;

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const const_f64_ptr_t coef_c = (const_f64_ptr_t)coef;

    uint16_t* s_ramp = (uint16_t*)smem;                              // 1 KiB: RampArray
    uint8_t* s_in = smem + 1024;                                     // 2 x 256 x IN_STRIDE raw packed input
    int32_t* s_ring = (int32_t*)(s_in + 2 * 256 * IN_STRIDE);        // [RING][256][CPL] rounded S24 outputs

    for (uint32_t i = tid; i < kRampTableCount; i += 256) s_ramp[i] = ramp_table[i];

    const SrcWork wk = work[blockIdx.x];
    const SrcSeg seg = segs[wk.seg];
    const uint32_t n_blocks = wk.n_blocks;
    const uint32_t row = tid;
    const bool lane_valid = row < n_blocks;
    const uint64_t blk = wk.first_block + row;
    const int64_t n_start = (int64_t)(blk * M_blk);     // absolute input frame at advance a = 0
    const uint64_t m_start = blk * L_blk;               // absolute output frame at j = 0
    const int64_t row_g = seg.src_base + (n_start - T) * (int64_t)FB_SRC;   // byte offset of the frame at a_lin = 0
    uint8_t* const row_dst = dst + seg.dst_base + (int64_t)(m_start * FB_DST);
    // The stream start reads as zeros.  Blocks are at least T input frames long (planner), so only a block that
    // starts at input frame 0 reaches before the stream: its whole warm-up pass (advances a < 0) must be zeros.
    const bool first_block = n_start == 0;

    // message that holds this lane's first output frame (messages of a segment tile its output range)
    uint32_t mi = seg.msg_begin;
    SegMsg cur;
    cur.out0 = 0; cur.n = 0xffffffffu; cur.ramp_start = 0; cur.ramp_end = 0; cur.flags = 0;
    if (lane_valid) {
        uint32_t lo = seg.msg_begin, hi = seg.msg_end;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (msgs[mid].out0 <= m_start) lo = mid; else hi = mid;
        }
        mi = lo;
        cur = msgs[mi];
    }

    // ---- input staging: stage q holds advances [8q - T, 8q + 8 - T) of every row, as raw packed bytes.
    // Thread `tid` moves pieces idx = it*256 + tid (it < IN_BLOCKS): piece `part` of row r = idx / IN_BLOCKS.
    int64_t piece_g[IN_BLOCKS];      // unaligned byte offset of that row's frame at a_lin = 0
    int32_t piece_part[IN_BLOCKS];   // -1: nothing to move
#pragma unroll
    for (int it = 0; it < IN_BLOCKS; it++) {
        const uint32_t idx = it * 256 + tid;
        const uint32_t r = idx / IN_BLOCKS;
        piece_g[it] = seg.src_base + ((int64_t)((wk.first_block + r) * M_blk) - T) * (int64_t)FB_SRC;
        piece_part[it] = (r < n_blocks) ? (int32_t)(idx - r * IN_BLOCKS) : -1;
    }
    auto issue_stage = [&](int q) __attribute__((always_inline)) {
        uint8_t* buf = s_in + (uint32_t)(q & 1) * 256 * IN_STRIDE;
        const int64_t shift = (int64_t)q * 8 * FB_SRC;
#pragma unroll
        for (int it = 0; it < IN_BLOCKS; it++) {
            if (piece_part[it] >= 0) {
                const int64_t addr = ((piece_g[it] + shift) & ~(int64_t)15) + 16 * (int64_t)piece_part[it];
                if (addr >= 0 && (uint64_t)addr + 16 <= src_arena_bytes) {
                    uint8_t* wave_dst = buf + (size_t)(it * 256 + wave * 64) * 16;   // LDS dest = wave-uniform base + lane*16
                    __builtin_amdgcn_global_load_lds((global_ptr_t)(src + addr), (lds_ptr_t)wave_dst, 16, 0, 0);
                } else {
                    uint8_t* d = buf + (size_t)(it * 256 + tid) * 16;               // piece straddles an end of the arena
                    for (int b = 0; b < 16; b++) {
                        const int64_t a1 = addr + b;
                        d[b] = (a1 >= 0 && (uint64_t)a1 < src_arena_bytes) ? src[a1] : (uint8_t)0;
                    }
                }
            }
        }
    };

    // ---- tail: ramp + pack + store every complete group of OC outputs sitting in the lane's ring ----
    uint32_t drained = 0;                                 // groups written so far (wave-uniform)
    auto drain = [&](int j_now) __attribute__((always_inline)) {
        while (drained < ((uint32_t)j_now >> OC_LOG2)) {
            if (lane_valid) {
                uint32_t packed[GROUP_DWORDS];
#pragma unroll
                for (int d = 0; d < GROUP_DWORDS; d++) packed[d] = 0;
                const uint32_t j0 = drained << OC_LOG2;
                static_for([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    const uint32_t jo = j0 + o;
                    const int32_t* e = s_ring + ((jo & (RING - 1)) * 256 + row) * CPL;
                    uint32_t i = (uint32_t)(m_start + jo - cur.out0);
                    while (i >= cur.n) {
                        mi++;
                        cur = msgs[mi];
                        i = (uint32_t)(m_start + jo - cur.out0);
                    }
                    const bool ramp = (cur.flags & OHGPU_FLAG_RAMP) != 0;
                    uint32_t mult = 0;
                    if (ramp) {
                        const int32_t tot = (int32_t)((uint32_t)cur.ramp_start - (uint32_t)cur.ramp_end);
                        mult = s_ramp[ramp_index(cur.ramp_start, tot, (int32_t)i, (int32_t)cur.n)];
                    }
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        uint32_t w = ((uint32_t)e[c]) << 8;                 // left-justified BE word
                        if (ramp) w = ramp_word(w, mult, 3, CPL, c);
                        if (DB == 4 && (cur.flags & OHGPU_FLAG_ZERO_LSB32)) w &= 0xffffff00u;
                        // v = the DB bytes in memory order, first byte in the low bits
                        const uint32_t v = DST_LE ? (w >> (32 - 8 * DB))
                                                  : (__builtin_bswap32(w) & (DB == 4 ? 0xffffffffu : ((1u << (8 * (DB & 3))) - 1)));
                        constexpr int pos = (o * CPL) * DB;
                        const int bp = pos + c * DB;
                        const int dw = bp >> 2, sh = (bp & 3) * 8;
                        packed[dw] |= v << sh;
                        if (sh + 8 * DB > 32) packed[dw + 1] |= v >> (32 - sh);
                    }
                }, std::make_integer_sequence<int, OC>{});
                uint4* out = (uint4*)(row_dst + (size_t)j0 * FB_DST);
#pragma unroll
                for (int q4 = 0; q4 < GROUP_DWORDS / 4; q4++)
                    out[q4] = make_uint4(packed[4 * q4], packed[4 * q4 + 1], packed[4 * q4 + 2], packed[4 * q4 + 3]);
            }
            drained++;
        }
    };

    double win[T][CPL];
#pragma unroll
    for (int s = 0; s < T; s++)
#pragma unroll
        for (int c = 0; c < CPL; c++) win[s][c] = 0.0;

    const int total = (int)M_blk + T;         // advances a = a_lin - T for a_lin in [0, total)
    int j = 0;                                // outputs emitted so far (wave-uniform)
    int t = 0;                                // j * M
    uint32_t in_off = 0;                      // byte offset (from smem) of the current stage's first frame of this row
    const __attribute__((address_space(3))) uint8_t* smem_lds = (const __attribute__((address_space(3))) uint8_t*)smem;
    const bool any_first = __any(first_block) != 0;

    issue_stage(0);

    for (int g = 0; g * T < total; g++) {
        static_for([&](auto slot) __attribute__((always_inline)) {
            constexpr int s = decltype(slot)::value;
            const int a_lin = g * T + s;
            if (a_lin >= total) return;
            const int a = a_lin - T;
            if constexpr ((s & 7) == 0) {
                const int q = a_lin >> 3;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                          // stage q landed everywhere; stage q-1 fully consumed
                if ((q + 1) * 8 < total) issue_stage(q + 1);
                drain(j);
                in_off = 1024 + (uint32_t)(q & 1) * 256 * IN_STRIDE + row * IN_STRIDE +
                         (uint32_t)((row_g + (int64_t)q * 8 * FB_SRC) & 15);
            }
            // coefficients of this advance's first output: the first line is requested before the frame is unpacked
            const bool emits = t < L * (a + 1);
            u32x16 qa, qb;
            const_f64_ptr_t cp = coef_c;
            if (emits) {
                cp = coef_c + (size_t)__builtin_amdgcn_readfirstlane(t - L * a) * T;
                coef_load8(qa, cp);
            }
            // ---- advance: frame (n_start + a) enters slot s ----
            {
                int32_t x[CPL];
                lds_load_frame<CPL, SB, SRC_LE>(smem_lds + in_off + (s & 7) * FB_SRC, x);
#pragma unroll
                for (int c = 0; c < CPL; c++) win[s][c] = (double)x[c];
            }
            // ---- emit the outputs whose newest input frame is this one: floor(t / L) == a ----
            // Coefficients move through two 16-SGPR buffers, one 64-byte line (8 taps) at a time: while the FMAs of
            // line i run, line i+1 is in flight.  (Every register an in-flight scalar load writes stays live until
            // a wait covers it: a destination that dies early would be reused and then overwritten by the late data.)
            if (emits) {
                while (true) {
                    double acc[CPL][2];
#pragma unroll
                    for (int c = 0; c < CPL; c++) acc[c][0] = acc[c][1] = 0.0;
                    static_for([&](auto line) __attribute__((always_inline)) {
                        constexpr int li = decltype(line)::value;
                        u32x16& qcur = (li & 1) ? qb : qa;
                        u32x16& qnext = (li & 1) ? qa : qb;
                        coef_wait(qcur);
                        if constexpr (li + 1 < T / 8) {
                            coef_load8(qnext, cp + 8 * (li + 1));
                        }
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) {
                            constexpr int k0 = li * 8;
                            const double ck = coef_get(qcur, kk);
#pragma unroll
                            for (int c = 0; c < CPL; c++)
                                acc[c][kk & 1] = fma(ck, win[(s - (k0 + kk) + 2 * T) % T][c], acc[c][kk & 1]);
                        }
                    }, std::make_integer_sequence<int, T / 8>{});
                    int32_t* e = s_ring + (((uint32_t)j & (RING - 1)) * 256 + row) * CPL;
#pragma unroll
                    for (int c = 0; c < CPL; c++) e[c] = src_round_s24(acc[c][0] + acc[c][1]);
                    j++;
                    t += M;
                    if (!(t < L * (a + 1))) break;
                    cp = coef_c + (size_t)__builtin_amdgcn_readfirstlane(t - L * a) * T;
                    coef_load8(qa, cp);
                }
            }
        }, std::make_integer_sequence<int, T>{});
        if (g == 0 && any_first) {                        // warm-up pass done: a stream's first block starts from silence
#pragma unroll
            for (int s = 0; s < T; s++)
#pragma unroll
                for (int c = 0; c < CPL; c++) win[s][c] = first_block ? 0.0 : win[s][c];
        }
    }
    drain(j);
}

// ---- instantiations: (T, CPL, source bytes, source LE, destination bytes, destination LE) ----
#define OHGPU_BLOCK_KERNELS(X)      \
    X(32, 2, 3, true, 3, false)     \
    X(32, 2, 3, true, 3, true)      \
    X(32, 2, 3, false, 3, false)    \
    X(32, 2, 3, true, 4, false)     \
    X(32, 2, 3, true, 2, false)     \
    X(32, 2, 2, true, 3, false)     \
    X(32, 2, 2, true, 2, true)      \
    X(32, 2, 2, true, 2, false)

template <int T, int CPL, int SB, bool SRC_LE, int DB, bool DST_LE>
static hipError_t launch_one(const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_block_kernel<T, CPL, SB, SRC_LE, DB, DST_LE>;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->fast.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(b->fast.n_work), dim3(256), b->fast.lds_bytes, s,
                       p.segs, p.msgs, p.work, p.coef, p.ramp_table, p.src, p.dst,
                       p.src_arena_bytes, (int)p.L, (int)p.M, p.L_blk, p.M_blk);
    return hipGetLastError();
}

bool src_block_supported(uint32_t T, uint32_t cpl, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le)
{
#define X(t, c, s_, sl, d, dl) \
    if (T == t && cpl == c && sb == s_ && (src_le != 0) == sl && db == d && (dst_le != 0) == dl) return true;
    OHGPU_BLOCK_KERNELS(X)
#undef X
    return false;
}

uint32_t src_block_in_blocks(uint32_t cpl, uint32_t sb)
{
    return ((8 * cpl * sb + 15 + 15) / 16) | 1;
}

hipError_t launch_src_block(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    if (!b->fast.enabled || b->fast.n_work == 0) return hipSuccess;
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    prm.ramp_table = ctx->d_ramp_table;
    const uint32_t T = b->fast.T, cpl = b->fast.cpl;
#define X(t, c, s_, sl, d, dl)                                                                                   \
    if (T == t && cpl == c && prm.sb == s_ && (prm.src_le != 0) == sl && prm.db == d && (prm.dst_le != 0) == dl) \
        return launch_one<t, c, s_, sl, d, dl>(b, prm, s);
    OHGPU_BLOCK_KERNELS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace ohgpu
