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
#include <hip/hip_runtime.h>

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

// one frame (CPL subsamples of sb bytes, packed) from LDS -> S24 integers
template <int CPL>
__device__ __forceinline__ void lds_load_frame(const uint8_t* fp, uint32_t sb, bool little, int32_t (&x)[CPL])
{
    if (CPL == 2 && sb == 3 && little) {
        // S24LE stereo, 2-byte aligned: three 16-bit reads h0 = b0 b1, h1 = b2 b3, h2 = b4 b5
        const uint32_t h0 = *(const uint16_t*)(fp), h1 = *(const uint16_t*)(fp + 2), h2 = *(const uint16_t*)(fp + 4);
        x[0] = ((int32_t)((h0 | (h1 << 16)) << 8)) >> 8;
        x[CPL - 1] = ((int32_t)(((h1 >> 8) | (h2 << 8)) << 8)) >> 8;
    } else {
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            uint32_t w = 0;
            for (uint32_t b = 0; b < sb; b++) w |= (uint32_t)fp[c * sb + (little ? sb - 1 - b : b)] << (24 - 8 * b);
            x[c] = ((int32_t)w) >> 8;
        }
    }
}

template <int T, int CPL, int DB>
__global__ __launch_bounds__(256, 2)
void src_block_kernel(const SrcSeg* __restrict__ segs, const SegMsg* __restrict__ msgs, const SrcWork* __restrict__ work,
                      const double* __restrict__ coef, const uint16_t* __restrict__ ramp_table,
                      const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const SrcFastParams p)
{
    constexpr int OC = (CPL * DB * 8) % 16 == 0 ? 8 : 16;     // outputs per store group: OC*CPL*DB is a multiple of 16
    constexpr int OC_LOG2 = OC == 8 ? 3 : 4;
    constexpr int RING = 2 * OC;                              // ring entries per lane; the planner checks OC-1 + outputs/stage <= RING
    constexpr int GROUP_DWORDS = OC * CPL * DB / 4;
    constexpr int FB_DST = CPL * DB;

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SrcWork wk = work[blockIdx.x];
    const SrcSeg seg = segs[wk.seg];
    const uint32_t sb = p.sb;
    const uint32_t fb_src = CPL * sb;
    const uint32_t in_blocks = p.in_blocks, in_stride = in_blocks * 16;
    const bool src_le = p.src_le != 0, dst_le = p.dst_le != 0;
    const int L = (int)p.L, M = (int)p.M;
    const const_f64_ptr_t coef_c = (const_f64_ptr_t)coef;

    uint16_t* s_ramp = (uint16_t*)smem;                 // 1 KiB: RampArray
    uint8_t* s_in = smem + 1024;                        // 2 x 256 x in_stride raw packed input
    int32_t* s_ring = (int32_t*)(s_in + 2 * 256 * in_stride);   // [RING][256][CPL] rounded S24 outputs

    for (uint32_t i = tid; i < kRampTableCount; i += 256) s_ramp[i] = ramp_table[i];

    const uint32_t row = tid;
    const bool lane_valid = row < wk.n_blocks;
    const uint64_t blk = wk.first_block + row;
    const int64_t n_start = (int64_t)(blk * p.M_blk);   // absolute input frame at advance a = 0
    const uint64_t m_start = blk * p.L_blk;             // absolute output frame at j = 0
    const int64_t row_src = seg.src_base + n_start * (int64_t)fb_src;
    uint8_t* const row_dst = dst + seg.dst_base + (int64_t)(m_start * FB_DST);

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

    // ---- input staging: stage q holds advances [8q - T, 8q + 8 - T) of every row, as raw packed bytes ----
    auto issue_stage = [&](int q) {
        uint8_t* buf = s_in + (uint32_t)(q & 1) * 256 * in_stride;
        const uint32_t total = 256 * in_blocks;
        for (uint32_t base = 0; base < total; base += 256) {
            const uint32_t idx = base + tid;
            const uint32_t r = idx / in_blocks, part = idx - r * in_blocks;
            if (r < wk.n_blocks) {
                const int64_t frame = (int64_t)((wk.first_block + r) * p.M_blk) + (int64_t)q * 8 - T;
                const int64_t g = seg.src_base + frame * (int64_t)fb_src;
                const int64_t addr = (g & ~(int64_t)15) + 16 * (int64_t)part;
                if (addr >= 0 && (uint64_t)addr + 16 <= p.src_arena_bytes) {
                    uint8_t* wave_dst = buf + (size_t)(base + wave * 64) * 16;   // LDS dest = wave-uniform base + lane*16
                    __builtin_amdgcn_global_load_lds((global_ptr_t)(src + addr), (lds_ptr_t)wave_dst, 16, 0, 0);
                } else {
                    uint8_t* d = buf + (size_t)idx * 16;        // piece straddles an end of the arena
                    for (int b = 0; b < 16; b++) {
                        const int64_t a1 = addr + b;
                        d[b] = (a1 >= 0 && (uint64_t)a1 < p.src_arena_bytes) ? src[a1] : (uint8_t)0;
                    }
                }
            }
        }
    };

    // ---- tail: ramp + pack + store every complete group of OC outputs sitting in the lane's ring ----
    uint32_t drained = 0;                                 // groups written so far (wave-uniform)
    auto drain = [&](int j_now) {
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
                    const bool zero_lsb = (cur.flags & OHGPU_FLAG_ZERO_LSB32) != 0;
                    uint32_t mult = 0;
                    if (ramp) {
                        const int32_t tot = (int32_t)((uint32_t)cur.ramp_start - (uint32_t)cur.ramp_end);
                        mult = s_ramp[ramp_index(cur.ramp_start, tot, (int32_t)i, (int32_t)cur.n)];
                    }
#pragma unroll
                    for (int c = 0; c < CPL; c++) {
                        uint32_t w = ((uint32_t)e[c]) << 8;                 // left-justified BE word
                        if (ramp) w = ramp_word(w, mult, 3, CPL, c);
                        if (DB == 4 && zero_lsb) w &= 0xffffff00u;
                        // v = the DB bytes in memory order, first byte in the low bits
                        const uint32_t v = dst_le ? (w >> (32 - 8 * DB)) : (__builtin_bswap32(w) & (DB == 4 ? 0xffffffffu : ((1u << (8 * (DB & 3))) - 1)));
                        constexpr int pos = (o * CPL) * DB;                // byte position of subsample c = 0 in the group
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

    const int total = (int)p.M_blk + T;       // advances a = a_lin - T for a_lin in [0, total)
    int j = 0;                                // outputs emitted so far (wave-uniform)
    int t = 0;                                // j * M
    const uint8_t* in_ptr = s_in;

    issue_stage(0);

    for (int g = 0; g * T < total; g++) {
        static_for([&](auto slot) __attribute__((always_inline)) {
            constexpr int s = decltype(slot)::value;
            const int a_lin = g * T + s;
            if (a_lin >= total) return;
            const int a = a_lin - T;
            if ((s & 7) == 0) {
                const int q = a_lin >> 3;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                          // stage q landed everywhere; stage q-1 fully consumed
                if ((q + 1) * 8 < total) issue_stage(q + 1);
                drain(j);
                const int64_t gq = row_src + ((int64_t)q * 8 - T) * (int64_t)fb_src;
                in_ptr = s_in + (uint32_t)(q & 1) * 256 * in_stride + row * in_stride + (uint32_t)(gq & 15);
            }
            // ---- advance: frame (n_start + a) enters slot s ----
            {
                int32_t x[CPL];
                lds_load_frame<CPL>(in_ptr + (uint32_t)(s & 7) * fb_src, sb, src_le, x);
                const bool before_start = (n_start + a) < 0;          // stream start: history is zeros
#pragma unroll
                for (int c = 0; c < CPL; c++) win[s][c] = before_start ? 0.0 : (double)x[c];
            }
            // ---- emit the outputs whose newest input frame is this one: floor(t / L) == a ----
            while (t < L * (a + 1)) {
                const int phase = __builtin_amdgcn_readfirstlane(t - L * a);
                const const_f64_ptr_t cp = coef_c + (size_t)phase * T;
                double acc[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) acc[c] = 0.0;
#pragma unroll
                for (int k = 0; k < T; k++) {
                    const double ck = cp[k];
#pragma unroll
                    for (int c = 0; c < CPL; c++) acc[c] = fma(ck, win[(s - k + T) % T][c], acc[c]);
                }
                int32_t* e = s_ring + (((uint32_t)j & (RING - 1)) * 256 + row) * CPL;
#pragma unroll
                for (int c = 0; c < CPL; c++) e[c] = src_round_s24(acc[c]);
                j++;
                t += M;
            }
        }, std::make_integer_sequence<int, T>{});
    }
    drain(j);
}

bool src_block_supported(uint32_t T, uint32_t cpl)
{
    return cpl == 2 && T == 32;
}

template <int T, int CPL, int DB>
static hipError_t launch_one(const ohgpu_batch* b, const SrcFastParams& prm, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute((const void*)src_block_kernel<T, CPL, DB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)b->fast.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((src_block_kernel<T, CPL, DB>), dim3(b->fast.n_work), dim3(256), b->fast.lds_bytes, s,
                       prm.segs, prm.msgs, prm.work, prm.coef, prm.ramp_table, prm.src, prm.dst, prm);
    return hipGetLastError();
}

hipError_t launch_src_block(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    if (!b->fast.enabled || b->fast.n_work == 0) return hipSuccess;
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    prm.ramp_table = ctx->d_ramp_table;
    const uint32_t T = b->fast.T, cpl = b->fast.cpl, db = prm.db;
    if (T == 32 && cpl == 2) {
        if (db == 3) return launch_one<32, 2, 3>(b, prm, s);
        if (db == 2) return launch_one<32, 2, 2>(b, prm, s);
        if (db == 4) return launch_one<32, 2, 4>(b, prm, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace ohgpu
