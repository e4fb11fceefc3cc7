// src_lean_kernel.hip -- round 2's resample -> ramp -> pack kernel ("lean block kernel"), the one bench.py times.
//
// Same mapping as src_block_kernel.hip (one lane = one channel of one phase-aligned block, the whole
// [L][T] coefficient table in LDS, tap k = ONE v_fmac_f64_dpp row_newbcast, a T-sample window in registers, input
// staged by global->LDS DMA, packed output drained from an LDS ring as whole 64-byte lines), with the per-output
// work that is not a tap cut down -- round 1's loop spent more issue slots around the taps than on them:
//   * PLAIN and RAMPED units.  The planner tells a unit whose output range meets no ramped message from one that
//     does.  A plain unit runs a body with no message state at all: no cursor, no event vote, no table in LDS.
//     Only a ramped unit carries the per-lane message cursor and RampApplicator's arithmetic.
//   * Rounding by bias.  The accumulator starts at 2^24 + 0.5, so the sum is positive and v_cvt_u32_f64's
//     truncation IS floor(sum + 0.5); the S24 value is then the low 24 bits and the clamp is one v_med3_u32.
//     (Needs sum|c| < 2^29 per phase so that 2^24 + |sum| stays below 2^25: 53 bits with the 28 fraction bits.
//     Filters that break it run on src_block_kernel.)
//   * The window holds samples x 256: ONE v_perm_b32 moves the subsample's bytes (either byte order, any alignment)
//     to the top of a dword and ONE v_cvt_f64_i32 converts; the 2^-8 rides in the coefficients' scale (2^-36).
//   * 24-bit stereo pairs are exchanged with two bank-masked DPP moves; the odd output alone packs and stores.
//   * Staging and write-back address with a wave-uniform 64-bit base in SGPRs plus a per-lane 32-bit offset that is
//     computed once per unit: no vector address arithmetic per stage or per line.
//   * Every LDS wait of the per-output loop is the same counted wait, lgkmcnt(NCR - 1) (derivation at the loop).
// Bit-exact against the integer model (oracle/ohp_pipeline.c) like its predecessor: the arithmetic is the same exact
// fp64 sum, only its representation (bias, x 256) differs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_block_common.h"

#pragma clang diagnostic ignored "-Winline-asm"     // (m0 in clobber lists: the staging DMA sets it by hand)

namespace ohgpu {

// ---- hand-issued instructions of the per-output loop ----
// The loop's LDS traffic is issued from inline asm so that the compiler's wait insertion does not see it (it would
// wait with lgkmcnt(0), i.e. for the prefetches just issued as well); the waits are counted by hand.  A wave's LDS
// operations complete in issue order, so lgkmcnt(N) retires all but the N youngest.  Every statement is volatile
// with a "memory" clobber: they keep their order among themselves and against compiler-issued memory operations.
template <int BYTES>
__device__ __forceinline__ void lean_issue_f64(double& dst, uint32_t addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(BYTES) : "memory");
}
template <int DW>
__device__ __forceinline__ void lean_issue_2xu32(uint64_t& dst, uint32_t addr)
{
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "i"(DW), "i"(DW + 1) : "memory");
}
template <int N>
__device__ __forceinline__ void lean_wait(double& x)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lean_wait(double& x, uint64_t& y)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "i"(N) : "memory");
}
__device__ __forceinline__ void lean_wait0(uint64_t& y)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y) : : "memory");
}

// sixteen taps of one coefficient register: taps 16 r + 15 .. 16 r, alternating between the two chains.  `w` are the
// sixteen window slots in tap order 15 .. 0.  (cv is only ever written by an LDS load: no VALU-write -> DPP-read hazard.)
#define OHGPU_FM(acc, k, x) "v_fmac_f64_dpp " acc ", %[cv], " x " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void lean_taps16(double& acc0, double& acc1, const double cv,
                                            const double w15, const double w14, const double w13, const double w12,
                                            const double w11, const double w10, const double w9, const double w8,
                                            const double w7, const double w6, const double w5, const double w4,
                                            const double w3, const double w2, const double w1, const double w0)
{
    asm volatile(
        OHGPU_FM("%[a1]", 15, "%[w15]") OHGPU_FM("%[a0]", 14, "%[w14]") OHGPU_FM("%[a1]", 13, "%[w13]") OHGPU_FM("%[a0]", 12, "%[w12]")
        OHGPU_FM("%[a1]", 11, "%[w11]") OHGPU_FM("%[a0]", 10, "%[w10]") OHGPU_FM("%[a1]", 9, "%[w9]") OHGPU_FM("%[a0]", 8, "%[w8]")
        OHGPU_FM("%[a1]", 7, "%[w7]") OHGPU_FM("%[a0]", 6, "%[w6]") OHGPU_FM("%[a1]", 5, "%[w5]") OHGPU_FM("%[a0]", 4, "%[w4]")
        OHGPU_FM("%[a1]", 3, "%[w3]") OHGPU_FM("%[a0]", 2, "%[w2]") OHGPU_FM("%[a1]", 1, "%[w1]") OHGPU_FM("%[a0]", 0, "%[w0]")
        : [a0] "+v"(acc0), [a1] "+v"(acc1)
        : [cv] "v"(cv), [w15] "v"(w15), [w14] "v"(w14), [w13] "v"(w13), [w12] "v"(w12), [w11] "v"(w11), [w10] "v"(w10), [w9] "v"(w9),
          [w8] "v"(w8), [w7] "v"(w7), [w6] "v"(w6), [w5] "v"(w5), [w4] "v"(w4), [w3] "v"(w3), [w2] "v"(w2), [w1] "v"(w1), [w0] "v"(w0));
}

// One subsample -> sample x 256 as an exact double: the two aligned words that hold it (LDS accepts unaligned reads
// but serialises them lane by lane), one byte permute with the lane's selector (alignment, byte order and the
// left-justification in one), one conversion.
__device__ __forceinline__ double lean_unpack(const uint64_t words, const uint32_t sel)
{
    uint32_t w;
    double d;
    asm volatile("v_perm_b32 %0, %2, %3, %4\n\tv_cvt_f64_i32 %1, %0"
                 : "=&v"(w), "=v"(d) : "v"((uint32_t)(words >> 32)), "v"((uint32_t)words), "v"(sel));
    return d;
}
// selector of lean_unpack for a subsample whose first byte sits `sh` bytes into the low word ({hi, lo} = bytes 7..0):
// result bytes 3..(4-SB) = the subsample most significant byte first, the rest zero (0x0c)
template <int SB, bool LE>
__device__ __forceinline__ uint32_t lean_unpack_sel(uint32_t sh)
{
    uint32_t sel = 0x0cu;                    // result byte 0: zero
#pragma unroll
    for (int b = 1; b < 4; b++) {            // result byte b holds the S24 value's byte b - 1; the value is the subsample left-justified to 24 bits
        const int from_msb = 3 - b;          // 0 = the subsample's most significant byte
        uint32_t pick = 0x0cu;
        if (from_msb < SB) pick = sh + (LE ? (uint32_t)(SB - 1 - from_msb) : (uint32_t)from_msb);
        sel |= pick << (8 * b);
    }
    return sel;
}

template <int T, int CH, int SB, int DB>
struct LeanGeom {
    static constexpr int BPW = 64 / CH;
    static constexpr int ROWS = BPW;
    static constexpr int MAX_WAVES = T <= 32 ? 12 : 8;
    static constexpr int MSG_SLOTS = 32;
    static constexpr int FB_SRC = CH * SB, FB_DST = CH * DB;
    // 16-byte pieces per staged row: the eight frames of a stage at any alignment ((15 + 8 FB_SRC) bytes), odd for the bank spread
    static constexpr int IN_BLOCKS = lean_in_blocks(CH, SB);
    static constexpr int IN_STRIDE = IN_BLOCKS * 16;
    static constexpr int IN_ITERS = (ROWS * IN_BLOCKS + 63) / 64;
};

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE>
__global__ __launch_bounds__((LeanGeom<T, CH, SB, DB>::MAX_WAVES * 64))
void src_lean_kernel(const SrcSeg* __restrict__ segs, const SegMsg* __restrict__ msgs, const SrcWork* __restrict__ work,
                     const uint32_t n_work, const double* __restrict__ coef, const uint16_t* __restrict__ ramp_table,
                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                     const uint64_t src_arena_bytes, const int L, const int M, const uint32_t L_blk, const uint32_t M_blk,
                     const uint32_t ring_bytes, uint32_t* __restrict__ unit_counter)
{
    static_assert(T % 16 == 0 && T >= 32 && T <= 64, "T / 16 coefficient registers per lane");
    using G = LeanGeom<T, CH, SB, DB>;
    constexpr int NCR = T / 16;
    constexpr int BPW = G::BPW, ROWS = G::ROWS, MSG_SLOTS = G::MSG_SLOTS;
    constexpr int FB_SRC = G::FB_SRC, FB_DST = G::FB_DST;
    constexpr int IN_BLOCKS = G::IN_BLOCKS, IN_STRIDE = G::IN_STRIDE, IN_ITERS = G::IN_ITERS;
    constexpr uint32_t OFF_IN = 0, OFF_MSG = OFF_IN + 2 * ROWS * IN_STRIDE, OFF_MSGM = OFF_MSG + MSG_SLOTS * 16, OFF_RING = OFF_MSGM + MSG_SLOTS * 4;
    constexpr bool PAIR = ring_pair_mode(CH, DB);
    static_assert(DB >= 2 && DB <= 4, "destination depths 16 / 24 / 32 bit");
    static_assert((8 * FB_SRC) % 16 == 0, "a stage advances every piece by a whole number of 16-byte pieces");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t table_bytes = (uint32_t)L * T * 8;
    const uint32_t coef_bytes = table_bytes + kRampLdsBytes;
    const uint32_t row_stride = ring_bytes + 4;
    const uint32_t ring_area = (ROWS * row_stride + 15) & ~15u;
    const uint32_t wave_lds = OFF_RING + ring_area + 64;               // + a slot for the stores of lanes beyond the last whole block

    // ---- coefficient table -> LDS once per workgroup, scaled by 2^-36 (exact): the window holds samples x 256, so the
    // accumulator is in sample units with 28 fraction bits.
    for (uint32_t i = tid; i < (uint32_t)L * T; i += blockDim.x)
        ((__attribute__((address_space(3))) double*)(lds_u8_t)smem)[i] = coef[i] * (1.0 / 68719476736.0);
    const __attribute__((address_space(3))) uint16_t* const ramp_lds =
        (const __attribute__((address_space(3))) uint16_t*)((lds_u8_t)smem + table_bytes);
    for (uint32_t i = tid; i < kRampLdsBytes / 4; i += blockDim.x)
        ((__attribute__((address_space(3))) uint32_t*)((lds_u8_t)smem + table_bytes))[i] = ((const uint32_t*)ramp_table)[i];
    __syncthreads();
    const uint32_t coef_lane = (uint32_t)(uintptr_t)((lds_u8_t)smem + (lane & 15) * 8);
    uint8_t* const wsmem = smem + coef_bytes + wave * wave_lds;
    const lds_u8_t lds = (lds_u8_t)wsmem;
    const uint32_t wave_lds_addr = (uint32_t)(uintptr_t)lds;
    const int Mr = M % L;

    const uint32_t bw = lane / CH;                       // block within the wave's unit
    const uint32_t c = lane - bw * CH;                   // this lane's channel
    const uint32_t row = bw;
    const bool lane_block = bw < (uint32_t)BPW;         // (64 % CH lanes at the end of the wave own no block)

    // pair mode selectors for v_perm_b32 {got (bytes 4-7), own (bytes 0-3)}: words 0/1 from the even frame, words 1/2 from the odd one
    constexpr uint32_t B0 = DST_LE ? 0 : 2, B1 = 1, B2 = DST_LE ? 2 : 0;      // byte of the S24 value that is memory byte 0, 1, 2
    const uint32_t sel_lo = c == 0 ? (B0 | B1 << 8 | B2 << 16 | (4 + B0) << 24) : (B1 | B2 << 8 | (4 + B0) << 16 | (4 + B1) << 24);
    const uint32_t sel_hi = c == 0 ? ((4 + B1) | (4 + B2) << 8 | B0 << 16 | B1 << 24) : ((4 + B2) | B0 << 8 | B1 << 16 | B2 << 24);
    const uint32_t dummy_lane = wave_lds_addr + OFF_RING + ring_area + (lane & 3) * 16;
    const uint32_t ring_lane0 = wave_lds_addr + OFF_RING + row * row_stride + (PAIR ? c * 4 : c * DB);
    const uint32_t ring_lane = lane_block ? ring_lane0 : dummy_lane;
    const double bias = 16777216.5;                     // 2^24 + 0.5
    const uint32_t clamp_lo = 0x00800000u, clamp_hi = 0x017fffffu;   // 2^24 - 2^23 .. 2^24 + 2^23 - 1

    const uint32_t first_claimed = gridDim.x * n_waves;
    uint32_t unit = blockIdx.x * n_waves + wave;
    while (unit < n_work) {
    uint32_t claim = 0;
    if (lane == 0) claim = atomicAdd(unit_counter, 1u);
    const SrcWork wk = work[unit];
    const SrcSeg seg = segs[wk.seg];
    const uint32_t n_blocks = wk.n_blocks;
    const bool ramped = (wk.flags & kWorkRamped) != 0;                 // wave-uniform
    const bool checked = (wk.flags & kWorkChecked) != 0;               // some staging piece of the unit lies outside the arena
    const bool lane_valid = lane_block && row < n_blocks;
    const uint64_t blk = wk.first_block + row;
    const int64_t n_start = (int64_t)(blk * M_blk);
    const int64_t row_g = seg.src_base + (n_start - T) * (int64_t)FB_SRC;   // byte offset of the row's frame at a_lin = 0
    const bool first_block = n_start == 0;

    // ---- messages: only a ramped unit looks at them (the table, the cursor and the arithmetic are round 1's) ----
    const __attribute__((address_space(3))) u32x4* msg_tab = (const __attribute__((address_space(3))) u32x4*)(lds + OFF_MSG);
    const uint64_t wave_m0 = wk.first_block * (uint64_t)L_blk;
    const uint32_t tab_lo = wk.msg_first;
    const int32_t lane_off = (int32_t)(bw * L_blk);
    uint32_t mi = 0;
    int32_t msg_rel0 = 0;
    uint32_t msg_n = 0x7fffffffu, msg_ramp = 0, msg_flags = 0, msg_m = 0;
    int32_t evt_j = 0x7fffffff;
    auto load_msg = [&](uint32_t idx) __attribute__((always_inline)) {
        if (idx < (uint32_t)MSG_SLOTS) {
            const u32x4 e = msg_tab[idx];
            msg_rel0 = (int32_t)e.x - lane_off; msg_n = e.y; msg_ramp = e.z; msg_flags = e.w;
            if (msg_flags & OHGPU_FLAG_RAMP) msg_m = ((const __attribute__((address_space(3))) uint32_t*)(lds + OFF_MSGM))[idx];
        } else {
            const SegMsg m = msgs[tab_lo + idx];
            msg_rel0 = (int32_t)(int64_t)(m.out0 - wave_m0) - lane_off;
            msg_n = m.n; msg_ramp = (uint32_t)m.ramp_start | ((uint32_t)m.ramp_end << 16); msg_flags = (uint32_t)m.flags | ((uint32_t)m.s_n1 << 8);
            msg_m = m.m_n1;
        }
    };
    if (ramped) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < (uint32_t)MSG_SLOTS) {
            uint32_t em = 0;
            u32x4 e = {0x7fffffffu, 0u, 0u, 0u};
            if (tab_lo + lane < seg.msg_end) {
                const SegMsg m = msgs[tab_lo + lane];
                e.x = (uint32_t)(int32_t)(int64_t)(m.out0 - wave_m0);
                e.y = m.n;
                e.z = (uint32_t)m.ramp_start | ((uint32_t)m.ramp_end << 16);
                e.w = (uint32_t)m.flags | ((uint32_t)m.s_n1 << 8);
                em = m.m_n1;
            }
            ((__attribute__((address_space(3))) u32x4*)(lds + OFF_MSG))[lane] = e;
            ((__attribute__((address_space(3))) uint32_t*)(lds + OFF_MSGM))[lane] = em;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane_valid) {
            uint32_t cnt = 0;
#pragma unroll
            for (int idx = 0; idx < MSG_SLOTS; idx++) cnt += ((int32_t)msg_tab[idx].x <= lane_off) ? 1u : 0u;
            mi = cnt ? cnt - 1 : 0;
            load_msg(mi);
            while ((uint32_t)(0 - msg_rel0) >= msg_n) load_msg(++mi);
        }
        evt_j = (msg_flags & OHGPU_FLAG_RAMP) ? 0 : msg_rel0 + (int32_t)msg_n;
    }

    // ---- input staging.  Stage q holds advances [8q - T, 8q + 8 - T) of every row as raw packed bytes: the aligned
    // 16-byte pieces that cover them, IN_BLOCKS per row, rows side by side.  Piece idx = it*64 + lane (row idx /
    // IN_BLOCKS) is moved by lane `lane` of DMA instruction `it`: its source is stage_base (wave-uniform, in SGPRs,
    // + 8 frames per stage) + piece_off[it] (per lane, fixed for the unit).  A row whose first frame sits early in its
    // first piece does not need its last piece: that lane re-reads the row's first piece instead (same line, no traffic).
    const int total = (int)M_blk + T;         // advances a = a_lin - T for a_lin in [0, total)
    const int n_stages = (total + 7) >> 3;
    const int64_t g0 = seg.src_base + ((int64_t)(wk.first_block * M_blk) - T) * (int64_t)FB_SRC;    // row 0's frame at a_lin = 0
    const uint32_t a0 = (uint32_t)g0 & 15u;
    int64_t stage_off = g0 - (int64_t)a0;     // arena offset of row 0's first piece of the NEXT stage to issue (wave-uniform)
    uint32_t piece_off[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; it++) {
        const uint32_t idx = it * 64 + lane;
        uint32_t r = idx / IN_BLOCKS;
        uint32_t part = idx - r * IN_BLOCKS;
        if (r >= n_blocks || r >= (uint32_t)ROWS) { r = 0; part = 0; }
        const uint32_t d_r = r * M_blk * FB_SRC;
        const uint32_t al = (a0 + d_r) & 15u;
        const uint32_t pieces_needed = (al + 8 * FB_SRC + 15) >> 4;      // bytes al .. al + 8 FB_SRC - 1 of the row's first piece onwards
        if (part >= pieces_needed) part = 0;
        piece_off[it] = d_r - al + a0 + 16 * part;
    }
    auto issue_stage = [&](int q) __attribute__((always_inline)) {
        const uint32_t buf = OFF_IN + (uint32_t)(q & 1) * ROWS * IN_STRIDE;
#pragma unroll
        for (int it = 0; it < IN_ITERS; it++) {
            constexpr int kTail = ROWS * IN_BLOCKS - (IN_ITERS - 1) * 64;     // lanes of the last instruction that own a piece
            const bool last_partial = it == IN_ITERS - 1 && kTail < 64;
            const uint32_t m0v = wave_lds_addr + buf + (uint32_t)(it * 64) * 16;
            if (!checked) {
                const uint64_t sbase = (uint64_t)(uintptr_t)src + (uint64_t)stage_off;
                if (last_partial) {
                    asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                                 : : "v"(piece_off[it]), "s"(sbase), "s"(m0v), "s"((1ull << (kTail & 63)) - 1ull) : "memory", "m0");
                } else {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 : : "v"(piece_off[it]), "s"(sbase), "s"(m0v) : "memory", "m0");
                }
            } else {
                // a unit at an end of the arena: a piece that straddles the end is copied byte by byte (the bytes that exist), one
                // that lies wholly outside is not moved -- its bytes are never used (frames before a stream's first read as zeros:
                // the window of a first block is cleared below; frames beyond the last only feed advances past the last output)
                const int64_t addr = stage_off + (int64_t)piece_off[it];
                const bool own = !last_partial || lane < (uint32_t)(kTail & 63);
                if (own) {
                    if (addr >= 0 && (uint64_t)addr + 16 <= src_arena_bytes) {
                        __builtin_amdgcn_global_load_lds((global_ptr_t)(src + addr), (lds_ptr_t)(wsmem + buf + (uint32_t)(it * 64) * 16), 16, 0, 0);
                    } else if (addr > -16 && addr < (int64_t)src_arena_bytes) {
                        const lds_u8_t d = lds + buf + (uint32_t)(it * 64 + lane) * 16;
                        for (int b = 0; b < 16; b++) {
                            const int64_t a1 = addr + b;
                            d[b] = (a1 >= 0 && (uint64_t)a1 < src_arena_bytes) ? src[a1] : (uint8_t)0;
                        }
                    }
                }
            }
        }
        stage_off += 8 * FB_SRC;
    };

    // ---- write-back: the block's output is a byte stream of L_blk*FB_DST bytes.  A finished subsample is packed and stored at
    // its place in the row's byte ring; whenever a 64-byte line of the stream is complete the wave writes that line of all its
    // blocks, lane l of pass `it` copying 16-byte piece (l & 3) of block (it*16 + l/4): every HBM write is a whole line.
    const int64_t wave_dst = seg.dst_base + (int64_t)(wk.first_block * L_blk) * FB_DST;
    const uint32_t wave_rows = n_blocks < (uint32_t)BPW ? n_blocks : (uint32_t)BPW;
    constexpr int DRAIN_ITERS = (BPW * 4 + 63) / 64;
    uint32_t drain_off[DRAIN_ITERS];          // per lane: its piece's offset from the unit's current line
    uint32_t drain_lds[DRAIN_ITERS];          // per lane: its row's ring
    bool drain_on[DRAIN_ITERS];
#pragma unroll
    for (int it = 0; it < DRAIN_ITERS; it++) {
        const uint32_t piece = it * 64 + lane;
        const uint32_t r = piece >> 2, part = piece & 3;
        drain_on[it] = r < wave_rows;
        drain_off[it] = (drain_on[it] ? r : 0u) * L_blk * FB_DST + part * 16;
        drain_lds[it] = OFF_RING + (drain_on[it] ? r : 0u) * row_stride;
    }
    uint32_t drained = 0;                                 // lines written so far (wave-uniform)
    uint32_t line_pos = 0;                                // ring position of line `drained`
    auto drain = [&](int frames_stored) __attribute__((always_inline)) {
        while (drained < (((uint32_t)frames_stored * FB_DST) >> 6)) {
            uint8_t* const line = dst + wave_dst + (uint64_t)drained * 64;        // wave-uniform
#pragma unroll
            for (int it = 0; it < DRAIN_ITERS; it++) {
                if (drain_on[it]) {
                    uint32_t pos = line_pos + ((it * 64 + lane) & 3) * 16;
                    if (pos >= ring_bytes) pos -= ring_bytes;
                    const __attribute__((address_space(3))) uint32_t* q =
                        (const __attribute__((address_space(3))) uint32_t*)(lds + drain_lds[it] + pos);
                    u32x4 v4;
                    v4.x = q[0]; v4.y = q[1]; v4.z = q[2]; v4.w = q[3];
                    __builtin_nontemporal_store(v4, (u32x4*)(line + drain_off[it]));
                }
            }
            drained++;
            line_pos += 64;
            if (line_pos >= ring_bytes) line_pos -= ring_bytes;
        }
    };

    double win[T];
#pragma unroll
    for (int s = 0; s < T; s++) win[s] = 0.0;

    int j = 0;                                // outputs emitted so far (wave-uniform)
    int t = 0;                                // j * M
    int p = 0;                                // phase of output j
    uint32_t ring_pos = 0;                    // ring position of frame j (pair mode: of the pair)
    constexpr int PH = (FB_SRC % 4 == 0) ? 1 : ((FB_SRC % 2 == 0) ? 2 : 4);   // frames s' and s' + PH share their place in a dword
    static_assert((FB_SRC * PH) % 4 == 0 && (8 % PH) == 0 && FB_SRC * 7 / 4 + 1 < 256, "immediate dword offsets of ds_read2_b32");
    const uint32_t in_base = wave_lds_addr + OFF_IN + row * IN_STRIDE + ((uint32_t)row_g & 15u) + c * SB;   // frame 0 of buffer 0
    uint32_t in_sel[PH];                      // byte selector of frame ph's subsample (the same in every stage)
    uint32_t in_addr[2][PH];                  // aligned LDS address of frame ph's dword, by buffer
#pragma unroll
    for (int ph = 0; ph < PH; ph++) {
        in_sel[ph] = lean_unpack_sel<SB, SRC_LE>((in_base + ph * FB_SRC) & 3u);
        in_addr[0][ph] = (lane_block ? ((in_base + ph * FB_SRC) & ~3u) : (wave_lds_addr + OFF_IN));
        in_addr[1][ph] = in_addr[0][ph] + ROWS * IN_STRIDE;
    }
    const bool any_first = __any(first_block) != 0;

    // the pending store: the bytes of the last finished output (pair), stored by the NEXT output after its first wait
    uint32_t st_addr = dummy_lane, st_lo = 0, st_hi = 0;
    uint32_t y_even = 0;
    bool pend = false;                        // (wave-uniform)
    int stored = 0;                           // frames whose bytes are in the ring or pending
    auto issue_store = [&]() __attribute__((always_inline)) {
        if constexpr (PAIR) asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(st_addr), "v"(st_lo), "v"(st_hi) : "memory");
        else if constexpr (DB == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(st_addr), "v"(st_lo) : "memory");
        else if constexpr (DB == 3) asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8_d16_hi %0, %1 offset:2"
                                                 : : "v"(st_addr), "v"(st_lo), "v"(st_lo >> 8) : "memory");
        else asm volatile("ds_write_b16 %0, %1" : : "v"(st_addr), "v"(st_lo) : "memory");
    };

    // coefficients of output j: register r = taps 16 r .. 16 r + 15, tap k in lane (k & 15) of every 16-lane row
    double cf[NCR];
    static_for([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = NCR - 1 - decltype(rc)::value;
        lean_issue_f64<r * 128>(cf[r], coef_lane);
    }, std::make_integer_sequence<int, NCR>{});
    uint64_t raw = 0;

    issue_stage(0);
    issue_stage(1);

    // ---- warm-up: the T advances before the block's first output only fill the window; two stages ahead ----
    static_for([&](auto stage) __attribute__((always_inline)) {
        constexpr int q = decltype(stage)::value;
        if constexpr ((q & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for([&](auto half) __attribute__((always_inline)) {
            constexpr int h = decltype(half)::value;
            uint64_t r4[4];
            static_for([&](auto k4) __attribute__((always_inline)) {
                constexpr int sp = 4 * h + decltype(k4)::value, ph = sp % PH;
                lean_issue_2xu32<FB_SRC * (sp - ph) / 4>(r4[sp & 3], in_addr[q & 1][ph]);
            }, std::make_integer_sequence<int, 4>{});
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]) : : "memory");
#pragma unroll
            for (int k = 0; k < 4; k++) win[8 * q + 4 * h + k] = lean_unpack(r4[k], in_sel[(4 * h + k) % PH]);
        }, std::make_integer_sequence<int, 2>{});
        if constexpr (q + 2 <= T / 8) {
            if ((q + 2) * 8 < total) issue_stage(q + 2);
        }
    }, std::make_integer_sequence<int, T / 8>{});
    if (any_first) {
#pragma unroll
        for (int s = 0; s < T; s++) win[s] = first_block ? 0.0 : win[s];
    }
    // The coefficient reads above were issued before the warm-up's waits: they have landed.  From here on, in issue order:
    //   per advance:  R (raw sample)
    //   per output:   W  [S: the previous output's pending store]  16 taps(c[NCR-1])  C'[NCR-1]
    //                 W  16 taps(c[NCR-2])  C'[NCR-2] ... W  unpack  16 taps(c[0])  C'[0]   round, clamp, (ramp,) pack
    // Every W awaits a reload C'[r] issued one output earlier (the last one the raw sample R as well).  After C'[r] come at
    // least the other NCR - 1 reloads (the rest of that output's, then this output's earlier ones) whatever else -- R, S --
    // was issued in between, and R, when there is one, is followed by the same: lgkmcnt(NCR - 1) is enough for every W on
    // every path.  Where more has been issued the wait also covers operations issued long before it (never the youngest).
    for (int g = 1; g * T < total; g++) {
        static_for([&](auto slot) __attribute__((always_inline)) {
            constexpr int s = decltype(slot)::value;
            const int a_lin = g * T + s;
            if (a_lin >= total) return;
            const int a = a_lin - T;
            if constexpr ((s & 3) == 0) {
                if constexpr ((s & 7) == 0) {
                    const int q = a_lin >> 3;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage q has landed (this wave issued all of it)
                    if ((q + 1) * 8 < total) issue_stage(q + 1);
                }
                if (pend) { issue_store(); pend = false; }
                drain(stored);
            }
            // ---- advance: this channel's sample of frame (n_start + a) enters slot s ----
            {
                constexpr int sp = s & 7, ph = sp % PH;
                lean_issue_2xu32<FB_SRC * (sp - ph) / 4>(raw, in_addr[(s >> 3) & 1][ph]);
            }
            if (!(t < L * (a + 1))) {                                   // no output needs it yet (M > L)
                lean_wait0(raw);
                win[s] = lean_unpack(raw, in_sel[(s & 7) % PH]);
            }
            // ---- emit the outputs whose newest input frame is this one: floor(t / L) == a ----
            while (t < L * (a + 1)) {
                double acc0, acc1;
                asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, 0" : "=&v"(acc0), "=&v"(acc1) : "s"(bias));
                p += Mr;                                                       // the next output's phase
                if (p >= L) p -= L;
                const uint32_t cp = coef_lane + (uint32_t)p * (T * 8);
                static_for([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = NCR - 1 - decltype(rc)::value;          // highest taps (oldest samples) first, the newest sample last
                    if constexpr (r == 0) {
                        lean_wait<NCR - 1>(cf[0], raw);
                        win[s] = lean_unpack(raw, in_sel[(s & 7) % PH]);
                    } else {
                        lean_wait<NCR - 1>(cf[r]);
                    }
                    if constexpr (r == NCR - 1) {
                        if (pend) { issue_store(); pend = false; }
                    }
#define W_(k) win[(s - (16 * r + (k)) + 2 * T) % T]
                    lean_taps16(acc0, acc1, cf[r], W_(15), W_(14), W_(13), W_(12), W_(11), W_(10), W_(9), W_(8),
                                W_(7), W_(6), W_(5), W_(4), W_(3), W_(2), W_(1), W_(0));
#undef W_
                    lean_issue_f64<r * 128>(cf[r], cp);
                }, std::make_integer_sequence<int, NCR>{});
                uint32_t y;
                {
                    double sum;
                    asm volatile("v_add_f64 %0, %2, %3\n\tv_cvt_u32_f64 %1, %0\n\tv_med3_u32 %1, %1, %4, %5"
                                 : "=&v"(sum), "=&v"(y) : "v"(acc0), "v"(acc1), "s"(clamp_lo), "v"(clamp_hi));
                }
                // y = 2^24 + the clamped S24 value: its low 24 bits are the value in two's complement
                if (ramped) {
                    if (__any(j >= evt_j) != 0) {                               // message boundary or ramping somewhere in the wave
                        if (lane_valid && j >= evt_j) {
                            while ((uint32_t)(j - msg_rel0) >= msg_n) load_msg(++mi);
                            if (msg_flags & OHGPU_FLAG_RAMP) {
                                const uint32_t rs = msg_ramp & 0xffffu, re = msg_ramp >> 16;
                                const uint32_t mult = ramp_lds[ramp_index_magic(rs, (int32_t)(rs - re), (uint32_t)(j - msg_rel0), msg_n, msg_m, (msg_flags >> 8) & 31u)];
                                y = ramp_word(y << 8, mult, 3, CH, c) >> 8;
                                evt_j = j + 1;
                            } else {
                                evt_j = msg_rel0 + (int32_t)msg_n;
                            }
                        }
                    }
                }
                if constexpr (PAIR) {
                    if (j & 1) {
                        // lane A (channel 0) needs B's even value, lane B needs A's odd one: every lane offers what its partner
                        // wants, one quad-permuted move fetches it (two wait states between the offer's write and its DPP read)
                        uint32_t give, got;
                        asm volatile("v_cndmask_b32_e64 %[give], %[ye], %[y], %[m]\n\t"
                                     "v_add_u32 %[sta], %[rp], %[rl]\n\t"
                                     "s_nop 0\n\t"
                                     "v_mov_b32_dpp %[got], %[give] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                     "v_perm_b32 %[lo], %[got], %[ye], %[sl]\n\t"
                                     "v_perm_b32 %[hi], %[got], %[y], %[sh]"
                                     : [give] "=&v"(give), [got] "=&v"(got), [sta] "=&v"(st_addr), [lo] "=&v"(st_lo), [hi] "=&v"(st_hi)
                                     : [ye] "v"(y_even), [y] "v"(y), [m] "s"(0x5555555555555555ull), [rp] "s"(ring_pos), [rl] "v"(ring_lane),
                                       [sl] "v"(sel_lo), [sh] "v"(sel_hi));
                        ring_pos += 2 * FB_DST;
                        if (ring_pos == ring_bytes) ring_pos = 0;
                        pend = true;
                        stored = j + 1;
                    } else {
                        y_even = y;
                    }
                } else {
                    const uint32_t w = y << 8;                                  // left-justified (a11)
                    st_lo = DST_LE ? (w >> (32 - 8 * DB)) : __builtin_bswap32(w);
                    st_addr = ring_lane + ring_pos;
                    ring_pos += FB_DST;
                    if (ring_pos == ring_bytes) ring_pos = 0;
                    pend = true;
                    stored = j + 1;
                }
                j++;
                t += M;
            }
        }, std::make_integer_sequence<int, T>{});
    }
    if (pend) { issue_store(); pend = false; }
    drain(j);
    unit = first_claimed + (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
    }   // units
    // The counters reset themselves: a wave reports in after its last claim, and the last wave of the grid zeroes both.
    if (lane == 0) {
        const uint32_t waves_total = gridDim.x * n_waves;
        if (atomicAdd(unit_counter + 1, 1u) == waves_total - 1) {
            __hip_atomic_store(unit_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(unit_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


#define OHGPU_LEAN_ARGS const SrcSeg*, const SegMsg*, const SrcWork*, uint32_t, const double*, const uint16_t*, const uint8_t*, \
                        uint8_t*, uint64_t, int, int, uint32_t, uint32_t, uint32_t, uint32_t*
#define X_DEFINE(t, c, s_, sl, d, dl) template __global__ void src_lean_kernel<t, c, s_, sl, d, dl>(OHGPU_LEAN_ARGS);
#define X_DECLARE(t, c, s_, sl, d, dl) extern template __global__ void src_lean_kernel<t, c, s_, sl, d, dl>(OHGPU_LEAN_ARGS);
#if defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 2
OHGPU_BLOCK_KERNELS_2(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 3
OHGPU_BLOCK_KERNELS_3(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART)
OHGPU_BLOCK_KERNELS_2(X_DECLARE)
OHGPU_BLOCK_KERNELS_3(X_DECLARE)
#endif

#if !defined(OHGPU_BLOCK_PART) || OHGPU_BLOCK_PART == 1
// Geometry the planner needs (must match the kernel's constexprs).  Returns false when the layout does not fit the CU's LDS.
bool src_lean_geometry(uint32_t L, uint32_t T, uint32_t ch, uint32_t sb, uint32_t db, uint32_t out_per_drain,
                       uint32_t* rows, uint32_t* in_blocks, uint32_t* ring_bytes, uint32_t* coef_lds_bytes, uint32_t* wave_lds_bytes, uint32_t* max_waves)
{
    const uint32_t bpw = 64 / ch;
    const uint32_t fb_dst = ch * db;
    const uint32_t inb = (uint32_t)lean_in_blocks((int)ch, (int)sb);
    const uint32_t rb = ring_bytes_for(fb_dst, out_per_drain, ring_pair_mode(ch, db));
    *rows = bpw;
    *in_blocks = inb;
    *ring_bytes = rb;
    *coef_lds_bytes = L * T * 8 + kRampLdsBytes;
    *wave_lds_bytes = 2 * bpw * inb * 16 + 32 * 16 + 32 * 4 + ((bpw * (rb + 4) + 15) & ~15u) + 64;
    const uint32_t budget = 160 * 1024;
    if (*coef_lds_bytes + *wave_lds_bytes > budget) return false;
    uint32_t w = (budget - *coef_lds_bytes) / *wave_lds_bytes;
    const uint32_t cap = T <= 32 ? 12u : 8u;
    if (w > cap) w = cap;
    if (w < 4) return false;
#ifdef OHGPU_DIAG
    if (const char* e = getenv("OHGPU_DIAG_MAX_WAVES")) { const uint32_t x = (uint32_t)atoi(e); if (x >= 4 && x < w) w = x; }   // (diagnostic builds: occupancy)
#endif
    *max_waves = w;
    return true;
}

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE>
static hipError_t launch_lean_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_lean_kernel<T, CH, SB, SRC_LE, DB, DST_LE>;
    const SrcFastPlan& f = b->fast;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t w = (f.n_work + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > f.lean_max_waves) w = f.lean_max_waves;
    uint32_t g = (f.n_work + w - 1) / w;
    if (g > cus) g = cus;
    const uint32_t lds = f.coef_lds_bytes + w * f.lean_wave_lds_bytes;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(g), dim3(w * 64), lds, s,
                       p.segs, p.msgs, p.work, f.n_work, p.coef, p.ramp_table, p.src, p.dst,
                       p.src_arena_bytes, (int)p.L, (int)p.M, p.L_blk, p.M_blk, f.ring_bytes, (uint32_t*)f.d_counter);
    return hipGetLastError();
}

hipError_t launch_src_lean(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    prm.ramp_table = ctx->d_ramp_table;
    const uint32_t T = b->fast.T;
#define X(t, c, s_, sl, d, dl)                                                                                            \
    if (T == t && prm.channels == c && prm.sb == s_ && (prm.src_le != 0) == sl && prm.db == d && (prm.dst_le != 0) == dl) \
        return launch_lean_one<t, c, s_, sl, d, dl>(ctx, b, prm, s);
    OHGPU_BLOCK_KERNELS(X)
#undef X
    return hipErrorInvalidValue;
}
#endif   // host code: part 1 (or the single translation unit)

}  // namespace ohgpu
