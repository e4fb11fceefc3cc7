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
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_block_common.h"

#pragma clang diagnostic ignored "-Winline-asm"     // (m0 in clobber lists: the staging DMA sets it by hand)

namespace ohgpu {

// ---- hand-issued instructions of the per-output loop ----
// (A load's destination is a "+v" operand: the register is written when the data arrives, not when the statement is issued,
// so the variable must LIVE in that register from the issue to the wait -- a read-write operand ties the two, where a pure
// output may be given a scratch register that the compiler then copies from at once.  tests/test_block_kernel_asm.py checks
// the generated code for such copies.)
// The loop's LDS traffic is issued from inline asm so that the compiler's wait insertion does not see it (it would
// wait with lgkmcnt(0), i.e. for the prefetches just issued as well); the waits are counted by hand.  A wave's LDS
// operations complete in issue order, so lgkmcnt(N) retires all but the N youngest.  Every statement is volatile
// with a "memory" clobber: they keep their order among themselves and against compiler-issued memory operations.
template <int BYTES>
__device__ __forceinline__ void lean_issue_f64(double& dst, uint32_t addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "i"(BYTES) : "memory");
}
template <int DW>
__device__ __forceinline__ void lean_issue_2xu32(uint64_t& dst, uint32_t addr)
{
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "+v"(dst) : "v"(addr), "i"(DW), "i"(DW + 1) : "memory");
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

// sixteen taps of one coefficient register: taps 16 r + 15 .. 16 r in one dependent chain (two or four chains were measured
// slower: their extra moves and adds cost more than the latency they hide -- tools/micro/fma_latency.hip, exp10).
// `w` are the sixteen window slots in tap order 15 .. 0.  (cv is only ever written by an LDS load: no VALU-write -> DPP-read
// hazard.)
// (The accumulator is an ordinary in-place operand.  Round 2 briefly named v[0:1] in the text and listed it as clobbered, to
// keep LLVM from padding consecutive statements that share it with s_nop: a register that is only "clobbered" is free
// for the compiler's own temporaries BETWEEN the statements, and the 6- and 8-channel kernels' byte-store shift landed
// in it -- one LSB off in four outputs of ten, found by bench.py --config 4's check, now also a test.  The padding is
// avoided instead by giving every gap an instruction of the compiler's own, see the output's body.)
#define OHGPU_FM(k, x) "v_fmac_f64_dpp %[acc], %[cv], " x " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
#define OHGPU_FM4(k3, k2, k1, k0, x3, x2, x1, x0) OHGPU_FM(k3, x3) OHGPU_FM(k2, x2) OHGPU_FM(k1, x1) OHGPU_FM(k0, x0)
__device__ __forceinline__ void lean_taps16(double& acc, const double cv,
                                            const double w15, const double w14, const double w13, const double w12,
                                            const double w11, const double w10, const double w9, const double w8,
                                            const double w7, const double w6, const double w5, const double w4,
                                            const double w3, const double w2, const double w1, const double w0)
{
    asm volatile(
        OHGPU_FM4(15, 14, 13, 12, "%[w15]", "%[w14]", "%[w13]", "%[w12]") OHGPU_FM4(11, 10, 9, 8, "%[w11]", "%[w10]", "%[w9]", "%[w8]")
        OHGPU_FM4(7, 6, 5, 4, "%[w7]", "%[w6]", "%[w5]", "%[w4]") OHGPU_FM4(3, 2, 1, 0, "%[w3]", "%[w2]", "%[w1]", "%[w0]")
        : [acc] "+v"(acc)
        : [cv] "v"(cv), [w15] "v"(w15), [w14] "v"(w14), [w13] "v"(w13), [w12] "v"(w12), [w11] "v"(w11), [w10] "v"(w10), [w9] "v"(w9),
          [w8] "v"(w8), [w7] "v"(w7), [w6] "v"(w6), [w5] "v"(w5), [w4] "v"(w4), [w3] "v"(w3), [w2] "v"(w2), [w1] "v"(w1), [w0] "v"(w0));
}
// The last tap group of an output, with the unpack of the advance's new sample (lean_unpack below) in the same statement.
template <bool PL>
__device__ __forceinline__ void lean_taps16_unpack(double& acc, const double cv, const uint64_t words, const uint32_t sel, double& w0,
                                                   const double w15, const double w14, const double w13, const double w12,
                                                   const double w11, const double w10, const double w9, const double w8,
                                                   const double w7, const double w6, const double w5, const double w4,
                                                   const double w3, const double w2, const double w1)
{
    uint32_t w;
    asm volatile(
        ".if %[pl]\n\tv_lshlrev_b32 %[w], %[sel], %[lo]\n\t.else\n\tv_perm_b32 %[w], %[hi], %[lo], %[sel]\n\t.endif\n\tv_cvt_f64_i32 %[w0], %[w]\n\t"
        OHGPU_FM4(15, 14, 13, 12, "%[w15]", "%[w14]", "%[w13]", "%[w12]") OHGPU_FM4(11, 10, 9, 8, "%[w11]", "%[w10]", "%[w9]", "%[w8]")
        OHGPU_FM4(7, 6, 5, 4, "%[w7]", "%[w6]", "%[w5]", "%[w4]") OHGPU_FM4(3, 2, 1, 0, "%[w3]", "%[w2]", "%[w1]", "%[w0]")
        : [acc] "+v"(acc), [w0] "=&v"(w0), [w] "=&v"(w)
        : [cv] "v"(cv), [hi] "v"((uint32_t)(words >> 32)), [lo] "v"((uint32_t)words), [sel] "v"(sel), [pl] "i"(PL ? 1 : 0),
          [w15] "v"(w15), [w14] "v"(w14), [w13] "v"(w13), [w12] "v"(w12), [w11] "v"(w11), [w10] "v"(w10), [w9] "v"(w9),
          [w8] "v"(w8), [w7] "v"(w7), [w6] "v"(w6), [w5] "v"(w5), [w4] "v"(w4), [w3] "v"(w3), [w2] "v"(w2), [w1] "v"(w1));
}

// One subsample -> sample x 256 as an exact double: the two aligned words that hold it (LDS accepts unaligned reads
// but serialises them lane by lane), one byte permute with the lane's selector (alignment, byte order and the
// left-justification in one), one conversion.
// A planar source's frame is an aligned TInt32 at its bit depth: the unpack is a shift to the top (sel = 32 - depth).
#define OHGPU_UNPACK_ASM(W, HI, LO, SEL, PLANAR) \
    ".if " PLANAR "\n\tv_lshlrev_b32 " W ", " SEL ", " LO "\n\t.else\n\tv_perm_b32 " W ", " HI ", " LO ", " SEL "\n\t.endif\n\t"
template <bool PL>
__device__ __forceinline__ double lean_unpack(const uint64_t words, const uint32_t sel)
{
    uint32_t w;
    double d;
    asm volatile(OHGPU_UNPACK_ASM("%0", "%2", "%3", "%4", "%5") "v_cvt_f64_i32 %1, %0"
                 : "=&v"(w), "=v"(d) : "v"((uint32_t)(words >> 32)), "v"((uint32_t)words), "v"(sel), "i"(PL ? 1 : 0));
    return d;
}
// ... and without the conversion: sample x 256 as an integer (the half-band kernel's delay line keeps the odd frames that way)
__device__ __forceinline__ uint32_t lean_unpack_raw(const uint64_t words, const uint32_t sel)
{
    uint32_t w;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"((uint32_t)(words >> 32)), "v"((uint32_t)words), "v"(sel));
    return w;
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

// SB == 0 stands for the PLANAR source of OHGPU_FLAG_SRC_PLANAR32 (SB == 4 is a packed 32-bit source): every
// channel of a block is a staging row of its own, of 4-byte frames -- lane = (block, channel) reads ITS row -- so the rows
// staged are BPW * CH; everything downstream of the unpack is the packed layout's.
// HB: the half-band 2:1 decimator (host_design.cpp: L = 1, M = 2, a 63-tap prototype stored as T = 64 with every odd tap but the
// centre exactly zero).  T stays the filter's stored length -- the trip, the warm-up and the history are 64 input frames -- but
// the window holds only the T / 2 EVEN-parity frames the T / 2 non-zero outer taps meet (an output is due on every even
// frame), and the odd-parity frames wait, packed, in a delay line of T / 4 registers for the centre tap, T / 2 - 1 frames later.
#ifndef OHGPU_LEAN_MAX_WAVES_T32
#define OHGPU_LEAN_MAX_WAVES_T32 0                       // (experiments: waves per workgroup of the 32-slot-window kernels)
#endif
#ifndef OHGPU_LEAN_MAX_WAVES_WIDE
#define OHGPU_LEAN_MAX_WAVES_WIDE 16                     // ... of those with six channels or more: they fit 128 registers, four waves per SIMD
#endif
// Waves per workgroup.  A 32-slot window: three per SIMD for stereo (134 registers; and what the LDS left by the coefficient
// table allows with 16-frame stages, 11), FOUR for six channels and more -- those kernels have no pair exchange, fit 128
// registers and their 8-frame stages leave the LDS room (same-box A/B on config 4, tools/exp_wide16.sh: six channels 1.75 ->
// 1.67 ms, eight 2.25 -> 2.09 ms).  A half-band kernel carries the delay line on top and stays at three: the six-channel one
// does not fit 128 registers (it would spill), and the eight-channel one, which just does, measured 2.5 % SLOWER squeezed
// into them (tools/exp_hb8.sh: 2.98-2.99 against 2.91 ms).  Two per SIMD when the window alone is 128 registers (T = 64).
static constexpr int lean_max_waves(int tw, int ch, bool halfband)
{
    if (tw > 32) return 8;
    if (OHGPU_LEAN_MAX_WAVES_T32 > 0) return OHGPU_LEAN_MAX_WAVES_T32;
#ifdef OHGPU_DIAG_HB8_WAVES
    if (halfband && ch == 8) return OHGPU_DIAG_HB8_WAVES;      // (diagnostic: the eight-channel half-band kernel's occupancy)
#endif
    return (ch >= 6 && !halfband) ? OHGPU_LEAN_MAX_WAVES_WIDE : 12;
}
template <int T, int CH, int SB, int DB, bool HB = false>
struct LeanGeom {
    static constexpr bool PL = SB == 0;
    static constexpr int BPW = 64 / CH;
    static constexpr int ROWS = BPW;
    static constexpr int IN_ROWS = PL ? BPW * CH : BPW;   // staged rows
    static constexpr int TW = HB ? T / 2 : T;            // window slots = taps that meet the window
    static constexpr int MAX_WAVES = lean_max_waves(TW, CH, HB);
    static constexpr int FB_SRC = PL ? 4 : CH * SB, FB_DST = CH * DB;     // bytes per frame of a staged row
    static constexpr int SF = lean_stage_frames(CH);      // frames per stage
    static constexpr int IN_BLOCKS = PL ? lean_in_blocks_planar(CH) : lean_in_blocks(CH, SB);
    static constexpr int IN_STRIDE = IN_BLOCKS * 16;
    static constexpr int IN_ITERS = (IN_ROWS * IN_BLOCKS + 63) / 64;
    static constexpr bool DUMMY = (64 % CH) != 0;         // the lanes beyond the last whole block store into a ring of their own
};

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE, bool HB = false>
__global__ __launch_bounds__((LeanGeom<T, CH, SB, DB, HB>::MAX_WAVES * 64))
void src_lean_kernel(const LeanUnit* __restrict__ units, const uint32_t n_work,
                     const double* __restrict__ coef, const uint16_t* __restrict__ planes, const uint32_t plane_stride,
                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                     const uint64_t src_arena_bytes, const int L, const int M, const uint32_t L_blk1, const uint32_t M_blk1,
                     const uint32_t ring_bytes, const uint32_t src_shift, uint32_t* __restrict__ unit_counter, uint64_t* __restrict__ dbg)
{
    static_assert(T % 16 == 0 && T >= 32 && T <= 64, "T / 16 coefficient registers per lane");
    constexpr bool PL = LeanGeom<T, CH, SB, DB, HB>::PL;       // planar TInt32 source (src_shift = 32 - its bit depth)
    static_assert(!PL || (SRC_LE && !LeanGeom<T, CH, SB, DB, HB>::DUMMY), "planar instantiations: host-endian planes, channel counts that divide 64");
    static_assert(!HB || (T == 64 && !PL), "the half-band kernel: 63 taps stored as 64, packed sources");
    using G = LeanGeom<T, CH, SB, DB, HB>;
    constexpr int TW = G::TW;                               // window slots (HB: the even-parity frames only)
#ifdef OHGPU_DIAG_STAMP
    // diagnostic build: shader-clock stamps per phase, summed per wave, written to dbg[wave][8] at the end (never read by the kernel)
    uint64_t st_setup = 0, st_warm = 0, st_stage = 0, st_drain = 0, st_out = 0, st_units = 0, st_mark = 0;
    const uint64_t st_begin = __builtin_amdgcn_s_memtime();
    const uint64_t st_begin_real = __builtin_amdgcn_s_memrealtime();          // (100 MHz: the shader clock = ticks / real ticks * 100 MHz)
#define STAMP(acc) { const uint64_t n_ = __builtin_amdgcn_s_memtime(); acc += n_ - st_mark; st_mark = n_; }
#else
#define STAMP(acc)
#endif
    constexpr int NCR = TW / 16;
    constexpr int BPW = G::BPW, ROWS = G::ROWS, IN_ROWS = G::IN_ROWS;
    constexpr int FB_SRC = G::FB_SRC, FB_DST = G::FB_DST;
    constexpr int IN_BLOCKS = G::IN_BLOCKS, IN_STRIDE = G::IN_STRIDE, IN_ITERS = G::IN_ITERS, SF = G::SF;
    static_assert(T % (2 * SF) == 0, "a trip of T advances is a whole number of stage pairs: the buffer of a slot is static");
    // (every region a DMA instruction is aimed at starts on a 128-byte boundary -- more than the 16 bytes it needs, so that a
    // wave's staging buffers sit the same way against the LDS banks whatever its number in the workgroup)
    constexpr uint32_t BUF_BYTES = (IN_ROWS * IN_STRIDE + 127) & ~127;
    constexpr uint32_t OFF_IN = 0, OFF_RING = OFF_IN + 2 * BUF_BYTES;
    constexpr bool PAIR = ring_pair_mode(CH, DB);
    static_assert(DB >= 2 && DB <= 4, "destination depths 16 / 24 / 32 bit");
    static_assert((SF * FB_SRC) % 16 == 0, "a stage advances every piece by a whole number of 16-byte pieces");

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t table_bytes = (uint32_t)L * TW * 8;
    // (No pad between the rows: round 1 put 4 bytes there "so that the rows start in different banks", but the LDS counters are
    // identical to the last digit at pitches of 100, 108 and 124 bytes -- the ring stores' "conflicts" are the second pass of a
    // dual-dword operation, not addresses -- and without it the stereo kernel's per-wave LDS is 10 240 bytes: twelve waves per
    // CU instead of eleven.  Same-box A/B, alternating: a wash on most boxes (0.458-0.479 against 0.468-0.489 ms), 11 % faster
    // on a box that ran the whole kernel 40 % slow (0.605 against 0.681 ms): the twelfth wave is insurance, not speed.)
#ifndef OHGPU_LEAN_RING_PAD
#define OHGPU_LEAN_RING_PAD 0
#endif
    const uint32_t row_stride = ring_bytes + OHGPU_LEAN_RING_PAD;
    const uint32_t ring_area = (ROWS * row_stride + 15) & ~15u;
    const uint32_t dummy_bytes = G::DUMMY ? ring_bytes + 64u : 0u;    // (their store address moves with ring_pos like everyone's)
    const uint32_t wave_lds = (OFF_RING + ring_area + dummy_bytes + 127u) & ~127u;

    // ---- coefficient table -> LDS once per workgroup, scaled by 2^-36 (exact): the window holds samples x 256, so the
    // accumulator is in sample units with 28 fraction bits.
    // (HB: the table is the T / 2 outer taps -- the even ones -- and the centre tap, the one odd tap that is not zero, is a scalar)
    for (uint32_t i = tid; i < (uint32_t)L * TW; i += blockDim.x)
        ((__attribute__((address_space(3))) double*)(lds_u8_t)smem)[i] = coef[HB ? 2 * i : i] * (1.0 / 68719476736.0);
    __syncthreads();
    const double centre_tap = HB ? coef[T / 2 - 1] * (1.0 / 68719476736.0) : 0.0;      // (wave-uniform: a scalar load)
    const uint32_t coef_lane = (uint32_t)(uintptr_t)((lds_u8_t)smem + (lane & 15) * 8);
    const uint32_t coef_lane_L = coef_lane + (uint32_t)L * (TW * 8);    // + (phase - L) * TW * 8, modulo 2^32
    uint8_t* const wsmem = smem + table_bytes + wave * wave_lds;
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
    const uint32_t ring_lane = lane_block ? wave_lds_addr + OFF_RING + row * row_stride + (PAIR ? c * 4 : c * DB)
                                          : wave_lds_addr + OFF_RING + ring_area + (lane & 3) * 16;
    // Pair mode, no pad between the rows (twelve waves per CU): rows 96 bytes apart start 24 dwords apart, so rows r and r + 8
    // meet in the same banks and the ring stores of a half-wave collide two by two (SQ_LDS_BANK_CONFLICT 40 M -> 117 M per
    // launch).  Every second group of eight rows therefore keeps its ring ROTATED by half its length: what the others store at
    // ring position p it stores at (p + ring / 2) mod ring -- 12 dwords on, clear of its neighbour -- and the write-back reads it
    // there.  Only where half a ring is a whole number of both frame pairs and 16-byte pieces (ring a multiple of 96 bytes:
    // every stereo S24 layout the planner makes today); otherwise no rotation.
    const uint32_t ring_half = (uint32_t)__builtin_amdgcn_readfirstlane((int)((PAIR && ring_bytes % 96u == 0) ? ring_bytes / 2 : 0u));   // (an SGPR operand below)
    const uint32_t ring_rot = (lane_block && ((row >> 3) & 1u)) ? ring_half : 0u;
    const uint32_t ring_lane_lo = ring_lane + ring_rot;   // for ring positions below the half ...
    const uint32_t ring_lane_hi = ring_lane - ring_rot;   // ... and from the half on (position - half)
    const double bias = 16777216.5;                     // 2^24 + 0.5
    const uint32_t clamp_lo = 0x00800000u, clamp_hi = 0x017fffffu;   // 2^24 - 2^23 .. 2^24 + 2^23 - 1

    const uint32_t first_claimed = gridDim.x * n_waves;
    uint32_t unit = blockIdx.x * n_waves + wave;
    LeanUnit wk = units[unit < n_work ? unit : 0u];
    while (unit < n_work) {
    // The next unit is claimed LATE -- at this unit's last stage boundary, with a stage's worth of outputs still to go to cover
    // the atomic's latency: a unit claimed at the start of the one before it is a unit nobody else can take for that long, and
    // the launch ends when the slowest such pair does.
    uint32_t claim = 0;
    bool claimed = false;
#ifdef OHGPU_DIAG_STAMP
    st_mark = __builtin_amdgcn_s_memtime(); st_units++;
#endif
    const uint32_t n_blocks = wk.n_blocks;
    // A row is `kb` consecutive blocks of its stream (src_plan.cpp: long units first, one-block units for the end of the launch):
    // the window stays warm from block to block, so a row pays the T warm-up advances and their history once, not per block.
    const uint32_t kb = (wk.flags >> 8) & 0xffu;
    const uint32_t L_blk = L_blk1 * kb, M_blk = M_blk1 * kb;           // outputs / input frames per ROW of this unit
    const bool ramped = (wk.flags & kWorkRamped) != 0;                 // wave-uniform
    const bool checked = (wk.flags & kWorkChecked) != 0;               // some staging piece of the unit lies outside the arena
    const bool lane_valid = lane_block && row < n_blocks;
    const int64_t row_g = wk.src_row0 + (int64_t)(row * M_blk) * FB_SRC + (PL ? (int64_t)c * wk.src_plane_stride : 0);   // byte offset of the row's (planar: the lane's) frame at a_lin = 0
    const bool first_block = (wk.flags & kWorkFirst) != 0 && row == 0;        // the stream's block 0: nothing before it

    // ---- a ramped unit reads RampApplicator's multiplier of every output frame from its plane (src_plan.cpp): one uint16 per
    // frame, rows side by side, 0xffff = the frame's message carries no ramp.  Eight frames per load.
    const uint8_t* const mbase = (const uint8_t*)planes + (uint64_t)wk.plane * plane_stride;      // (wave-uniform)
    const uint32_t mrow = lane_valid ? row * L_blk * 2u : 0u;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;              // the next multipliers of this lane's row, 16 bits each, next one lowest in m0
    uint32_t moff = mrow;                                 // where this lane's next eight are, from mbase
    const uint32_t ramped_u = ramped ? 1u : 0u;

    // ---- input staging.  Stage q holds advances [SF q - T, SF q + SF - T) of every row as raw packed bytes: the aligned
    // 16-byte pieces that cover them, IN_BLOCKS per row, rows side by side.  Piece idx = it*64 + lane (row idx /
    // IN_BLOCKS) is moved by lane `lane` of DMA instruction `it`: its source is stage_base (wave-uniform, in SGPRs,
    // + SF frames per stage) + piece_off[it] (per lane, fixed for the unit).  A row whose first frame sits early in its
    // first piece does not need its last piece: that lane re-reads the row's first piece instead (same line, no traffic).
    const int total = (int)M_blk + T;         // advances a = a_lin - T for a_lin in [0, total)
    const int64_t g0 = wk.src_row0;            // row 0's frame at a_lin = 0
    const uint32_t a0 = (uint32_t)g0 & 15u;
    int64_t stage_off = g0 - (int64_t)a0;     // arena offset of row 0's first piece of the NEXT stage to issue (wave-uniform)
    uint32_t piece_off[IN_ITERS];
#pragma unroll
    for (int it = 0; it < IN_ITERS; it++) {
        const uint32_t idx = it * 64 + lane;
        uint32_t r = idx / IN_BLOCKS;                       // staged row: a block (planar: a block's channel)
        uint32_t part = idx - r * IN_BLOCKS;
        if ((PL ? r / CH : r) >= n_blocks || r >= (uint32_t)IN_ROWS) { r = 0; part = 0; }
        const uint32_t blk_r = PL ? r / CH : r;
        const uint32_t d_r = blk_r * M_blk * FB_SRC + (PL ? (r - blk_r * CH) * wk.src_plane_stride : 0u);
        const uint32_t al = (a0 + d_r) & 15u;
        const uint32_t pieces_needed = (al + SF * FB_SRC + 15) >> 4;     // bytes al .. al + SF FB_SRC - 1 of the row's first piece onwards
        if (part >= pieces_needed) part = 0;
        piece_off[it] = d_r - al + a0 + 16 * part;
    }
    auto issue_stage = [&](int q) __attribute__((always_inline)) {
#ifdef OHGPU_DIAG_NO_DMA
        if (q >= 0) { stage_off += SF * FB_SRC; return; }
#endif
        const uint32_t buf = OFF_IN + (uint32_t)(q & 1) * BUF_BYTES;
#pragma unroll
        for (int it = 0; it < IN_ITERS; it++) {
            constexpr int kTail = IN_ROWS * IN_BLOCKS - (IN_ITERS - 1) * 64;  // lanes of the last instruction that own a piece
            const bool last_partial = it == IN_ITERS - 1 && kTail < 64;
            const uint32_t m0v = wave_lds_addr + buf + (uint32_t)(it * 64) * 16;
            if (!checked) {
                const uint64_t sbase = (uint64_t)(uintptr_t)src + (uint64_t)stage_off;
#if defined(OHGPU_DIAG_DMA_POLICY_ID) && OHGPU_DIAG_DMA_POLICY_ID == 1     // (diagnostic builds: a cache policy on the staging loads)
#define LEAN_DMA_POLICY " nt"
#elif defined(OHGPU_DIAG_DMA_POLICY_ID) && OHGPU_DIAG_DMA_POLICY_ID == 2
#define LEAN_DMA_POLICY " sc1"
#elif defined(OHGPU_DIAG_DMA_POLICY_ID) && OHGPU_DIAG_DMA_POLICY_ID == 3
#define LEAN_DMA_POLICY " sc0 sc1"
#elif defined(OHGPU_DIAG_DMA_POLICY_ID) && OHGPU_DIAG_DMA_POLICY_ID == 4
#define LEAN_DMA_POLICY " sc0"
#else
#define LEAN_DMA_POLICY ""
#endif
                if (last_partial) {
                    // (the lanes beyond the last piece are masked off for this one instruction; EXEC is saved and put back, not
                    // assumed full: the statement then holds in whatever control flow a compiler leaves around it)
                    uint64_t keep;
                    asm volatile("s_mov_b64 %[keep], exec\n\ts_and_b64 exec, %[keep], %[mask]\n\ts_mov_b32 m0, %[m0v]\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %[off], %[base]" LEAN_DMA_POLICY "\n\ts_mov_b64 exec, %[keep]"
                                 : [keep] "=&s"(keep)
                                 : [off] "v"(piece_off[it]), [base] "s"(sbase), [m0v] "s"(m0v), [mask] "s"((1ull << (kTail & 63)) - 1ull)
                                 : "memory", "m0", "scc");
                } else {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" LEAN_DMA_POLICY
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
        stage_off += SF * FB_SRC;
    };

    // ---- write-back: the block's output is a byte stream of L_blk*FB_DST bytes.  A finished subsample is packed and stored at
    // its place in the row's byte ring; whenever a 64-byte line of the stream is complete the wave writes that line of all its
    // blocks, lane l of pass `it` copying 16-byte piece (l & 3) of block (it*16 + l/4): every HBM write is a whole line.  The
    // destination is a wave-uniform base (the unit's current line, pass `it`'s first row) plus the lane's fixed offset.
    const int64_t wave_dst = wk.dst_row0;
    const uint32_t wave_rows = n_blocks < (uint32_t)BPW ? n_blocks : (uint32_t)BPW;
    constexpr int DRAIN_ITERS = (BPW * 4 + 63) / 64;
    const uint32_t drain_off = (lane >> 2) * L_blk * FB_DST + (lane & 3) * 16;
    const uint32_t drain_lds = OFF_RING + (lane >> 2) * row_stride;
    uint32_t drained = 0;                                 // lines written so far (wave-uniform)
    uint32_t line_pos = 0;                                // ring position of line `drained`
    auto drain = [&](int frames_stored) __attribute__((always_inline)) {
#ifdef OHGPU_DIAG_NO_DRAIN
        if (frames_stored >= 0) return;
#endif
        while (drained < (((uint32_t)frames_stored * FB_DST) >> 6)) {
            uint32_t pos = line_pos + (lane & 3) * 16 + (((lane >> 5) & 1u) ? ring_half : 0u);   // (row = it * 16 + lane / 4: its bit 3 is the lane's bit 5)
            if (pos >= ring_bytes) pos -= ring_bytes;
            // every pass's pieces are read before any is stored: one LDS round trip per line, not one per pass
            u32x4 v4[DRAIN_ITERS];
#pragma unroll
            for (int it = 0; it < DRAIN_ITERS; it++) {
                const __attribute__((address_space(3))) uint32_t* q =
                    (const __attribute__((address_space(3))) uint32_t*)(lds + drain_lds + (uint32_t)(it * 16) * row_stride + pos);
                v4[it].x = q[0]; v4[it].y = q[1]; v4[it].z = q[2]; v4[it].w = q[3];
            }
            // (the reads are awaited HERE, by every lane: left to the compiler the wait sits inside the stores' lane mask, a wave
            // that skips a pass leaves its reads pending, and the compiler then opens the output loop that follows with lgkmcnt(0))
            // (LDS reads return in order: the last pass's registers name the wait for all of them)
            static_assert(DRAIN_ITERS <= 4, "one wait for every pass's reads");
            if constexpr (DRAIN_ITERS == 1) asm volatile("" : "+v"(v4[0]));
            else asm volatile("" : "+v"(v4[0]), "+v"(v4[DRAIN_ITERS - 1]));
#pragma unroll
            for (int it = 0; it < DRAIN_ITERS; it++) {
                if ((uint32_t)(it * 16) + (lane >> 2) < wave_rows) {
                    uint8_t* const line = dst + wave_dst + (uint64_t)drained * 64 + (uint64_t)(it * 16) * L_blk * FB_DST;    // wave-uniform
#if defined(OHGPU_DIAG_STORE_PLAIN)
                    *(u32x4*)(line + drain_off) = v4[it];
#elif !defined(OHGPU_DIAG_NO_STORE)
                    __builtin_nontemporal_store(v4[it], (u32x4*)(line + drain_off));
#else
                    if (v4[it].x == 0x12345678u && v4[it].y == 0x9abcdef0u) *(u32x4*)(line + drain_off) = v4[it];
#endif
                }
            }
            drained++;
            line_pos += 64;
            if (line_pos >= ring_bytes) line_pos -= ring_bytes;
        }
    };

    double win[TW];                           // (every slot is written by the warm-up before it is read)
    uint32_t dly[HB ? T / 4 : 1];             // HB: the last T / 4 odd-parity frames as sample x 256 integers (the centre tap's delay line)

    // (wave-uniform scalars; the loop's control flow is kept to single compares -- the scalar unit serves the CU's four SIMDs
    // one instruction per cycle between them, and round 2's first cut of this loop, with compound conditions, issued as many
    // scalar instructions as vector ones and ran no faster for having fewer of the latter)
    int j = 0;                                // outputs emitted so far
    int t = 0;                                // j * M
    int tl = 0;                               // L * (advances done): output j is due once t < tl ...
    const int t_end = (int)L_blk * M;         // ... and the block's last output has been emitted once t == t_end
    uint32_t pu = 0u - (uint32_t)L;           // phase of the NEXT coefficient reload, minus L, as an unsigned word (so that the wrap is a carry)
    uint32_t ring_pos = 0;                    // ring position of frame j (pair mode: of the pair)
    constexpr int PH = (FB_SRC % 4 == 0) ? 1 : ((FB_SRC % 2 == 0) ? 2 : 4);   // frames s' and s' + PH share their place in a dword
    static_assert((FB_SRC * PH) % 4 == 0 && (SF % PH) == 0 && FB_SRC * (SF - 1) / 4 + 1 < 256, "immediate dword offsets of ds_read2_b32");
    const uint32_t in_base = wave_lds_addr + OFF_IN + (PL ? lane : row) * IN_STRIDE + ((uint32_t)row_g & 15u) + (PL ? 0u : c * SB);   // frame 0 of buffer 0
    uint32_t in_sel[PH];                      // byte selector of frame ph's subsample (the same in every stage)
    uint32_t in_addr[2][PH];                  // aligned LDS address of frame ph's dword, by buffer
#pragma unroll
    for (int ph = 0; ph < PH; ph++) {
        in_sel[ph] = PL ? src_shift : lean_unpack_sel<SB, SRC_LE>((in_base + ph * FB_SRC) & 3u);   // (planar: the unpack is a shift)
        in_addr[0][ph] = (lane_block ? ((in_base + ph * FB_SRC) & ~3u) : (wave_lds_addr + OFF_IN));
        in_addr[1][ph] = in_addr[0][ph] + BUF_BYTES;
    }
    const bool any_first = __any(first_block) != 0;

    // The store of the last finished output (pair) is issued by the NEXT output, after its first wait, so that no wait ever
    // covers a store just issued.  It is issued by EVERY output: when nothing new is pending the same bytes go to the same
    // place once more (no branch, and the count of LDS operations per output stays fixed).
    uint32_t st_addr = PAIR ? ring_lane_lo : ring_lane, st_lo = 0, st_hi = 0;
    uint32_t y_even = 0, y_odd = 0;                           // frames whose bytes are in the ring or in the pending store
    auto issue_store = [&]() __attribute__((always_inline)) {
#ifdef OHGPU_DIAG_NO_RING
        return;
#endif
        if constexpr (PAIR) asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(st_addr), "v"(st_lo), "v"(st_hi) : "memory");
        else if constexpr (DB == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(st_addr), "v"(st_lo) : "memory");
        else if constexpr (DB == 3) asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8_d16_hi %0, %1 offset:2"
                                                 : : "v"(st_addr), "v"(st_lo), "v"(st_lo >> 8) : "memory");
        else asm volatile("ds_write_b16 %0, %1" : : "v"(st_addr), "v"(st_lo) : "memory");
    };

    // coefficients of output j: register r = taps 16 r .. 16 r + 15, tap k in lane (k & 15) of every 16-lane row
    double cf[NCR];
#pragma unroll
    for (int r = 0; r < NCR; r++) cf[r] = 0.0;
    static_for([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = NCR - 1 - decltype(rc)::value;
        lean_issue_f64<r * 128>(cf[r], coef_lane);
    }, std::make_integer_sequence<int, NCR>{});
    uint64_t rawp[2] = {0, 0};

    // ---- warm-up: the T advances before the block's first output only fill the window.  Stages 0 and 1 are issued together;
    // from then on stage q + 1 is issued when stage q begins (into the buffer stage q - 1 was read from), here and in the loop.
    issue_stage(0);
    issue_stage(1);
    STAMP(st_setup)
    static_for([&](auto stage) __attribute__((always_inline)) {
        constexpr int q = decltype(stage)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // stage q has landed
        if constexpr (q >= 1) {
            if ((q + 1) * SF < total) issue_stage(q + 1);
        }
        static_for([&](auto quad) __attribute__((always_inline)) {
            constexpr int h = decltype(quad)::value;                           // four sample reads per LDS round trip
            uint64_t r4[4] = {0, 0, 0, 0};
            static_for([&](auto k4) __attribute__((always_inline)) {
                constexpr int sp = 4 * h + decltype(k4)::value, ph = sp % PH;
                lean_issue_2xu32<FB_SRC * (sp - ph) / 4>(r4[sp & 3], in_addr[q & 1][ph]);
            }, std::make_integer_sequence<int, 4>{});
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]) : : "memory");
            if constexpr (HB) {               // frame a_lin = SF q + 4 h + k: even ones into the window, odd ones into the delay line
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    win[((SF * q + 4 * h + k) / 2) % TW] = lean_unpack<PL>(r4[k], in_sel[(4 * h + k) % PH]);
                    dly[((SF * q + 4 * h + k + 1) / 2) % (T / 4)] = lean_unpack_raw(r4[k + 1], in_sel[(4 * h + k + 1) % PH]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) win[SF * q + 4 * h + k] = lean_unpack<PL>(r4[k], in_sel[(4 * h + k) % PH]);
            }
        }, std::make_integer_sequence<int, SF / 4>{});
    }, std::make_integer_sequence<int, T / SF>{});
    if (any_first) {
#pragma unroll
        for (int s = 0; s < TW; s++) win[s] = first_block ? 0.0 : win[s];
        if constexpr (HB) {
#pragma unroll
            for (int s = 0; s < T / 4; s++) dly[s] = first_block ? 0u : dly[s];
        }
    }
    STAMP(st_warm)
    // The coefficient reads above were issued before the warm-up's waits: they have landed.  From here on, in issue order:
    //   per advance:  X (raw sample; of the NEXT advance where that lies in the same stage -- see the loop)
    //   per output:   W  S  16 taps(c[NCR-1])  C'[NCR-1]   W  16 taps(c[NCR-2])  C'[NCR-2] ...
    //                 W  unpack  16 taps(c[0])  C'[0]   round, clamp, (ramp,) pack
    // Every W awaits a reload C'[r] issued one output earlier (the last one the raw sample X as well).  After C'[r] come at
    // least the other NCR - 1 reloads (the rest of that output's, then this output's earlier ones) whatever else -- X, S --
    // was issued in between, and X, when there is one, is followed by S and NCR - 1 reloads: lgkmcnt(NCR - 1) is enough for
    // every W on every path.  Where more has been issued the wait also covers operations issued long before it, never the
    // youngest ones.  The block ends after L_blk outputs; the advances left in the trip then only move samples.
    for (int g = 1; g * T < total; g++) {
        static_for([&](auto slot) __attribute__((always_inline)) {
            constexpr int s = decltype(slot)::value;
            if constexpr ((s & 3) == 0) {
                STAMP(st_out)
                if constexpr ((s % SF) == 0) {
                    const int q = (g * T + s) / SF;
#ifndef OHGPU_DIAG_NO_STAGE_WAIT                                       // (diagnostic: wrong audio, the wait's share of the time)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage q has landed (this wave issued all of it)
#endif
                    if ((q + 1) * SF < total) issue_stage(q + 1);
                    else if (!claimed) {
                        if (lane == 0) claim = atomicAdd(unit_counter, 1u);
                        claimed = true;
                    }
                    STAMP(st_stage)
                }
                issue_store();
                drain(PAIR ? (j & ~1) : j);
                STAMP(st_drain)
            }
            // ---- advance: this channel's sample of the next input frame enters slot s ----
            tl += L;
            const int tle = tl < t_end ? tl : t_end;
            // The raw sample of advance s lands in rawp[s & 1].  It is read ONE ADVANCE AHEAD wherever the next advance lies in
            // the same stage (15 of 16, 7 of 8): an advance that emits nothing -- every second one at 96 -> 48 kHz -- then finds
            // its sample there instead of waiting out an LDS round trip.  Not across a stage's end (the next stage may not have
            // landed) and so never across the trip's back-edge (T is a whole number of stages): an in-flight register must not
            // be loop-carried.  Neither arm below touches the register the look-ahead is aimed at.
            constexpr int sp = s % SF, ph = sp % PH;
            constexpr int ws = HB ? s / 2 : s;                          // the window slot of this advance's frame (HB: even frames only)
            constexpr bool had_ahead = sp != 0, look_ahead = sp != SF - 1;
#ifndef OHGPU_DIAG_NO_X
            if constexpr (!had_ahead) lean_issue_2xu32<FB_SRC * (sp - ph) / 4>(rawp[s & 1], in_addr[(s / SF) & 1][ph]);
            if constexpr (look_ahead) {
                constexpr int sn = sp + 1, phn = sn % PH;
                lean_issue_2xu32<FB_SRC * (sn - phn) / 4>(rawp[(s + 1) & 1], in_addr[(s / SF) & 1][phn]);
            }
#endif
            if constexpr (HB && (s & 1) != 0) {
                // half-band, an odd-parity frame: no output is ever due on it and no outer tap ever meets it.  It waits, as
                // sample x 256, for the centre tap of the output T / 4 outputs on: wait for the sample and unpack, ONE statement
                // (the slot it takes is the one the previous output's centre tap has just read)
                asm volatile("s_waitcnt lgkmcnt(%[n])\n\tv_perm_b32 %[w], %[hi], %[lo], %[sel]"
                             : [w] "=&v"(dly[(s / 2) % (T / 4)])
                             : [hi] "v"((uint32_t)(rawp[s & 1] >> 32)), [lo] "v"((uint32_t)rawp[s & 1]), [n] "i"(look_ahead ? 1 : 0),
                               [sel] "v"(in_sel[sp % PH]) : "memory");
#ifdef OHGPU_DIAG_NO_EXPECT
            } else if (!(t < tle)) {
#else
            } else if (__builtin_expect(!(t < tle), 0)) {       // (unlikely: kept out of the line of the outputs, one taken branch less per output)
#endif
                // no output needs it yet (M > L), or the block is done: wait for the sample (everything but the look-ahead just
                // issued) and convert, in ONE statement
                uint32_t w;
                asm volatile("s_waitcnt lgkmcnt(%[n])\n\t"
                             ".if %[pl]\n\tv_lshlrev_b32 %[w], %[sel], %[lo]\n\t.else\n\tv_perm_b32 %[w], %[hi], %[lo], %[sel]\n\t.endif\n\t"
                             "v_cvt_f64_i32 %[d], %[w]"
                             : [w] "=&v"(w), [d] "=&v"(win[ws])
                             : [hi] "v"((uint32_t)(rawp[s & 1] >> 32)), [lo] "v"((uint32_t)rawp[s & 1]), [n] "i"(look_ahead ? 1 : 0),
                               [sel] "v"(in_sel[sp % PH]), [pl] "i"(PL ? 1 : 0) : "memory");
            } else {
                do {
                // ---- emit the outputs whose newest input frame is this one ----
                uint32_t cp;
                // the accumulator starts at the rounding bias; the next output's phase: pu += M mod L, and -L again on a carry; its
                // coefficient row
                double acc;
                if constexpr (HB) {
                    // ... and, half-band, the centre tap at once: the odd frame T / 2 - 1 back, out of the delay line (a fixed
                    // phase: the coefficient row never moves)
                    double mid;
                    asm volatile("v_mov_b64 %[acc], %[bias]\n\t"
                                 "v_cvt_f64_i32 %[mid], %[dl]\n\t"
                                 "v_lshl_add_u32 %[cp], %[pu], %[sh], %[cl]\n\t"
                                 "v_fmac_f64 %[acc], %[cen], %[mid]"
                                 : [acc] "=&v"(acc), [cp] "=&v"(cp), [mid] "=&v"(mid)
                                 : [bias] "s"(bias), [pu] "s"(pu), [sh] "i"(TW == 32 ? 8 : 9), [cl] "v"(coef_lane_L),
                                   [dl] "v"(dly[(s / 2) % (T / 4)]), [cen] "s"(centre_tap));
                } else {
                asm volatile("v_mov_b64 %[acc], %[bias]\n\t"
                             "s_add_u32 %[pu], %[pu], %[mr]\n\t"
                             "s_cselect_b32 vcc_lo, %[nl], 0\n\t"
                             "s_add_u32 %[pu], %[pu], vcc_lo\n\t"
                             "v_lshl_add_u32 %[cp], %[pu], %[sh], %[cl]"
                             : [acc] "=v"(acc), [pu] "+s"(pu), [cp] "=v"(cp)
                             : [bias] "s"(bias), [mr] "s"((uint32_t)Mr), [nl] "s"(0u - (uint32_t)L), [sh] "i"(TW == 32 ? 8 : 9), [cl] "v"(coef_lane_L)
                             : "vcc", "scc");
                }
                static_for([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = NCR - 1 - decltype(rc)::value;          // highest taps (oldest samples) first, the newest sample last
                    // (no register operands: the statements keep their order among themselves, and an operand written right in front
                    // of a statement that reads it gets that statement an s_nop)
                    if constexpr (r == 0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(rawp[s & 1]) : "i"(NCR - 1) : "memory");   // (the raw sample passes through its wait)
                    else asm volatile("s_waitcnt lgkmcnt(%0)" : : "i"(NCR - 1) : "memory");
                    // Consecutive statements that touch the accumulator are padded apart with an s_nop by the compiler (the gfx940 dst_sel forwarding hazard, assumed of every inline asm)
                    // unless an instruction of its own lies between them: the output is counted in front of its first taps and t
                    // moves on in front of its last ones, each pinned by scheduling barriers (the constant the pack needs happens to
                    // be rebuilt in front of the third).  The round-clamp-pack statement therefore sees j already incremented.
                    if constexpr (r == NCR - 1) {
                        issue_store();
                        __builtin_amdgcn_sched_barrier(0);
                        j++;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (r == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        t += M;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#define W_(k) win[(ws - (16 * r + (k)) + 2 * TW) % TW]
#ifndef OHGPU_DIAG_NO_TAPS
                    if constexpr (r == 0)
                        lean_taps16_unpack<PL>(acc, cf[0], rawp[s & 1], in_sel[(s % SF) % PH], win[ws], W_(15), W_(14), W_(13), W_(12), W_(11), W_(10), W_(9), W_(8),
                                           W_(7), W_(6), W_(5), W_(4), W_(3), W_(2), W_(1));
                    else
                        lean_taps16(acc, cf[r], W_(15), W_(14), W_(13), W_(12), W_(11), W_(10), W_(9), W_(8),
                                    W_(7), W_(6), W_(5), W_(4), W_(3), W_(2), W_(1), W_(0));
#else
                    if constexpr (r == 0) win[ws] = lean_unpack<PL>(rawp[s & 1], in_sel[(s % SF) % PH]);
#endif
#undef W_
                    lean_issue_f64<r * 128>(cf[r], cp);      // (no ablation switch here: the waits above count this reload)
                }, std::make_integer_sequence<int, NCR>{});
                // ---- round, clamp, (ramp,) pack: ONE hand-written statement.  u = trunc(2^24 + 0.5 + sum) clamped to
                // 2^24 + [-2^23, 2^23 - 1]; its low 24 bits are the S24 value.  Everything that survives the output is an in-place
                // operand, and the branches (ramped unit?  multipliers to fetch?  second frame of a pair?) are inside the statement:
                // a value written in one arm of a C++ branch comes back as register copies at the join, on every output of every unit.
                // A ramped unit applies RampApplicator::GetNextSample (Msg.cpp:840-895) to the 24-bit value -- top 16 bits * Q15 >> 15,
                // low byte zero -- with the multiplier in the low 16 bits of {m1, m0}, which then move on by one frame; eight frames'
                // multipliers per fetch, waited for in place (once per eight outputs of a ramped unit; the other waves cover it).
#define OHGPU_RAMP_ASM(Y, L0, L1)                                 \
                    "s_and_b32 vcc_lo, %[j], 3\n\t"                \
                    "s_cmp_lg_u32 vcc_lo, 1\n\t"                   \
                    "s_cbranch_scc1 " #L1 "f\n\t"                  \
                    "s_bitcmp1_b32 %[j], 2\n\t"                    \
                    "s_cbranch_scc1 " #L0 "f\n\t"                  \
                    "global_load_dword %[m0], %[moff], %[mb]\n\t"  \
                    "global_load_dword %[m1], %[moff], %[mb] offset:4\n\t"  \
                    "global_load_dword %[m2], %[moff], %[mb] offset:8\n\t"  \
                    "global_load_dword %[m3], %[moff], %[mb] offset:12\n\t" \
                    "v_add_u32 %[moff], 16, %[moff]\n\t"           \
                    "s_waitcnt vmcnt(0)\n\t"                       \
                    "s_branch " #L1 "f\n"                          \
                    #L0 ":\n\t"                                    \
                    "v_mov_b32 %[m0], %[m2]\n\t"                   \
                    "v_mov_b32 %[m1], %[m3]\n"                     \
                    #L1 ":\n\t"                                    \
                    "v_bfe_i32 %[t], " Y ", 8, 16\n\t"             \
                    "v_and_b32 %[mu], 0xffff, %[m0]\n\t"           \
                    "v_mul_i32_i24 %[t], %[t], %[mu]\n\t"          \
                    "v_alignbit_b32 %[m0], %[m1], %[m0], 16\n\t"   \
                    "v_ashrrev_i32 %[t], 15, %[t]\n\t"             \
                    "v_lshrrev_b32 %[m1], 16, %[m1]\n\t"           \
                    "v_cmp_ne_u32 vcc, 0xffff, %[mu]\n\t"          \
                    "v_lshlrev_b32 %[t], 8, %[t]\n\t"              \
                    "v_cndmask_b32 " Y ", " Y ", %[t], vcc\n\t"
                uint32_t t16, mu;
                if constexpr (PAIR) {
                    // (laid out so that an output of a plain unit takes exactly one taken branch)
                    uint32_t give, got;
                    asm volatile(
                        "s_bitcmp0_b32 %[j], 0\n\t"
                        "s_cbranch_scc1 40f\n\t"
                        // first frame of a pair: its value waits in ye
                        "v_cvt_u32_f64 %[ye], %[acc]\n\t"
                        "s_cmp_lg_u32 %[rf], 0\n\t"
                        "v_med3_u32 %[ye], %[ye], %[clo], %[chi]\n\t"
                        "s_cbranch_scc1 70f\n\t"
                        "s_branch 99f\n"
                        "70:\n\t"
                        OHGPU_RAMP_ASM("%[ye]", 71, 72)
                        "s_branch 99f\n"
                        "80:\n\t"
                        OHGPU_RAMP_ASM("%[yo]", 81, 82)
                        "s_branch 41f\n"
                        "40:\n\t"
                        // second frame: lane A (channel 0) needs B's first value, lane B needs A's second one; every lane offers what
                        // its partner wants and one quad-permuted move fetches it (two instructions between the offer's write and its
                        // DPP read); two byte permutes make the lane's two words of the pair's three
                        "v_cvt_u32_f64 %[yo], %[acc]\n\t"
                        "s_cmp_lg_u32 %[rf], 0\n\t"
                        "v_med3_u32 %[yo], %[yo], %[clo], %[chi]\n\t"
                        "s_cbranch_scc1 80b\n"
                        "41:\n\t"
                        "s_cmp_lt_u32 %[rp], %[half]\n\t"           // (the rotated rows' place: + half below the half, - half from it on)
                        "s_cselect_b64 vcc, -1, 0\n\t"
                        "v_cndmask_b32_e64 %[give], %[ye], %[yo], %[m55]\n\t"
                        "v_cndmask_b32 %[sta], %[rlh], %[rll], vcc\n\t"
                        "v_add_u32 %[sta], %[rp], %[sta]\n\t"
                        "s_add_u32 %[rp], %[rp], %[step]\n\t"
                        "v_mov_b32_dpp %[got], %[give] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "s_cmp_eq_u32 %[rp], %[ring]\n\t"
                        "v_perm_b32 %[lo], %[got], %[ye], %[sl]\n\t"
                        "s_cselect_b32 %[rp], 0, %[rp]\n\t"
                        "v_perm_b32 %[hi], %[got], %[yo], %[sh]\n"
                        "99:"
                        : [ye] "+v"(y_even), [yo] "+v"(y_odd), [sta] "+v"(st_addr), [lo] "+v"(st_lo), [hi] "+v"(st_hi), [rp] "+s"(ring_pos),
                          [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [m3] "+v"(m3), [moff] "+v"(moff),
                          [give] "=&v"(give), [got] "=&v"(got), [t] "=&v"(t16), [mu] "=&v"(mu)
                        : [acc] "v"(acc), [clo] "s"(clamp_lo), [chi] "v"(clamp_hi), [j] "s"(j), [rf] "s"(ramped_u), [mb] "s"(mbase),
                          [m55] "s"(0x5555555555555555ull), [rll] "v"(ring_lane_lo), [rlh] "v"(ring_lane_hi), [half] "s"(ring_half),
                          [ring] "s"(ring_bytes), [sl] "v"(sel_lo), [sh] "v"(sel_hi), [step] "i"(2 * FB_DST)
                        : "vcc", "scc", "memory");
                } else {
                    asm volatile(
                        "v_cvt_u32_f64 %[yo], %[acc]\n\t"
                        "s_cmp_lg_u32 %[rf], 0\n\t"
                        "v_med3_u32 %[yo], %[yo], %[clo], %[chi]\n\t"
                        "s_cbranch_scc0 41f\n\t"
                        OHGPU_RAMP_ASM("%[yo]", 81, 82)
                        "41:\n\t"
                        // the DB bytes in memory order, first byte lowest: little endian = the top DB bytes of the value left-justified
                        // to 32 bits (32-bit: value << 8; 24: the value; 16: value >> 8), big endian = bytes 2, 1, 0 of the value, then zero
                        "v_add_u32 %[sta], %[rp], %[rl]\n\t"
                        "s_add_u32 %[rp], %[rp], %[step]\n\t"
                        ".if %[le] == 2\n\t"
                        "v_lshlrev_b32 %[lo], 8, %[yo]\n\t"
                        ".elseif %[le] == 1\n\t"
                        "v_lshrrev_b32 %[lo], %[shr], %[yo]\n\t"
                        ".else\n\t"
                        "v_perm_b32 %[lo], %[yo], %[yo], %[bsw]\n\t"
                        ".endif\n\t"
                        "s_cmp_eq_u32 %[rp], %[ring]\n\t"
                        "s_cselect_b32 %[rp], 0, %[rp]"
                        : [yo] "+v"(y_odd), [sta] "+v"(st_addr), [lo] "+v"(st_lo), [rp] "+s"(ring_pos),
                          [m0] "+v"(m0), [m1] "+v"(m1), [m2] "+v"(m2), [m3] "+v"(m3), [moff] "+v"(moff), [t] "=&v"(t16), [mu] "=&v"(mu)
                        : [acc] "v"(acc), [clo] "s"(clamp_lo), [chi] "v"(clamp_hi), [j] "s"(j), [rf] "s"(ramped_u), [mb] "s"(mbase),
                          [rl] "v"(ring_lane), [ring] "s"(ring_bytes), [step] "i"(FB_DST), [le] "i"(DST_LE ? (DB == 4 ? 2 : 1) : 0),
                          [shr] "i"(DB == 4 ? 0 : 24 - 8 * DB), [bsw] "s"(0x0c000102u)
                        : "vcc", "scc", "memory");
                }
#undef OHGPU_RAMP_ASM
                } while (t < tle);
            }
        }, std::make_integer_sequence<int, T>{});
    }
    STAMP(st_out)
    if (!claimed && lane == 0) claim = atomicAdd(unit_counter, 1u);          // (a unit too short for a stage boundary of its own)
    // the next unit's descriptor is fetched while the last lines are written back
    unit = first_claimed + (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
    const LeanUnit next_wk = units[unit < n_work ? unit : n_work - 1u];
    issue_store();
    drain(j);
    wk = next_wk;
    STAMP(st_drain)
    }   // units
#ifdef OHGPU_DIAG_STAMP
    if (dbg != nullptr && lane == 0) {
        uint64_t* o = dbg + (size_t)(blockIdx.x * n_waves + wave) * 10;
        o[0] = st_setup; o[1] = st_warm; o[2] = st_stage; o[3] = st_drain; o[4] = st_out; o[5] = st_units;
        o[6] = st_begin; o[7] = __builtin_amdgcn_s_memtime();
        o[8] = st_begin_real; o[9] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // The counters reset themselves: a wave reports in after its last claim, and the last wave of the grid zeroes both.
    if (lane == 0) {
        const uint32_t waves_total = gridDim.x * n_waves;
        if (atomicAdd(unit_counter + 1, 1u) == waves_total - 1) {
            __hip_atomic_store(unit_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(unit_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define OHGPU_LEAN_ARGS const LeanUnit*, uint32_t, const double*, const uint16_t*, uint32_t, const uint8_t*, \
                        uint8_t*, uint64_t, int, int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint64_t*
#define X_DEFINE(t, c, s_, sl, d, dl) template __global__ void src_lean_kernel<t, c, s_, sl, d, dl>(OHGPU_LEAN_ARGS);
#define X_DECLARE(t, c, s_, sl, d, dl) extern template __global__ void src_lean_kernel<t, c, s_, sl, d, dl>(OHGPU_LEAN_ARGS);
#define X_DEFINE_HB(t, c, s_, sl, d, dl) template __global__ void src_lean_kernel<t, c, s_, sl, d, dl, true>(OHGPU_LEAN_ARGS);
#define X_DECLARE_HB(t, c, s_, sl, d, dl) extern template __global__ void src_lean_kernel<t, c, s_, sl, d, dl, true>(OHGPU_LEAN_ARGS);
#if defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 2
OHGPU_BLOCK_KERNELS_2(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 3
OHGPU_BLOCK_KERNELS_3(X_DEFINE)
OHGPU_LEAN_PLANAR_KERNELS(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 4
OHGPU_LEAN_HB_KERNELS(X_DEFINE_HB)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 5
OHGPU_LEAN_ONLY_KERNELS(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 6
OHGPU_LEAN_MORE_KERNELS(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART)
OHGPU_BLOCK_KERNELS_2(X_DECLARE)
OHGPU_BLOCK_KERNELS_3(X_DECLARE)
OHGPU_LEAN_PLANAR_KERNELS(X_DECLARE)
OHGPU_LEAN_HB_KERNELS(X_DECLARE_HB)
OHGPU_LEAN_ONLY_KERNELS(X_DECLARE)
OHGPU_LEAN_MORE_KERNELS(X_DECLARE)
#endif

#if !defined(OHGPU_BLOCK_PART) || OHGPU_BLOCK_PART == 1
// Geometry the planner needs (must match the kernel's constexprs).  Returns false when the layout does not fit the CU's LDS.
bool src_lean_geometry(uint32_t L, uint32_t T, bool halfband, uint32_t ch, uint32_t sb, uint32_t db, uint32_t out_per_drain,
                       uint32_t* rows, uint32_t* in_blocks, uint32_t* stage_frames, uint32_t* ring_bytes, uint32_t* coef_lds_bytes,
                       uint32_t* wave_lds_bytes, uint32_t* max_waves)
{   // (sb == 0: the planar source, LeanGeom)
    const bool planar = sb == 0;
    const uint32_t bpw = 64 / ch;
    const uint32_t fb_dst = ch * db;
    const uint32_t inb = planar ? (uint32_t)lean_in_blocks_planar((int)ch) : (uint32_t)lean_in_blocks((int)ch, (int)sb);
    const uint32_t in_rows = planar ? bpw * ch : bpw;
    *stage_frames = (uint32_t)lean_stage_frames((int)ch);
    const uint32_t rb = ring_bytes_for(fb_dst, out_per_drain, ring_pair_mode(ch, db));
    *rows = bpw;
    *in_blocks = inb;
    *ring_bytes = rb;
    const uint32_t tw = halfband ? T / 2 : T;              // (LeanGeom::TW: the taps the LDS table holds)
    *coef_lds_bytes = L * tw * 8;
    *wave_lds_bytes = (2 * ((in_rows * inb * 16 + 127u) & ~127u) + ((bpw * (rb + OHGPU_LEAN_RING_PAD) + 15) & ~15u) + ((64 % ch) ? rb + 64u : 0u) + 127u) & ~127u;
    const uint32_t budget = 160 * 1024;
    if (*coef_lds_bytes + *wave_lds_bytes > budget) return false;
    uint32_t w = (budget - *coef_lds_bytes) / *wave_lds_bytes;
    const uint32_t cap = (uint32_t)lean_max_waves((int)tw, (int)ch, halfband);
    if (w > cap) w = cap;
    if (w < 4) return false;
#ifdef OHGPU_DIAG
    if (const char* e = getenv("OHGPU_DIAG_MAX_WAVES")) { const uint32_t x = (uint32_t)atoi(e); if (x >= 4 && x < w) w = x; }   // (diagnostic builds: occupancy)
#endif
    *max_waves = w;
    return true;
}

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE, bool HB = false>
static hipError_t launch_lean_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_lean_kernel<T, CH, SB, SRC_LE, DB, DST_LE, HB>;
    const SrcFastPlan& f = b->fast;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t w = (f.n_lean + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > f.lean_max_waves) w = f.lean_max_waves;
    uint32_t g = (f.n_lean + w - 1) / w;
    if (g > cus) g = cus;
    const uint32_t lds = f.lean_coef_lds_bytes + w * f.lean_wave_lds_bytes;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t* dbg = nullptr;
#ifdef OHGPU_DIAG_STAMP
    const char* stamp_path = getenv("OHGPU_DIAG_STAMP_FILE");
    const size_t n_dbg = (size_t)g * w * 10;
    if (stamp_path && hipMalloc((void**)&dbg, n_dbg * 8) == hipSuccess) hipMemsetAsync(dbg, 0, n_dbg * 8, s);
#endif
    hipLaunchKernelGGL(kernel, dim3(g), dim3(w * 64), lds, s,
                       (const LeanUnit*)f.d_lean_units, f.n_lean, p.coef, (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst,
                       p.src_arena_bytes, (int)p.L, (int)p.M, p.L_blk, p.M_blk, f.ring_bytes, 32u - b->src_bits, (uint32_t*)f.d_counter, dbg);
#ifdef OHGPU_DIAG_STAMP
    if (dbg) {      // diagnostic build only: wait, sum up, write a text report, never on the product path
        hipStreamSynchronize(s);
        std::vector<uint64_t> h(n_dbg);
        hipMemcpy(h.data(), dbg, n_dbg * 8, hipMemcpyDeviceToHost);
        hipFree(dbg);
        if (FILE* fo = fopen(stamp_path, "w")) {
            double sum[6] = {0, 0, 0, 0, 0, 0}, t0 = 1e300, t1 = 0, life = 0, last_min = 1e300, life_real = 0;
            for (size_t i = 0; i < n_dbg; i += 10) {
                for (int k = 0; k < 6; k++) sum[k] += (double)h[i + k];
                if ((double)h[i + 6] < t0) t0 = (double)h[i + 6];
                if ((double)h[i + 7] > t1) t1 = (double)h[i + 7];
                if ((double)h[i + 7] < last_min) last_min = (double)h[i + 7];
                life += (double)(h[i + 7] - h[i + 6]);
                life_real += (double)(h[i + 9] - h[i + 8]);
            }
            const double nw = (double)(n_dbg / 10);
            fprintf(fo, "shader clock over the waves' lives: %.3f GHz (s_memtime / s_memrealtime at 100 MHz)\n", life / life_real * 0.1);
            fprintf(fo, "waves %.0f (grid %u x %u), units %.0f; kernel span %.0f ticks, first wave done at %.0f; mean wave life %.0f\n", nw, g, w, sum[5], t1 - t0, last_min - t0, life / nw);
            fprintf(fo, "mean ticks per wave: set-up %.0f warm-up %.0f stage wait+issue %.0f drain %.0f outputs %.0f | per unit: %.0f %.0f %.0f %.0f %.0f\n",
                    sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw,
                    sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5], sum[3] / sum[5], sum[4] / sum[5]);
            fclose(fo);
        }
    }
#endif
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
    if (b->src_planar) {                                   // (SB == 0 is the planar layout, whatever the samples' depth)
        prm.sb = 0;
        prm.src_le = 1;
        OHGPU_LEAN_PLANAR_KERNELS(X)
        return hipErrorInvalidValue;
    }
    if (b->fast.lean_halfband) {                           // the 2:1 decimator: 33 products per output instead of 64 (decided where the plan's LDS was sized: src_plan.cpp)
#define XHB(t, c, s_, sl, d, dl)                                                                                          \
    if (T == t && prm.channels == c && prm.sb == s_ && (prm.src_le != 0) == sl && prm.db == d && (prm.dst_le != 0) == dl) \
        return launch_lean_one<t, c, s_, sl, d, dl, true>(ctx, b, prm, s);
        OHGPU_LEAN_HB_KERNELS(XHB)
#undef XHB
        return hipErrorInvalidValue;                       // (the plan's table and waves are the half-band kernel's: never the plain one's launch)
    }
    OHGPU_BLOCK_KERNELS(X)
    OHGPU_LEAN_ONLY_KERNELS(X)
    OHGPU_LEAN_MORE_KERNELS(X)
#undef X
    return hipErrorInvalidValue;
}

// Layouts this kernel is instantiated for and round 1's is not (src_block_common.h: mono, packed 32-bit sources, wide
// little-endian outputs): the planner then makes a lean-only plan.
bool src_lean_only_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le)
{
#define X(t, c, s_, sl, d, dl) \
    if (T == t && ch == c && sb == s_ && (src_le != 0) == sl && db == d && (dst_le != 0) == dl) return true;
    OHGPU_LEAN_ONLY_KERNELS(X)
    OHGPU_LEAN_MORE_KERNELS(X)
#undef X
    return false;
}

// ... and the layouts its half-band form (T / 2 + 1 products per output, a table half as long, twelve waves) is instantiated for: the ONE
// place that says so, for the planner's geometry and for the dispatch above.
bool src_lean_halfband_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le)
{
#define X(t, c, s_, sl, d, dl) \
    if (T == t && ch == c && sb == s_ && (src_le != 0) == sl && db == d && (dst_le != 0) == dl) return true;
    OHGPU_LEAN_HB_KERNELS(X)
#undef X
    return false;
}
#endif   // host code: part 1 (or the single translation unit)

}  // namespace ohgpu
