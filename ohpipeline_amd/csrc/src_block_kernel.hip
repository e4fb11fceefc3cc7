// src_block_kernel.hip -- round 1's resample -> ramp -> pack kernel ("block kernel").  Since round 2 the batches run on
// src_lean_kernel.hip, since round 4 on src_mfma_wg_kernel.hip.  RETIRED as a selectable kernel in round 5: its nineteen
// instantiations and ohgpu_set_kernel_variant(ctx, 2) exist only in a legacy build (-DOHGPU_LEGACY_KERNELS: OHGPU_LEGACY=1 python
// ohpipeline_amd/build.py, or tools/build_variant.sh -- the same-box A/B reference).  What the shipped library keeps of it:
//   * the FALLBACK for filters whose phase sums break the lean kernel's rounding bias (sum|c| >= 2^29: this kernel rounds by
//     add-floor-convert and is exact up to the design's own bound of 2^30) -- five stereo instantiations
//     (OHGPU_BLOCK_FALLBACK_KERNELS), one translation unit.  ohgpu_src_design's 48 -> 44.1 kHz and 32 -> 48 kHz filters are such;
//   * what the planner takes from this file: which layouts the block kernels' main list holds (src_block_supported) and the rows /
//     ring geometry the lean kernel shares with it (src_block_geometry).
//
// Mapping (DESIGN.md "Resampler kernel"):
//   * A stream's output is cut into BLOCKS of L_blk frames that start where the polyphase phase is 0
//     (L_blk is a multiple of L), so every block walks the same phase sequence.
//   * One lane owns one CHANNEL of one block; the CH lanes of a block sit side by side in one wave.  All 64 lanes
//     of a wave are therefore at the SAME phase at the same instruction.
//   * The whole coefficient table ([L][T] doubles, 40 KB for 160 x 32) lives in LDS, loaded once per workgroup; a
//     workgroup is up to 12 waves that stay on their CU and loop over work units.  For an output, lane l reads taps
//     (l & 15) and (l & 15) + 16 of the phase's row with ONE ds_read2_b64 (conflict-free: 16 consecutive doubles,
//     the four 16-lane rows read the same addresses), and tap k is ONE v_fmac_f64_dpp row_newbcast:(k & 15), which
//     broadcasts lane k of every row as the coefficient -- full fp64 rate (tools/micro/dpp_fma.hip), no scalar
//     cache in the loop (its 16 KB thrashed on the table: 59 % misses, the waves spent half their time waiting).
//   * The lane keeps its T-sample sliding window in registers as exact integer-valued doubles.  The advance loop
//     is unrolled T times so that the circular window is indexed statically (slot = advance mod T).
//   * Input is staged through LDS by direct global->LDS loads (16 B per lane, two buffers, eight advances per
//     stage).  Rounded, ramped outputs go to an LDS ring as left-justified words; every four advances the wave
//     writes the 64-byte output lines that have become complete, four lanes per line, each HBM write a whole line.
//   * Accumulation is fp64 FMA on integer-valued operands with |sum| < 2^53: exact, hence bit-identical to the
//     integer model regardless of order.  No MFMA: this is a 1-D filter.
// Formats are template parameters (the per-advance unpack sits in the unrolled hot path); layouts without an
// instantiation run on the generic kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_block_common.h"

namespace ohgpu {

// ---- hand-issued LDS traffic of the per-output loop, with counted waits ----
// hipcc waits for LDS data with `s_waitcnt lgkmcnt(0)`, which also waits for whatever was issued last -- here the
// ring store of the previous output and the reads for the next one, i.e. a full LDS round trip per output with only
// three waves per SIMD to hide it.  The loop's LDS operations are therefore issued from inline asm, invisible to the
// compiler's wait insertion, in a fixed order per output (see the loop), and waited for by hand: a wave's LDS
// operations complete in issue order, so `lgkmcnt(N)` retires all but the N youngest.  N must not exceed the number
// of LDS operations issued after the awaited one on ANY path (a smaller N only waits longer).  Every statement has
// a "memory" clobber so that compiler-issued LDS traffic (message table, drain) keeps its place between them, and
// a wait names the awaited registers as "+v" so that no use can move above it.  No scalar memory operation may
// be in flight in the loop (they share the counter and complete out of order): tools/inspect_kernel.sh checks that.
template <int BYTES>
__device__ __forceinline__ void lds_issue_f64(double& dst, uint32_t addr)        // the double at addr + BYTES
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(BYTES) : "memory");
}
template <int DW>
__device__ __forceinline__ void lds_issue_2xu32(uint64_t& dst, uint32_t addr)     // the aligned words at addr + 4*DW, addr + 4*DW + 4
{
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "i"(DW), "i"(DW + 1) : "memory");
}
__device__ __forceinline__ void lds_issue_store_2xu32(uint32_t addr, uint32_t lo, uint32_t hi)
{
    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(addr), "v"(lo), "v"(hi) : "memory");
}
__device__ __forceinline__ void lds_issue_store_u32(uint32_t addr, uint32_t v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_issue_store_u16(uint32_t addr, uint32_t v)
{
    asm volatile("ds_write_b16 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
#define OHGPU_WAIT_INSN "s_waitcnt lgkmcnt(%"
__device__ __forceinline__ void lds_issue_store_3xu8(uint32_t addr, uint32_t v)   // bytes 0, 1, 2 of v at addr, addr + 1, addr + 2
{
    asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\tds_write_b8_d16_hi %0, %1 offset:2"
                 : : "v"(addr), "v"(v), "v"(v >> 8) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(double& x)
{
    asm volatile(OHGPU_WAIT_INSN "1)" : "+v"(x) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(double& x, uint64_t& y)
{
    asm volatile(OHGPU_WAIT_INSN "2)" : "+v"(x), "+v"(y) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(uint64_t& y)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(y) : "i"(N) : "memory");
}

// acc += (lane K of this lane's 16-lane row of cv) * x.  cv is only ever written by LDS loads (a VALU write would
// need two wait states before a DPP read; the s_nop in front of each output's first tap covers moves the
// compiler might insert anyway).
template <int K, bool FIRST>
__device__ __forceinline__ void fmac_bcast(double& acc, const double cv, const double x)
{
    if constexpr (FIRST)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cv), "v"(x), "i"(K));
    else
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cv), "v"(x), "i"(K));
}

// One subsample (SB packed bytes at any byte alignment) from LDS.  The eight frames of a stage are FB_SRC bytes apart,
// so frame s' sits at the same place in its dword as frame s' - P, P = 4 / gcd(4, FB_SRC): per stage P aligned base
// addresses and P shift amounts are computed, and the read itself carries an immediate offset (no address arithmetic
// per advance).  The LDS accepts unaligned addresses but
// serialises such an access lane by lane (SQ_LDS_UNALIGNED_STALL was half of the kernel's time), so the two ALIGNED
// words that hold the subsample are read (one ds_read2_b32) and shifted into place when the sample enters the window.
struct RawSubsample { uint64_t words; uint32_t shift; };
template <int SB, bool LE>
__device__ __forceinline__ int32_t unpack_subsample(const RawSubsample& r)          // -> S24 integer
{
    const uint32_t w = __builtin_amdgcn_alignbyte((uint32_t)(r.words >> 32), (uint32_t)r.words, r.shift);   // the 4 bytes that start at it
    if constexpr (SB == 3) {
        return LE ? ((int32_t)(w << 8)) >> 8 : ((int32_t)__builtin_bswap32(w)) >> 8;
    } else if constexpr (SB == 2) {
        return LE ? ((int32_t)(w << 16)) >> 8 : ((int32_t)(__builtin_bswap32(w) & 0xffff0000u)) >> 8;
    } else if constexpr (SB == 4) {
        return LE ? ((int32_t)w) >> 8 : ((int32_t)__builtin_bswap32(w)) >> 8;
    } else {
        return ((int32_t)(w << 24)) >> 8;
    }
}

template <int T, int CH>
struct BlockGeom {
    static constexpr int BPW = 64 / CH;                 // blocks per wave
    static constexpr int ROWS = BPW;
    static constexpr int MAX_WAVES = T <= 32 ? 12 : 8;  // waves per workgroup: 3 per SIMD (168 VGPRs each), 2 when the window is 64 deep
    static constexpr int MSG_SLOTS = 32;                // messages of a wave's output range kept in LDS
};

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE>
__global__ __launch_bounds__((BlockGeom<T, CH>::MAX_WAVES * 64))
void src_block_kernel(const SrcSeg* __restrict__ segs, const SegMsg* __restrict__ msgs, const SrcWork* __restrict__ work,
                      const uint32_t n_work, const double* __restrict__ coef, const uint16_t* __restrict__ ramp_table,
                      const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                      const uint64_t src_arena_bytes, const int L, const int M, const uint32_t L_blk, const uint32_t M_blk,
                      const uint32_t ring_bytes, uint32_t* __restrict__ unit_counter)
{
    static_assert(T % 16 == 0 && T >= 32 && T <= 64, "T / 16 coefficient registers per lane: register r holds taps 16 r + (lane & 15)");
    constexpr int NCR = T / 16;
    constexpr int BPW = BlockGeom<T, CH>::BPW, ROWS = BlockGeom<T, CH>::ROWS, MSG_SLOTS = BlockGeom<T, CH>::MSG_SLOTS;
    constexpr int FB_SRC = CH * SB, FB_DST = CH * DB;
    constexpr int IN_BLOCKS = ((8 * FB_SRC + 15 + 15) / 16) | 1;   // 16-byte pieces per staged row, odd (bank spread)
    constexpr int IN_STRIDE = IN_BLOCKS * 16;
    constexpr int IN_ITERS = (ROWS * IN_BLOCKS + 63) / 64;
    // a wave's private LDS region (after the shared coefficient table): input stages, message table, output ring
    constexpr uint32_t OFF_IN = 0, OFF_MSG = OFF_IN + 2 * ROWS * IN_STRIDE, OFF_MSGM = OFF_MSG + MSG_SLOTS * 16, OFF_RING = OFF_MSGM + MSG_SLOTS * 4;

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t table_bytes = (uint32_t)L * T * 8;
    const uint32_t coef_bytes = table_bytes + kRampLdsBytes;          // the shared part: coefficient table, then the ramp table
    const uint32_t row_stride = ring_bytes + 4;                       // output ring: rows start in different banks
    const uint32_t ring_area = (ROWS * row_stride + 15) & ~15u;
    const uint32_t wave_lds = OFF_RING + ring_area + 512;            // + 8 bytes per lane that absorb pair mode's idle stores

    // ---- coefficient table -> LDS, once per workgroup (the only workgroup barrier of the kernel).  The Q28
    // integers are scaled by 2^-28 on the way (exact): the accumulator is then in sample units, every partial sum
    // is a multiple of 2^-28 below 2^25 (exact in fp64), and rounding is floor(sum + 0.5) with the 0.5 preloaded.
    for (uint32_t i = tid; i < (uint32_t)L * T; i += blockDim.x)
        ((__attribute__((address_space(3))) double*)(lds_u8_t)smem)[i] = coef[i] * (1.0 / 268435456.0);
    // the 512 ramp multipliers too: a ramped output looks its multiplier up here, not in memory (a global load in the
    // per-output path waits, with vmcnt(0), for every staging load in flight)
    const __attribute__((address_space(3))) uint16_t* const ramp_lds =
        (const __attribute__((address_space(3))) uint16_t*)((lds_u8_t)smem + table_bytes);
    for (uint32_t i = tid; i < kRampLdsBytes / 4; i += blockDim.x)
        ((__attribute__((address_space(3))) uint32_t*)((lds_u8_t)smem + table_bytes))[i] = ((const uint32_t*)ramp_table)[i];
    __syncthreads();
    const uint32_t coef_lane = (uint32_t)(uintptr_t)((lds_u8_t)smem + (lane & 15) * 8);          // + phase * T * 8
    uint8_t* const wsmem = smem + coef_bytes + wave * wave_lds;       // this wave's region: staging, ring and
    const lds_u8_t lds = (lds_u8_t)wsmem;                            // write-back are private to the wave
    const int Mr = M % L;


    // Work units: every wave starts on its own unit, then claims further ones from a counter, so that the waves of
    // the whole grid finish together (waves sharing a SIMD run at very different speeds; a fixed share per wave
    // left the chip a quarter idle).  The claim is issued at the start of a unit and consumed at its end.
    const uint32_t first_claimed = gridDim.x * n_waves;
    uint32_t unit = blockIdx.x * n_waves + wave;
    while (unit < n_work) {
    uint32_t claim = 0;
    if (lane == 0) claim = atomicAdd(unit_counter, 1u);
    const SrcWork wk = work[unit];
    const SrcSeg seg = segs[wk.seg];
    const uint32_t n_blocks = wk.n_blocks;
    const uint32_t bw = lane / CH;                       // block within the wave's unit
    const uint32_t c = lane - bw * CH;                   // this lane's channel
    const uint32_t row = bw;
    const bool lane_valid = bw < BPW && row < n_blocks;
    const uint64_t blk = wk.first_block + row;
    const int64_t n_start = (int64_t)(blk * M_blk);      // absolute input frame at advance a = 0
    const int64_t row_g = seg.src_base + (n_start - T) * (int64_t)FB_SRC;   // byte offset of the frame at a_lin = 0
    // The stream start reads as zeros.  Blocks are at least T input frames long (planner), so only a block that
    // starts at input frame 0 reaches before the stream: its whole warm-up pass (advances a < 0) must be zeros.
    const bool first_block = n_start == 0;

    // Messages (ramp parameters) of the unit's output range.  Lane 0's block starts the range, in message
    // wk.msg_first; that and the next MSG_SLOTS - 1 messages go to an LDS table, compacted and made relative to the unit's first
    // output frame, so that the per-output path never issues a global load (it would have to wait for the staging
    // loads in flight).  A range with more messages than the table holds falls back to reading them from memory.
    const __attribute__((address_space(3))) u32x4* msg_tab = (const __attribute__((address_space(3))) u32x4*)(lds + OFF_MSG);
    const uint64_t wave_m0 = wk.first_block * (uint64_t)L_blk;
    const uint32_t tab_lo = wk.msg_first;                           // (the planner found it: no search here)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the previous unit's table has been read
        __builtin_amdgcn_wave_barrier();
        if (lane < (uint32_t)MSG_SLOTS) {
            uint32_t em = 0;
            u32x4 e = {0x7fffffffu, 0u, 0u, 0u};                   // past the segment: starts "never"
            if (tab_lo + lane < seg.msg_end) {
                const SegMsg m = msgs[tab_lo + lane];
                e.x = (uint32_t)(int32_t)(int64_t)(m.out0 - wave_m0);
                e.y = m.n;
                e.z = (uint32_t)m.ramp_start | ((uint32_t)m.ramp_end << 16);
                e.w = (uint32_t)m.flags | ((uint32_t)m.s_n1 << 8);
                em = m.m_n1;
            }
            ((__attribute__((address_space(3))) u32x4*)(lds + OFF_MSG))[lane] = e;
            ((__attribute__((address_space(3))) uint32_t*)(lds + OFF_MSGM))[lane] = em;    // the ramp's exact multiplier, read only when ramping
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // this lane's message cursor, relative to its own block: output j is frame j - msg_rel0 of the message
    const int32_t lane_off = (int32_t)(bw * L_blk);                // block's first output relative to the unit's
    uint32_t mi = 0;                                                // index into the table (tab_lo + mi in memory)
    int32_t msg_rel0 = 0;
    uint32_t msg_n = 0x7fffffffu, msg_ramp = 0, msg_flags = 0;     // msg_ramp = start | end << 16; msg_flags = flags | division shift << 8
    uint32_t msg_m = 0;                                             // x / (msg_n - 1) == umulhi(x, msg_m) >> shift
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
    if (lane_valid) {
        // first message that holds this block's output 0: the last table entry that starts at or before it.  Counted
        // over the whole table at once (32 pipelined broadcast reads, one wait): walking there entry by entry cost the
        // last block of a unit some twenty dependent LDS round trips (a quarter of the unit's set-up time).
        uint32_t cnt = 0;
#pragma unroll
        for (int idx = 0; idx < MSG_SLOTS; idx++) cnt += ((int32_t)msg_tab[idx].x <= lane_off) ? 1u : 0u;
        mi = cnt ? cnt - 1 : 0;
        load_msg(mi);
        while ((uint32_t)(0 - msg_rel0) >= msg_n) load_msg(++mi);   // (beyond the table: the range has more than MSG_SLOTS messages)
    }
    // first output index at which this lane needs the slow path: always while ramping, else at the message's end
    int32_t evt_j = (msg_flags & OHGPU_FLAG_RAMP) ? 0 : msg_rel0 + (int32_t)msg_n;

    // ---- input staging: stage q holds advances [8q - T, 8q + 8 - T) of every row, as raw packed bytes.
    // Lane `lane` moves pieces idx = it*64 + lane: piece `part` of row r = idx / IN_BLOCKS.  A stage is 8 frames =
    // a multiple of 16 bytes further on than the last, so a piece's (16-byte aligned) address just advances by that.
    static_assert((8 * FB_SRC) % 16 == 0, "a stage advances every piece by a whole number of 16-byte pieces");
    const int total = (int)M_blk + T;         // advances a = a_lin - T for a_lin in [0, total)
    const int n_stages = (total + 7) >> 3;
    int64_t piece_a[IN_ITERS];       // arena offset of this lane's piece of the NEXT stage to issue (16-byte aligned)
    bool piece_on[IN_ITERS];         // this lane has a piece to move
    bool unit_safe = true;           // every piece of every stage lies inside the arena: no per-piece checks
#pragma unroll
    for (int it = 0; it < IN_ITERS; it++) {
        const uint32_t idx = it * 64 + lane;
        const uint32_t r = idx / IN_BLOCKS;
        const int64_t g = seg.src_base + ((int64_t)((wk.first_block + r) * M_blk) - T) * (int64_t)FB_SRC;   // row's frame at a_lin = 0
        piece_a[it] = (g & ~(int64_t)15) + 16 * (int64_t)(idx - r * IN_BLOCKS);
        // a row whose first frame sits early in its piece does not reach into the last piece of the window
        // (8 frames + the 7 bytes the aligned two-word reads may touch beyond the last subsample)
        const uint32_t pieces_needed = (((uint32_t)g & 15u) + 8 * FB_SRC + 7 + 15) >> 4;
        piece_on[it] = r < n_blocks && r < (uint32_t)ROWS && (idx - r * IN_BLOCKS) < pieces_needed;
        const int64_t last = piece_a[it] + (int64_t)(n_stages - 1) * (8 * FB_SRC);
        if (piece_on[it] && (piece_a[it] < 0 || (uint64_t)last + 16 > src_arena_bytes)) unit_safe = false;
    }
    unit_safe = __all(unit_safe) != 0;
    auto issue_stage = [&](int q) __attribute__((always_inline)) {
        const uint32_t buf = OFF_IN + (uint32_t)(q & 1) * ROWS * IN_STRIDE;
#pragma unroll
        for (int it = 0; it < IN_ITERS; it++) {
            if (piece_on[it]) {
                const int64_t addr = piece_a[it];
                if (unit_safe || (addr >= 0 && (uint64_t)addr + 16 <= src_arena_bytes)) {
                    // LDS destination = wave-uniform base + lane*16
                    __builtin_amdgcn_global_load_lds((global_ptr_t)(src + addr),
                                                     (lds_ptr_t)(wsmem + buf + (uint32_t)(it * 64) * 16), 16, 0, 0);
                } else {
                    // piece straddles an end of the arena: copy only the bytes that exist
                    const lds_u8_t d = lds + buf + (uint32_t)(it * 64 + lane) * 16;
                    for (int b = 0; b < 16; b++) {
                        const int64_t a1 = addr + b;
                        d[b] = (a1 >= 0 && (uint64_t)a1 < src_arena_bytes) ? src[a1] : (uint8_t)0;
                    }
                }
            }
            piece_a[it] += 8 * FB_SRC;
        }
    };

    // ---- tail: the block's output is a byte stream of L_blk*FB_DST bytes that starts 64-byte aligned (planner).
    // A finished subsample is ramped, packed and stored at its place in the row's byte ring.  Whenever a 64-byte
    // line of the stream is complete, the wave writes that line of all its blocks: lane l of pass `it` copies
    // 16-byte piece (l & 3) of block (it*16 + l/4) from the ring to memory.  Four neighbouring lanes write one whole
    // line: every HBM write is a full, aligned line.
    // 24-bit stereo ("pair mode"): the two channel lanes of a block pack two frames = 12 bytes = three aligned words
    // between them (one lane swap, two byte permutes with per-lane selectors) and store them with ONE ds_write2_b32
    // every second output: lane A words 0-1, lane B words 1-2 (word 1 twice, same value).  Byte stores would cost
    // three LDS store instructions per output.  Other layouts store each subsample's bytes where they belong.
    constexpr bool PAIR = ring_pair_mode(CH, DB);
    // LDS stores per output: one aligned store, or three byte stores for 24-bit output that is not stereo
    constexpr int NS = (!PAIR && DB == 3) ? 3 : 1;
    static_assert(DB >= 2 && DB <= 4, "destination depths 16 / 24 / 32 bit");
    constexpr uint32_t B0 = DST_LE ? 0 : 2, B1 = 1, B2 = DST_LE ? 2 : 0;      // byte of the (right-justified) S24 value that is memory byte 0, 1, 2
    const uint32_t wave_lds_addr = (uint32_t)(uintptr_t)lds;
    const uint32_t ring_lane = wave_lds_addr + OFF_RING + row * row_stride + (PAIR ? c * 4 : c * DB);   // this lane's place in frame (pair) 0
    const uint32_t idle_lane = wave_lds_addr + OFF_RING + ring_area + lane * 8;                          // where pair mode's idle stores go
    // selectors for v_perm_b32 {got (bytes 4-7), own (bytes 0-3)}: words 0/1 from the even frame, words 1/2 from the odd one
    const uint32_t sel_lo = c == 0 ? (B0 | B1 << 8 | B2 << 16 | (4 + B0) << 24) : (B1 | B2 << 8 | (4 + B0) << 16 | (4 + B1) << 24);
    const uint32_t sel_hi = c == 0 ? ((4 + B1) | (4 + B2) << 8 | B0 << 16 | B1 << 24) : ((4 + B2) | B0 << 8 | B1 << 16 | B2 << 24);
    uint32_t w_even = 0;                                  // pair mode: the even output's word, waiting for its partner
    uint32_t st_addr = idle_lane, st_lo = 0, st_hi = 0;   // the store the NEXT output issues (this output's, or idle)
    const int64_t wave_dst = seg.dst_base + (int64_t)(wk.first_block * L_blk) * FB_DST;   // first block of this unit
    const uint32_t wave_rows = n_blocks < (uint32_t)BPW ? n_blocks : (uint32_t)BPW;
    uint32_t drained = 0;                                 // lines written so far (wave-uniform)
    uint32_t line_pos = 0;                                // ring position of line `drained`
    uint32_t ring_pos = 0;                                // ring position of frame j
    auto drain = [&](int j_now) __attribute__((always_inline)) {
        while (drained < (((uint32_t)(PAIR ? (j_now & ~1) : j_now) * FB_DST) >> 6)) {
#pragma unroll
            for (int it = 0; it < (BPW * 4 + 63) / 64; it++) {
                const uint32_t piece = it * 64 + lane;
                const uint32_t r = piece >> 2, part = piece & 3;
                if (r < wave_rows) {
                    uint32_t pos = line_pos + part * 16;
                    if (pos >= ring_bytes) pos -= ring_bytes;
                    const __attribute__((address_space(3))) uint32_t* q =
                        (const __attribute__((address_space(3))) uint32_t*)(lds + OFF_RING + r * row_stride + pos);
                    u32x4 v4;
                    v4.x = q[0]; v4.y = q[1]; v4.z = q[2]; v4.w = q[3];
                    u32x4* const o = (u32x4*)(dst + wave_dst + (int64_t)((uint64_t)r * L_blk) * FB_DST + drained * 64 + part * 16);
                    // written once, never read here: a non-temporal store keeps the output from pushing the input lines, which
                    // two or three stages re-read, out of the XCD's L2 (tools/exp_traffic.sh: 2.17 -> 1.80 GB per launch, -3.6 % time)
                    __builtin_nontemporal_store(v4, o);
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
    int p = 0;                                // phase of output j = t mod L (every block starts at phase 0)
    constexpr int PH = (FB_SRC % 4 == 0) ? 1 : ((FB_SRC % 2 == 0) ? 2 : 4);   // frames s' and s' + PH share their place in a dword
    static_assert((FB_SRC * PH) % 4 == 0 && (8 % PH) == 0 && FB_SRC * 7 / 4 + 1 < 256, "immediate dword offsets of ds_read2_b32");
    const uint32_t in_base = wave_lds_addr + OFF_IN + row * IN_STRIDE + ((uint32_t)row_g & 15u) + c * SB;   // frame 0 of buffer 0
    uint32_t in_shift[PH];                    // byte position of frame p's subsample in its dword (the same in every stage)
    uint32_t in_addr[PH];                     // aligned LDS address of frame p's dword in the current stage's buffer
#pragma unroll
    for (int ph = 0; ph < PH; ph++) { in_shift[ph] = (in_base + ph * FB_SRC) & 3u; in_addr[ph] = 0; }
    const bool any_first = __any(first_block) != 0;
    auto issue_store = [&]() __attribute__((always_inline)) {
        if constexpr (PAIR) lds_issue_store_2xu32(st_addr, st_lo, st_hi);
        else if constexpr (DB == 4) lds_issue_store_u32(st_addr, st_lo);
        else if constexpr (DB == 3) lds_issue_store_3xu8(st_addr, st_lo);
        else lds_issue_store_u16(st_addr, st_lo);
    };
    // coefficients of output j: register r = taps 16 r .. 16 r + 15, tap k in lane (k & 15) of every 16-lane row
    double cf[NCR];
    static_for([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = NCR - 1 - decltype(rc)::value;                  // issued in the order they are used: highest taps first
        lds_issue_f64<r * 128>(cf[r], coef_lane);
    }, std::make_integer_sequence<int, NCR>{});
    RawSubsample raw;

    issue_stage(0);
    issue_stage(1);                                       // (both buffers: the warm-up below runs two stages ahead)

    // LDS operations of the loop, in issue order.  Per advance: R (raw sample).  Per output: S (the previous output's
    // ring store; pair mode stores every second output and sends the other one to an idle slot so that the count is
    // fixed), B' and A' (the next output's coefficient registers, each reloaded as soon as its 16 taps are done).
    //     advance:  R  { W  S  16 taps(c[NCR-1])  C'[NCR-1]   W  16 taps(c[NCR-2])  C'[NCR-2]  ...
    //                      W  unpack  16 taps(c[0])  C'[0]   round, ramp, pack }*
    // The first wait awaits c[NCR-1]: issued last output, followed at least by that output's other NCR - 1 reloads
    //                                                                          -> lgkmcnt(NCR - 1)
    // Every later wait awaits c[r] (and the last one R as well): followed by the rest of last output's reloads, this
    // output's NS stores and the reloads already issued in this output                -> lgkmcnt(NCR - 1 + NS)
    // The newest sample is tap 0, used last, so its read has the first 16 taps to land.
    // ---- warm-up: the T advances before the block's first output only fill the window.  Nothing overlaps the staging
    // loads here, so their latency is what this pass costs: it runs TWO stages ahead (stages 0 and 1 were issued together,
    // stage q + 2 goes into the buffer stage q has just been read from), which halves the number of exposed waits; the
    // main loop then continues one stage ahead, from its second pass.  Four sample reads per LDS round trip.
    static_for([&](auto stage) __attribute__((always_inline)) {
        constexpr int q = decltype(stage)::value;
        if constexpr ((q & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stages q and q + 1 have landed
#pragma unroll
        for (int ph = 0; ph < PH; ph++)
            in_addr[ph] = ((in_base + ph * FB_SRC) & ~3u) + (uint32_t)(q & 1) * ROWS * IN_STRIDE;
        static_for([&](auto half) __attribute__((always_inline)) {
            constexpr int h = decltype(half)::value;
            RawSubsample r4[4];
            static_for([&](auto k4) __attribute__((always_inline)) {
                constexpr int sp = 4 * h + decltype(k4)::value, ph = sp % PH;
                lds_issue_2xu32<FB_SRC * (sp - ph) / 4>(r4[sp & 3].words, in_addr[ph]);
                r4[sp & 3].shift = in_shift[ph];
            }, std::make_integer_sequence<int, 4>{});
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r4[0].words), "+v"(r4[1].words), "+v"(r4[2].words), "+v"(r4[3].words) : : "memory");
#pragma unroll
            for (int k = 0; k < 4; k++) win[8 * q + 4 * h + k] = (double)unpack_subsample<SB, SRC_LE>(r4[k]);
        }, std::make_integer_sequence<int, 2>{});
        if constexpr (q + 2 <= T / 8) {
            if ((q + 2) * 8 < total) issue_stage(q + 2);               // into the buffer just read (its reads have been waited for)
        }
    }, std::make_integer_sequence<int, T / 8>{});
    if (any_first) {                                      // a stream's first block starts from silence
#pragma unroll
        for (int s = 0; s < T; s++) win[s] = first_block ? 0.0 : win[s];
    }
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
#pragma unroll
                    for (int ph = 0; ph < PH; ph++)     // (a stage shifts the row by whole 16-byte pieces: same misalignment)
                        in_addr[ph] = ((in_base + ph * FB_SRC) & ~3u) + (uint32_t)(q & 1) * ROWS * IN_STRIDE;
                }
                issue_store();                                          // the last output's bytes must be in the ring
                st_addr = idle_lane;
                drain(j);
            }
            // ---- advance: this channel's sample of frame (n_start + a) enters slot s ----
            {
                constexpr int sp = s & 7, ph = sp % PH;                 // frame sp of the stage = frame ph + PH * k
                lds_issue_2xu32<FB_SRC * (sp - ph) / 4>(raw.words, in_addr[ph]);
                raw.shift = in_shift[ph];
            }
            if (!(t < L * (a + 1))) {                                   // no output needs it yet (warm-up, or M > L)
                lds_wait<0>(raw.words);
                win[s] = (double)unpack_subsample<SB, SRC_LE>(raw);
            }
            // ---- emit the outputs whose newest input frame is this one: floor(t / L) == a ----
            while (t < L * (a + 1)) {
                double acc0 = 0.5, acc1 = 0.0;                                 // round half up: floor(sum + 0.5)
                p += Mr;                                                       // the next output's phase
                if (p >= L) p -= L;
                const uint32_t cp = coef_lane + (uint32_t)p * (T * 8);
                static_for([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = NCR - 1 - decltype(rc)::value;          // highest taps (oldest samples) first, the newest sample last
                    if constexpr (r == NCR - 1) {
                        lds_wait<NCR - 1>(cf[r]);
                        issue_store();
                    } else if constexpr (r == 0) {
                        lds_wait<NCR - 1 + NS>(cf[0], raw.words);
                        win[s] = (double)unpack_subsample<SB, SRC_LE>(raw);
                    } else {
                        lds_wait<NCR - 1 + NS>(cf[r]);
                    }
                    static_for([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = 14 - 2 * decltype(kc)::value;       // taps 16 r + 15 .. 16 r
                        fmac_bcast<k + 1, k == 14>(acc1, cf[r], win[(s - (16 * r + k + 1) + 2 * T) % T]);
                        fmac_bcast<k, false>(acc0, cf[r], win[(s - (16 * r + k) + 2 * T) % T]);
                    }, std::make_integer_sequence<int, 8>{});
                    lds_issue_f64<r * 128>(cf[r], cp);
                }, std::make_integer_sequence<int, NCR>{});
                int32_t y = (int32_t)floor(acc0 + acc1);
                y = y > 8388607 ? 8388607 : (y < -8388608 ? -8388608 : y);
                // pair mode packs straight from the S24 value; the other layouts (and the ramp) use the left-justified word (a11)
                uint32_t w = PAIR ? (uint32_t)y : (uint32_t)y << 8;
                if (__builtin_expect(__any(j >= evt_j) != 0, 0)) {              // message boundary or ramping somewhere in the wave (rare: out of line)
                    if (lane_valid && j >= evt_j) {
                        while ((uint32_t)(j - msg_rel0) >= msg_n) load_msg(++mi);   // next message of the segment
                        if (msg_flags & OHGPU_FLAG_RAMP) {
                            const uint32_t rs = msg_ramp & 0xffffu, re = msg_ramp >> 16;
                            const uint32_t mult = ramp_lds[ramp_index_magic(rs, (int32_t)(rs - re), (uint32_t)(j - msg_rel0), msg_n, msg_m, (msg_flags >> 8) & 31u)];
                            w = PAIR ? ramp_word(w << 8, mult, 3, CH, c) >> 8 : ramp_word(w, mult, 3, CH, c);
                            evt_j = j + 1;
                        } else {
                            evt_j = msg_rel0 + (int32_t)msg_n;
                        }
                    }
                }
                // pack; the next output (or the end of the unit) issues the store
                if constexpr (PAIR) {
                    if (j & 1) {
                        // lane A needs B's even word, lane B needs A's odd word
                        const uint32_t give = c == 0 ? w : w_even;
                        const uint32_t got = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                        st_lo = __builtin_amdgcn_perm(got, w_even, sel_lo);
                        st_hi = __builtin_amdgcn_perm(got, w, sel_hi);
                        st_addr = ring_lane + ring_pos;
                        ring_pos += 2 * FB_DST;
                        if (ring_pos == ring_bytes) ring_pos = 0;
                    } else {
                        w_even = w;
                        st_addr = idle_lane;
                    }
                } else {
                    st_lo = DST_LE ? (w >> (32 - 8 * DB)) : __builtin_bswap32(w);       // the DB bytes in memory order, first byte low
                    st_addr = ring_lane + ring_pos;
                    if constexpr (64 % CH != 0) st_addr = bw < (uint32_t)BPW ? st_addr : idle_lane;   // lanes beyond the last whole block
                    ring_pos += FB_DST;
                    if (ring_pos == ring_bytes) ring_pos = 0;
                }
                j++;
                t += M;
            }
        }, std::make_integer_sequence<int, T>{});
    }
    // the last output's store, then the lines it completes (LDS operations of a wave execute in order)
    issue_store();
    drain(j);
    unit = first_claimed + (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
    }   // units
    // The counters reset themselves: a wave reports in after its last claim, and the last wave of the grid to do so
    // zeroes both words for the next launch (no memset between launches; the batch owns the counters, one launch at a time).
    if (lane == 0) {
        const uint32_t waves_total = gridDim.x * n_waves;
        if (atomicAdd(unit_counter + 1, 1u) == waves_total - 1) {
            __hip_atomic_store(unit_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(unit_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- instantiations: (T, channels, source bytes, source LE, destination bytes, destination LE) ----
// The list is compiled in parts so that the build can run them side by side (ohpipeline_amd/build.py compiles this file
// once per part with -DOHGPU_BLOCK_PART=k): part 1 also holds the host code and only DECLARES the other parts' kernels;
// parts 2.. hold nothing but their kernels.  Without the macro (tools, tests) the file is one translation unit.
#ifdef OHGPU_LEGACY_KERNELS
#define OHGPU_KERNEL_ARGS const SrcSeg*, const SegMsg*, const SrcWork*, uint32_t, const double*, const uint16_t*, const uint8_t*, \
                          uint8_t*, uint64_t, int, int, uint32_t, uint32_t, uint32_t, uint32_t*
#define X_DEFINE(t, c, s_, sl, d, dl) template __global__ void src_block_kernel<t, c, s_, sl, d, dl>(OHGPU_KERNEL_ARGS);
#define X_DECLARE(t, c, s_, sl, d, dl) extern template __global__ void src_block_kernel<t, c, s_, sl, d, dl>(OHGPU_KERNEL_ARGS);
#if defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 2
OHGPU_BLOCK_KERNELS_2(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART) && OHGPU_BLOCK_PART == 3
OHGPU_BLOCK_KERNELS_3(X_DEFINE)
#elif defined(OHGPU_BLOCK_PART)
OHGPU_BLOCK_KERNELS_2(X_DECLARE)
OHGPU_BLOCK_KERNELS_3(X_DECLARE)
#endif

#endif   // OHGPU_LEGACY_KERNELS (the shipped library: one translation unit, the fallback list's kernels instantiated by their launches below)

#if !defined(OHGPU_BLOCK_PART) || OHGPU_BLOCK_PART == 1
// Launch shape: up to MAX_WAVES waves per workgroup (what the LDS left by the coefficient table allows), one
// workgroup per CU, waves loop over the work units; a small batch is spread as one-wave workgroups instead.
static void launch_shape(const ohgpu_ctx* ctx, const ohgpu_batch* b, uint32_t* grid, uint32_t* waves, uint32_t* lds)
{
    const SrcFastPlan& f = b->fast;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t w = (f.n_work + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > f.max_waves) w = f.max_waves;
    uint32_t g = (f.n_work + w - 1) / w;
    if (g > cus) g = cus;
    *grid = g; *waves = w; *lds = f.coef_lds_bytes + w * f.wave_lds_bytes;
}

template <int T, int CH, int SB, bool SRC_LE, int DB, bool DST_LE>
static hipError_t launch_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_block_kernel<T, CH, SB, SRC_LE, DB, DST_LE>;
    uint32_t grid, waves, lds;
    launch_shape(ctx, b, &grid, &waves, &lds);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(waves * 64), lds, s,
                       p.segs, p.msgs, p.work, b->fast.n_work, p.coef, p.ramp_table, p.src, p.dst,
                       p.src_arena_bytes, (int)p.L, (int)p.M, p.L_blk, p.M_blk, b->fast.ring_bytes,
                       (uint32_t*)b->fast.d_counter);
    return hipGetLastError();
}

bool src_block_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le)
{
#define X(t, c, s_, sl, d, dl) \
    if (T == t && ch == c && sb == s_ && (src_le != 0) == sl && db == d && (dst_le != 0) == dl) return true;
    OHGPU_BLOCK_KERNELS(X)
#undef X
    return false;
}

// Geometry the planner needs (must match the kernel's constexprs).  `out_per_drain` = the most outputs four
// consecutive advances can emit (ceil(4L/M)).  Returns false when the layout does not fit the CU's LDS.
bool src_block_geometry(uint32_t L, uint32_t T, uint32_t ch, uint32_t sb, uint32_t db, uint32_t out_per_drain,
                        uint32_t* rows, uint32_t* ring_bytes, uint32_t* coef_lds_bytes, uint32_t* wave_lds_bytes, uint32_t* max_waves)
{
    const uint32_t bpw = 64 / ch;
    const uint32_t fb_src = ch * sb, fb_dst = ch * db;
    const uint32_t in_blocks = ((8 * fb_src + 15 + 15) / 16) | 1;
    const uint32_t rb = ring_bytes_for(fb_dst, out_per_drain, ring_pair_mode(ch, db));
    *rows = bpw;
    *ring_bytes = rb;
    *coef_lds_bytes = L * T * 8 + kRampLdsBytes;
    *wave_lds_bytes = 2 * bpw * in_blocks * 16 + 32 * 16 + 32 * 4 + ((bpw * (rb + 4) + 15) & ~15u) + 512;
    const uint32_t budget = 160 * 1024;
    if (*coef_lds_bytes + *wave_lds_bytes > budget) return false;
    uint32_t w = (budget - *coef_lds_bytes) / *wave_lds_bytes;
    if (w > (T <= 32 ? 12u : 8u)) w = T <= 32 ? 12u : 8u;
    if (w < 4) return false;            // too few waves per CU to be worth it: the generic kernel takes the batch
    *max_waves = w;
    return true;
}

// the kernels THIS library has: the whole list in a legacy build, the fallback list otherwise
#ifdef OHGPU_LEGACY_KERNELS
#define OHGPU_BLOCK_BUILT(X) OHGPU_BLOCK_KERNELS(X)
#else
#define OHGPU_BLOCK_BUILT(X) OHGPU_BLOCK_FALLBACK_KERNELS(X)
#endif

bool src_block_built(uint32_t T, uint32_t ch, uint32_t sb, uint32_t src_le, uint32_t db, uint32_t dst_le)
{
#define X(t, c, s_, sl, d, dl) \
    if (T == t && ch == c && sb == s_ && (src_le != 0) == sl && db == d && (dst_le != 0) == dl) return true;
    OHGPU_BLOCK_BUILT(X)
#undef X
    return false;
}

hipError_t launch_src_block(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    if (!b->fast.enabled || b->fast.n_work == 0) return hipSuccess;
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    prm.ramp_table = ctx->d_ramp_table;
    const uint32_t T = b->fast.T;
#define X(t, c, s_, sl, d, dl)                                                                                            \
    if (T == t && prm.channels == c && prm.sb == s_ && (prm.src_le != 0) == sl && prm.db == d && (prm.dst_le != 0) == dl) \
        return launch_one<t, c, s_, sl, d, dl>(ctx, b, prm, s);
    OHGPU_BLOCK_BUILT(X)
#undef X
    return hipErrorInvalidValue;
}

#endif   // host code: part 1 (or the single translation unit)

}  // namespace ohgpu
