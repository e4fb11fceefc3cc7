// src_mfma_wg_kernel.hip -- the matrix-pipe resampler (src_mfma_kernel.hip has the arithmetic: int8 digit planes, twelve
// v_mfma_i32_16x16x64_i8 per tile of 16 outputs x 16 columns, 32-bit recombination) with the work cut for the MEMORY system:
// one unit per WORKGROUP, one output step per WAVE.
//
// Why.  With a unit per wave (src_mfma_kernel.hip) a wave reads 96 bytes of each of its 32 rows per step and writes 192 bytes of
// each per pair of steps.  Every byte is fetched once and every sector written whole (1.09 x the algorithmic traffic), and still
// the launch is bound by exactly that access pattern: with the arithmetic compiled out it takes 0.41 ms of the 0.42, because
// 65 000 row streams advancing a hundred bytes at a time leave no DRAM page open for its next visitor.  The same kernel moving
// the same bytes as one contiguous 6 KB run per wave and pair of steps takes 0.30 ms (`tools/exp_mfma.sh`, MF_DIAG_IO_CONTIG).
// So a unit has to arrive and leave in ONE piece.
//
// How.  A unit is up to 32 CONSECUTIVE blocks of a stream (src_plan.cpp: rows of one block): 28 KB of input and 30 KB of
// output, each contiguous in memory.  A block is `spb` = 10 steps of 16 output frames, and a step's coefficient image is the same
// for every block: wave w of the workgroup owns step w -- its A operands stay in 16 registers for the whole launch, no table is
// read in the loop -- and computes that step's four column tiles (32 rows x 2 channels) of every unit.  Per unit the workgroup
//   (A) copies the rows' input, 32 x 1152 bytes, from registers (loaded a unit ahead, lane-contiguous) into an LDS image,
//   (S) splits it into the digit planes (lane = one row's eight frames: src_mfma_common.h),
//   (C) runs the 40 tiles, packed results into an LDS image of the unit's output,
//   (D) writes that image out: 30 KB contiguous, whole lines, non-temporal,
// with a workgroup barrier between the phases; the input image and the output image share their LDS (72.5 KB a workgroup, two
// workgroups per CU), the next unit's input is in flight in registers during (C) and (D).
// Units whose 32-row input image does not lie wholly inside the source arena (kWorkEdge: the first of the first stream, the last of
// the last) stay with the unit-per-wave kernel, so nothing is checked here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_mfma_common.h"

namespace ohgpu {

constexpr uint32_t kWgSteps = 10;                   // steps per block (a 160-output block)
constexpr uint32_t kWgWaves = 8;                    // waves per workgroup: two per SIMD, and two workgroups per CU
constexpr uint32_t kWgThreads = kWgWaves * 64;
constexpr uint32_t kWgTilesPerWave = kWgSteps * 4 / kWgWaves;      // a unit is 10 steps x 4 column tiles; a wave takes 5 in step-major order
constexpr uint32_t kWgRows = 32;
constexpr uint32_t kWgChunks = 12;                  // chunks (16 frames) a row's outputs touch: frames -32 .. 159 of the row
constexpr uint32_t kWgRowIn = kWgChunks * 96;       // bytes of a row's input image
constexpr uint32_t kWgRowInPitch = kWgRowIn + 16;   // ... and its pitch in LDS (16 rows, 16 bytes each, then meet all 64 banks once)
constexpr uint32_t kWgRowOut = 160 * 6;             // bytes of a row's output
constexpr uint32_t kWgPlaneBytes = 3 * kWgChunks * 1024;   // [digit][chunk][column tile 4][column 16][16 frames]
constexpr uint32_t kWgStageBytes = kWgRows * kWgRowInPitch;  // the input image; the output image (32 x 960) lies over it
constexpr uint32_t kWgBiasBytes = kWgSteps * 768;   // [step][b0, b1, b2][output 16][4 copies] dwords: an output's value as the four-register C operand of its tile
constexpr uint32_t kWgLdsBytes = kWgPlaneBytes + kWgStageBytes + kWgBiasBytes;
static_assert(2 * kWgLdsBytes <= 160 * 1024, "two workgroups per CU");
constexpr uint32_t kWgOutPieces = kWgRows * (kWgRowOut / 16);    // 1920
static_assert(kWgTilesPerWave * kWgWaves == kWgSteps * 4 && kWgTilesPerWave == 5, "five tiles per wave: at most two steps");
static_assert(kWgThreads == 16 * kWgRows && kWgRowIn == 72 * 16, "sixteen lanes per row of the input image, 4.5 pieces each");

#ifndef MF_DIAG_BARRIER_MASK
#define MF_DIAG_BARRIER_MASK 0xf
#endif
template <int WHICH = 0>
__device__ __forceinline__ void wg_barrier()
{
    if constexpr (((MF_DIAG_BARRIER_MASK >> WHICH) & 1) == 0) {            // (timing experiments only: a barrier left out)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
    }
    // every LDS access of this wave has completed; nothing moves across (global loads in flight stay in flight)
#ifdef MF_DIAG_NO_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <bool SRC_LE, bool DST_LE>
__global__ __launch_bounds__(kWgThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))       // (two workgroups of eight waves per CU: 128 registers)
void src_mfma_wg_kernel(const LeanUnit* __restrict__ units, const uint32_t n_work,
                        const uint8_t* __restrict__ amat, const MfStep* __restrict__ steps,
                        const uint16_t* __restrict__ planes, const uint32_t plane_stride,
                        const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                        const uint32_t row_src_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const pl_lds = smem;
    uint8_t* const stage = smem + kWgPlaneBytes;
    uint8_t* const bias_lds = stage + kWgStageBytes;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // A unit is 40 tiles: (step 0..9) x (column tile 0..3).  Wave w takes tiles 5 w .. 5 w + 4 in step-major order: the last
    // n_a = 4 - w % 4 column tiles of step s_a = 5 (w / 4) + w % 4 ... no: its first n_a tiles are step s_a's, the rest step s_a + 1's.
    // Eight waves are two per SIMD whatever SIMD the first one lands on; ten (one per step) are 3, 3, 2, 2, and the phase between
    // two barriers lasts as long as its slowest wave.
    const uint32_t tile0 = kWgTilesPerWave * wave;
    const uint32_t step_a = tile0 >> 2, n_a = 4u - (tile0 & 3u);                         // tiles [0, n_a) of the wave are step_a's (column tiles 4 - n_a ..)
    const uint32_t step_b = step_a + 1u;                                               // (n_a = 4 and w = 7: step 10 is never used: 5 * 7 = 35 -> step 8, n_a = 1; fine)

    // the accumulators' initial values (MfStep::b0..b2) of the block's steps
    for (uint32_t i = tid; i < kWgSteps * 48u; i += kWgThreads) {
        const uint32_t t = i / 48u, r = i - 48u * t;
        const uint32_t v = steps[t].b0[r];                 // (b0, b1, b2 lie one after the other)
        ((u32x4*)bias_lds)[i] = u32x4{v, v, v, v};
    }
    // this wave's A operands, for good: its two steps'
    v4i a_a[4], a_b[4];
    const uint32_t step_b_c = step_b < kWgSteps ? step_b : step_a;
#pragma unroll
    for (int j = 0; j < 4; j++) a_a[j] = *(const v4i*)(amat + ((uint64_t)step_a * kMfStepImage + j * 1024u + lane * 16u));
#pragma unroll
    for (int j = 0; j < 4; j++) a_b[j] = *(const v4i*)(amat + ((uint64_t)step_b_c * kMfStepImage + j * 1024u + lane * 16u));
    const uint32_t kc_a = steps[step_a].kc, kc_b = steps[step_b_c].kc;

    // ---- lane roles ----
    // matrix operands: the SAMPLES are the A operand (lane = column n of the tile, K group g), the coefficients the B operand (lane =
    // output n of the step, K group g); the result's lane (g, n) then holds output frame n of the four columns 4 g .. 4 g + 3 =
    // rows 2 g and 2 g + 1 of the tile, both channels: a whole frame of each in one lane, nothing to exchange
    const uint32_t g = lane >> 4, n = lane & 15;
    const uint8_t* const b_lds = pl_lds + g * 1024u + n * 16u;                    // + kc * 1024 + digit * 12288 + tile * 256
    const uint8_t* const my_bias = bias_lds + 16u * n;                            // + step * 768: b0; b1 at + 256, b2 at + 512
    uint8_t* const out_lds = stage + 2u * g * kWgRowOut + 6u * n;                  // + 96 * step + tile * 8 rows (+ a row for the second frame)
    // the input image: sixteen lanes per row; lane `sub` of a row moves its pieces sub, sub + 16, .. sub + 48 and (sub < 8) sub + 64 of
    // the row's 72 -- an instruction reads 256 contiguous bytes of every row, and a lane's addresses differ by constants
    const uint32_t in_row = tid >> 4, in_sub = tid & 15u;
    const uint32_t in_src = in_row * row_src_bytes + 16u * in_sub;                 // + 256 k
    const uint32_t in_lds = in_row * kWgRowInPitch + 16u * in_sub;
    const uint32_t in_last = in_sub < 8u ? 1024u : 0u;                             // (the fifth round's spare lanes repeat their first piece)
    // the split: task q = 512 k + tid (k = 0, 1; q < 768) is half chunk q / 32 of row q % 32 (32 rows side by side: their plane
    // bytes are 32 contiguous bytes each)
    const uint32_t sp_row = tid & 31u, sp_hc0 = tid >> 5;                          // (second round: half chunk + 16)

    // pack: a frame's six bytes from its two 24-bit values, L then R, each most significant byte first (big endian) or last
    constexpr uint32_t kB0 = DST_LE ? 0 : 2, kB1 = 1, kB2 = DST_LE ? 2 : 0;       // byte of the 24-bit value that is memory byte 0, 1, 2
    constexpr uint32_t sel_lo = kB0 | kB1 << 8 | kB2 << 16 | (4 + kB0) << 24;     // {R, L} -> L's three bytes, R's first
    constexpr uint32_t sel_hi = (4 + kB1) | (4 + kB2) << 8 | 0x0c0c0000u;          // {R, L} -> R's other two

    auto issue_input = [&](const LeanUnit& w, u32x4 (&raw)[5]) __attribute__((always_inline)) {
        // (scalar base + 32-bit lane offset is the load's scalar-base form, but only if the offset is widened in THIS block: hoisted out
        // of the loop as a 64-bit pair it costs eight registers for the whole launch and a 64-bit add per load -- mf_here pins it)
        const uint8_t* const base = src + w.src_row0;
#ifdef MF_DIAG_NO_LOAD
        (void)base;
#pragma unroll
        for (int k = 0; k < 5; k++) raw[k] = u32x4{tid, in_src, (uint32_t)w.n_blocks, (uint32_t)k};
#else
        const uint32_t o = mf_here(in_src);
#pragma unroll
        for (int k = 0; k < 4; k++) raw[k] = *(const u32x4_u*)(base + o + 256 * k);
        raw[4] = *(const u32x4_u*)(base + mf_here(in_src + in_last));
#endif
    };
    auto stage_input = [&](const u32x4 (&raw)[5]) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_STAGE
        return;
#endif
#pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4*)(stage + in_lds + 256 * k) = raw[k];
        *(u32x4*)(stage + in_lds + in_last) = raw[4];
    };
    auto split_task = [&](uint32_t hc, bool first) __attribute__((always_inline)) {
        const uint8_t* const from = stage + sp_row * kWgRowInPitch + 48u * hc;
        u32x4 mine[3];
#pragma unroll
        for (int k = 0; k < 3; k++) mine[k] = *(const u32x4*)(from + 16 * k);
        uint32_t w[12] = {mine[0].x, mine[0].y, mine[0].z, mine[0].w, mine[1].x, mine[1].y, mine[1].z, mine[1].w, mine[2].x, mine[2].y, mine[2].z, mine[2].w};
        // the stream's block 0: the frames before it (chunks 0 and 1 of row 0) read as zeros
        const bool zero = first && sp_row == 0 && hc < 4u;
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = zero ? 0u : w[k];
        uint32_t pl[6][2];
        mf_split48(w, pl);
        uint8_t* const to = pl_lds + (hc >> 1) * 1024u + sp_row * 32u + (hc & 1u) * 8u;     // + digit * 12288 + channel * 16
#pragma unroll
        for (int c0 = 0; c0 < 6; c0++) {
            const int chn = c0 / 3, bpos = c0 % 3, digit = SRC_LE ? bpos : 2 - bpos;
            const uint32_t flip = digit < 2 ? 0x80808080u : 0u;
            *(u32x2*)(to + digit * 12288 + chn * 16) = u32x2{pl[c0][0] ^ flip, pl[c0][1] ^ flip};
        }
    };
    auto split_all = [&](bool first) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_SPLIT
        return;
#endif
        split_task(sp_hc0, first);
        if (tid < 256u) split_task(sp_hc0 + 16u, first);
    };

    // Units are dealt round robin: they cost the same (a ramped one a few instructions per tile more), so a workgroup's share
    // is even to within one unit in a hundred, and a unit index that is a launch constant plus a counter stays in scalar registers --
    // a claimed one would come back in a vector register, and with it every address of the unit.
    const uint32_t n_groups = gridDim.x;
    uint32_t u_cur = blockIdx.x;                           // the unit in the planes
    if (u_cur >= n_work) return;                            // (uniform; the launch keeps the grid within the units)
    uint32_t u_nxt = u_cur + n_groups;                     // the unit whose input is in flight
    LeanUnit wk = units[u_cur];
    u32x4 raw[5];
    issue_input(wk, raw);
    __syncthreads();                                        // (the bias table)
    stage_input(raw);
    wg_barrier();
    split_all((wk.flags & kWorkFirst) != 0);
    LeanUnit wk_nxt = units[u_nxt < n_work ? u_nxt : u_cur];
    issue_input(wk_nxt, raw);                               // (past the last unit: the current one again, never used)
    wg_barrier();

    while (true) {
        // ---- (C) this wave's step of the unit: four column tiles ----
        const uint32_t n_blocks = wk.n_blocks;
        const bool ramped = (wk.flags & kWorkRamped) != 0;
        const uint8_t* const mbase = (const uint8_t*)planes + (uint64_t)wk.plane * plane_stride;
#ifdef MF_DIAG_IO_ONLY
#pragma unroll
        for (int i = 0; i < 0; i++) {
#else
#pragma nounroll
        for (int i = 0; i < (int)kWgTilesPerWave; i++) {
#endif
            const bool first_step = (uint32_t)i < n_a;                 // (wave-uniform)
            const uint32_t step = first_step ? step_a : step_b;
            const uint32_t ct = (tile0 + (uint32_t)i) & 3u;
            const uint8_t* const bl = b_lds + (first_step ? kc_a : kc_b) * 1024u + ct * 256u;
            const uint8_t* const bi = my_bias + step * 768u;
            v4i bd[3];
#pragma unroll
            for (int d = 0; d < 3; d++) bd[d] = *(const v4i*)(bl + d * 12288);
            v4i s0 = *(const v4i*)bi, s1 = v4i{0, 0, 0, 0}, s2 = *(const v4i*)(bi + 256), s3 = v4i{0, 0, 0, 0},
                s4 = *(const v4i*)(bi + 512), s5 = v4i{0, 0, 0, 0};
            auto taps = [&](const v4i (&a)[4]) __attribute__((always_inline)) {
                s0 = MF_MFMA(bd[0], a[0], s0);
                s1 = MF_MFMA(bd[0], a[1], s1);
                s2 = MF_MFMA(bd[0], a[2], s2);
                s3 = MF_MFMA(bd[0], a[3], s3);
                s1 = MF_MFMA(bd[1], a[0], s1);
                s2 = MF_MFMA(bd[1], a[1], s2);
                s3 = MF_MFMA(bd[1], a[2], s3);
                s4 = MF_MFMA(bd[1], a[3], s4);
                s2 = MF_MFMA(bd[2], a[0], s2);
                s3 = MF_MFMA(bd[2], a[1], s3);
                s4 = MF_MFMA(bd[2], a[2], s4);
                s5 = MF_MFMA(bd[2], a[3], s5);
            };
            if (first_step) taps(a_a); else taps(a_b);
            int y[4];
#pragma unroll
            for (int v = 0; v < 4; v++) y[v] = mf_recombine(s0[v], s1[v], s2[v], s3[v], s4[v], s5[v]);
            if (ramped) {
                // RampApplicator::GetNextSample on the 24-bit value (Msg.cpp:840-895): top 16 bits * Q15 >> 15, low byte zero; the
                // multiplier of the lane's frame in each of its two rows comes from the unit's plane (0xffff: the frame's message has no ramp)
                const uint32_t row = ct * 8u + 2u * g;
                const uint32_t e0 = (row < n_blocks ? row * 160u : 0u) + 16u * step + n;
                const uint32_t e1 = (row + 1u < n_blocks ? (row + 1u) * 160u : 0u) + 16u * step + n;
                const uint32_t mu[2] = {*(const uint16_t*)(mbase + mf_here(2u * e0)), *(const uint16_t*)(mbase + mf_here(2u * e1))};
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int top = (int)((uint32_t)y[v] << 8) >> 16;  // bits 8..23, signed
                    const int r = (int)((uint32_t)((top * (int)mu[v >> 1]) >> 15) << 8);
                    y[v] = mu[v >> 1] != 0xffffu ? r : y[v];
                }
            }
            // pack: two permutes and three 16-bit stores per frame into the output image (a frame starts on an even byte)
            uint8_t* const os = out_lds + ct * (8u * kWgRowOut) + 96u * step;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t lo = mf_perm((uint32_t)y[2 * q + 1], (uint32_t)y[2 * q], sel_lo);
                const uint32_t hi = mf_perm((uint32_t)y[2 * q + 1], (uint32_t)y[2 * q], sel_hi);
                // (written out: left to the compiler the first two become one 4-byte store, misaligned for odd frames.  LDS operations
                // complete in order and every barrier here waits for lgkmcnt(0), so the compiler's own counts stay safe)
                const uint32_t at = (uint32_t)(uintptr_t)(lds_u8_t)(os + q * kWgRowOut);
                asm volatile("ds_write_b16 %0, %1\n\tds_write_b16_d16_hi %0, %1 offset:2\n\tds_write_b16 %0, %2 offset:4"
                             : : "v"(at), "v"(lo), "v"(hi) : "memory");
            }
        }
        wg_barrier<0>();                                    // the output image is whole; the planes are free

        // ---- (D) the unit leaves: 1920 lane-contiguous pieces.  vmcnt counts loads and stores together, in issue order: the next
        // unit's input -- requested a whole phase (C) ago -- is waited for HERE, in front of the stores, not behind them ----
        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]));
        {
            uint8_t* const unit_dst = dst + wk.dst_row0;
            const uint32_t out_bytes = n_blocks * kWgRowOut;
            constexpr int kRounds = (kWgOutPieces + kWgThreads - 1) / kWgThreads;      // 4: the last one three quarters full
            u32x4 op[kRounds];
#pragma unroll
            for (int k = 0; k < kRounds; k++) {
                uint32_t f = kWgThreads * k + tid;
                if (f >= kWgOutPieces) f = kWgOutPieces - 1;
                op[k] = *(const u32x4*)(stage + 16u * f);
            }
#pragma unroll
            for (int k = 0; k < kRounds; k++) {
                const uint32_t o = mf_here(16u * (kWgThreads * k + tid));
#if defined(MF_DIAG_NO_STORE)
                if (o < out_bytes && n_blocks > 1000000u) *(u32x4_u*)(unit_dst + o) = op[k];
#else
                if (o < out_bytes) __builtin_nontemporal_store(op[k], (u32x4_u*)(unit_dst + o));
#endif
            }
        }
        wg_barrier<1>();                                    // the output image has been read
        if (u_nxt >= n_work) break;                         // (uniform)

        // ---- (A) + (S) the next unit: registers -> input image -> planes; then its successor's input is requested ----
        stage_input(raw);
        wg_barrier<2>();
        split_all((wk_nxt.flags & kWorkFirst) != 0);
        wk = wk_nxt;
        u_cur = u_nxt;
        u_nxt += n_groups;
        wk_nxt = units[u_nxt < n_work ? u_nxt : u_cur];
        issue_input(wk_nxt, raw);
        wg_barrier<3>();
    }
}

bool src_mfma_wg_supported(uint32_t L_blk, uint32_t M_blk, uint32_t ch, uint32_t sb, uint32_t db)
{
    return ch == 2 && sb == 3 && db == 3 && L_blk == 16u * kWgSteps && (M_blk + 31u) / 16u + 1u == kWgChunks;
}

// does a unit's input image -- 32 rows of kWgRowIn bytes, whatever the number of blocks the unit holds -- lie inside the arena?
bool src_mfma_wg_unit_inside(int64_t src_row0, uint32_t row_src_bytes, uint64_t src_arena_bytes)
{
    return src_row0 >= 0 && (uint64_t)src_row0 + (uint64_t)(kWgRows - 1) * row_src_bytes + kWgRowIn <= src_arena_bytes;
}

template <bool SRC_LE, bool DST_LE>
static hipError_t launch_wg_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_mfma_wg_kernel<SRC_LE, DST_LE>;
    const SrcFastPlan& f = b->fast;
    if (f.n_wg == 0) return hipSuccess;
    if (!src_mfma_wg_supported(p.L_blk, p.M_blk, p.channels, p.sb, p.db)) return hipErrorInvalidValue;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t gsz = 2u * cus;                                 // two workgroups per CU (LDS), eight waves each
    if (gsz > f.n_wg) gsz = f.n_wg;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgLdsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(gsz), dim3(kWgThreads), kWgLdsBytes, s,
                       (const LeanUnit*)f.d_lean_units, f.n_wg, (const uint8_t*)f.d_mf_amat, (const MfStep*)f.d_mf_steps,
                       (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst, p.M_blk * 6u);
    return hipGetLastError();
}

hipError_t launch_src_mfma_wg(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    if (prm.src_le) return prm.dst_le ? launch_wg_one<true, true>(ctx, b, prm, s) : launch_wg_one<true, false>(ctx, b, prm, s);
    return prm.dst_le ? launch_wg_one<false, true>(ctx, b, prm, s) : launch_wg_one<false, false>(ctx, b, prm, s);
}

}  // namespace ohgpu
