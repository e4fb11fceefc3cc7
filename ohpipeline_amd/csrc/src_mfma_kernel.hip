// src_mfma_kernel.hip -- round 4's resample -> ramp -> pack kernel for 24-bit stereo: the taps on the MATRIX pipe.
//
// RETIRED from the shipped library in round 5: src_mfma_wg_kernel.hip (a unit per WORKGROUP) serves every layout this kernel did,
// edge units included.  The device kernel and its launcher are compiled only with -DOHGPU_LEGACY_KERNELS (OHGPU_LEGACY=1 python
// ohpipeline_amd/build.py, or tools/build_variant.sh: the A/B reference, ohgpu_set_kernel_variant(ctx, 5)); what is compiled
// always is the HOST half both matrix kernels share -- the coefficients' digit tables and step images (build_mfma_tables,
// build_mfma_images, build_mfma_halfband) -- and this text, which is where the arithmetic is explained.
//
// Why.  The lean kernel (src_lean_kernel.hip) computes an output as 32 dependent v_fmac_f64 per lane; three rounds of work on
// it ended at 0.37-0.40 of the HBM roofline with the per-output loop, not the memory system, as the bound.
// The sums are exact integers, y = sum_k c[p][k] x[n0 - k] with Q28 coefficients and 24-bit samples, so they can be written in
// base-256 digits, x = d0 + d1 2^8 + d2 2^16, c = e0 + e1 2^8 + e2 2^16 + e3 2^24, and then
//      y = sum_{s = 0..5} 2^(8 s) S_s,      S_s = sum over i + j = s of  sum_k d_i[k] e_j[k]
// where every S_s is an int8 dot product.  A TILE is 16 consecutive output frames of 16 columns (a column = one channel of
// one block row); the 16 windows of a tile lie inside one run of 64 input frames, so with the coefficients laid out as a
// banded 16 x 64 matrix per digit a tile is twelve v_mfma_i32_16x16x64_i8 into six accumulators -- 0.75 matrix-pipe cycles
// per output subsample against the 2 vector-pipe cycles of 32 fp64 FMAs -- and the per-output vector work shrinks from 44
// instructions (32 taps + 12) to about 15 (recombination 10, pack 3, digit planes 2.5), all of them 32-bit integer.
//
// How.
//   * Same units as the lean kernel (src_plan.cpp: LeanUnit, up to 32 rows of `kb` consecutive phase-aligned blocks, ramp
//     multiplier planes for ramped units): a wave owns a unit, 32 rows x 2 channels = 64 columns = FOUR column tiles.
//   * SAMPLE DIGITS.  The low two bytes of a sample are used as OFFSET digits, u - 128 = u ^ 0x80 read as int8 (no carry
//     chain; the top byte is the signed digit as it stands): x = [s8(b0^0x80) + 2^8 s8(b1^0x80) + 2^16 s8(b2)] + 128 * 257,
//     and the constant's share of an output, 32896 * sum_k c[p][k], is a per-phase constant the host folds -- together with
//     the rounding 2^27 -- into the accumulators' INITIAL VALUES (MfStep::b0..b2).  Raw zero bytes are the value 0, so frames
//     before a stream's start are zero bytes.
//   * DIGIT PLANES in LDS: per wave [digit 3][column tile 4][chunk slot 4][column 16][16 frames] bytes = 12 KB.  A chunk is 16
//     consecutive input frames; the four slots are a ring over chunk index (a tile's window is exactly four chunks: its K run
//     starts on a chunk boundary).  The B operand of lane (g = lane / 16, n = lane % 16) is ONE ds_read_b128: chunk kc + g,
//     column n -- conflict-free.
//   * EVERY GLOBAL ACCESS IS LANE-CONTIGUOUS.  The vector memory path handles a wave's access per lane unless neighbouring
//     lanes touch neighbouring bytes: the first version of this kernel -- every lane fetching its own 48 bytes, 12-byte stores
//     per lane, 16-byte coefficient rows per lane -- kept the texture addresser busy 98 % of a 0.70 ms launch (179 L1 accesses
//     per load instruction, TA_TA_BUSY = the launch).  So a chunk of the 32 rows (32 x 96 bytes) is fetched as 192 pieces of
//     16 bytes in row-major order -- six neighbouring lanes per row, three instructions -- into a 3 KB STAGE in LDS, from
//     which the split's lanes take their 48 bytes; the packed output of a step (32 rows x 96 bytes) goes through the same
//     stage the other way and leaves as 192 pieces; and the A operands are a lane-linear image per step (host table, 4 KB a
//     step, L2 resident), not rows gathered per lane.
//   * THE SPLIT: lane (row = lane / 2, half = lane % 2) transposes the 48 bytes of its row's 8 frames into six plane dword
//     pairs with a two-level v_perm_b32 network (2 permutes per plane dword, compile-time selectors), flips the offset digits'
//     top bits and writes three ds_write2_b64.
//   * ONE WAIT PER STEP.  vmcnt counts loads and stores together, in issue order: the next step's operands and the next chunk
//     are requested at the top of a step and waited for once, behind the step's arithmetic.
//   * RECOMBINATION in 32-bit integers: T0 = S0 + (S1 << 8), T1 = S2 + (S3 << 8), T2 = S4 + (S5 << 8) (each below 2^30),
//     U = T1 + (T0 >> 16), W = T2 + (U >> 16), y = (W << 4) | bits 12..15 of U = floor(acc / 2^28) exactly (the low 16 bits of
//     T0 and the low 12 of U cannot carry into bit 28), then one v_med3_i32.  Ten instructions per output.
//   * PACK: the two channel lanes of a row hold four frames each; they exchange two values (DPP quad_perm), three v_perm_b32
//     make the lane's 12 bytes of the row's 96.
// Bit-exact against the integer model (oracle/ohp_pipeline.c) like the kernels before it: same sum, same rounding.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_mfma_common.h"

namespace ohgpu {

#ifndef OHGPU_MFMA_WAVES
#define OHGPU_MFMA_WAVES 8                         // waves per workgroup = per CU (LDS: 18.2 KB each)
#endif
constexpr uint32_t kMfPlaneBytes = 3 * 4 * 4 * 256; // [digit][column tile][chunk slot][column][16]
constexpr uint32_t kMfStageHalf = 32 * 96 + 96;     // one step of every row, packed (+ 96: the two halves start in different banks)
constexpr uint32_t kMfStageBytes = 2 * kMfStageHalf; // two steps' output; the second half also takes a chunk of every row as it lies in memory
constexpr uint32_t kMfWaveLds = kMfPlaneBytes + kMfStageBytes;
constexpr uint32_t kMfBiasSteps = 16;               // steps per block the workgroup's table of accumulator biases holds (they repeat block by block)
constexpr uint32_t kMfBiasBytes = kMfBiasSteps * 192; // [step][b0 16, b1 16, b2 16] dwords

// Sixteen bytes from arena offset a, bytes outside the arena read as zero (they are frames before a stream's first or beyond the
// last frame any output needs).  Only units at an end of the arena come here (kWorkChecked).
__device__ __noinline__ u32x4 mf_load_piece_checked(const uint8_t* __restrict__ src, int64_t a, uint64_t arena_bytes)
{
    if (a >= 0 && (uint64_t)a + 16 <= arena_bytes) return *(const u32x4_u*)(src + a);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma nounroll
    for (int bb = 0; bb < 16; bb++) {
        const int64_t a1 = a + bb;
        if (a1 >= 0 && (uint64_t)a1 < arena_bytes) w[bb >> 2] |= (uint32_t)src[a1] << (8 * (bb & 3));
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}

#ifdef OHGPU_LEGACY_KERNELS                         // (the unit-per-wave kernel: retired from the shipped library in round 5, see the head of this file)
template <bool SRC_LE, bool DST_LE>
__global__ __launch_bounds__(OHGPU_MFMA_WAVES * 64)
void src_mfma_kernel(const LeanUnit* __restrict__ units, const uint32_t n_work,
                     const uint8_t* __restrict__ amat, const MfStep* __restrict__ steps,
                     const uint16_t* __restrict__ planes, const uint32_t plane_stride,
                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint64_t src_arena_bytes,
                     const uint32_t L_blk1, const uint32_t M_blk1, uint32_t* __restrict__ unit_counter)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t n_waves = blockDim.x >> 6;
    // the accumulators' initial values (MfStep::b0..b2): they depend on the outputs' phases only, so a block's steps hold them all
    const uint32_t spb = L_blk1 >> 4;                       // steps per block (<= kMfBiasSteps: the launch checks)
    for (uint32_t i = tid; i < spb * 48u; i += blockDim.x) {
        const uint32_t t = i / 48u, r = i - 48u * t;
        ((uint32_t*)smem)[i] = steps[t].b0[r];             // (b0, b1, b2 lie one after the other)
    }
    __syncthreads();
    uint8_t* const wl = smem + kMfBiasBytes + wave * kMfWaveLds;
    uint8_t* const stage0 = wl + kMfPlaneBytes;             // even steps' packed output
    uint8_t* const stage = stage0 + kMfStageHalf;          // odd steps' packed output; chunks on their way to the planes

    // ---- lane roles ----
    // matrix operands and results: g = K group / output quad, n = A row (output of the step) / column of the tile
    const uint32_t g = lane >> 4, n = lane & 15;
    const uint32_t rr = n >> 1, ch = n & 1;                // the column's row within its tile, its channel
    const uint8_t* const b_lds = wl + n * 16;              // + (digit * 4 + tile) * 1024 + slot * 256
    // the split: row and half chunk
    const uint32_t rs = lane >> 1, hs = lane & 1;
    uint8_t* const split_lds = wl + (rs >> 3) * 1024 + (rs & 7) * 32 + hs * 8;   // + digit * 4096 + slot * 256 + channel * 16
    const uint8_t* const split_stage = stage + rs * 96 + hs * 48;
    // transfers: piece f = 64 k + lane of the stage (k = 0..2) is piece f % 6 of row f / 6
    uint32_t tr_row[3], tr_piece[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t f = 64u * k + lane;
        tr_row[k] = f / 6u;
        tr_piece[k] = f - 6u * tr_row[k];
    }
    uint8_t* const tr_stage = stage + lane * 16;           // + 1024 k
    // a tile's packed result: 12 bytes at 24 g + 12 ch of the row's 96 -- an aligned 8-byte and a 4-byte store
    uint8_t* const out_stage = stage0 + rr * 96 + 24 * g + 12 * ch;   // + tile * 768 + (step & 1) * kMfStageHalf
    // the write-back, every second step: piece f = 64 k + lane (k = 0..5) is piece f % 12 of row f / 12's 192 bytes = three whole
    // 64-byte sectors when the stream's output starts on one (a block is a whole number of them).  64 k = 12 (5 k) + 4 k and
    // 4 (k + 3) = 12 + 4 k: instructions k and k + 3 differ by 16 rows, so three (row, piece) pairs serve the six.
    uint32_t wb_row[3], wb_piece16[3];
    const uint8_t* wb_stage[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t q = 4u * k + lane;
        wb_row[k] = 5u * k + q / 12u;                      // (instruction k + 3: + 16)
        const uint32_t pc = q % 12u;
        wb_piece16[k] = 16u * pc;
        wb_stage[k] = stage0 + (pc >= 6u ? kMfStageHalf + 16u * (pc - 6u) : 16u * pc) + wb_row[k] * 96u;   // (+ 1536)
    }

    // pack: {got, own} -> the lane's three dwords of two frames.  Channel 0 stores its own frames v = 0, 1 with its partner's,
    // channel 1 its own v = 2, 3 with its partner's: own = (L lane ? L : R), got = the other channel's value of the same frame.
    // Memory bytes of a frame: L then R, each most significant byte first (big endian) or last.
    constexpr uint32_t kB0 = DST_LE ? 0 : 2, kB1 = 1, kB2 = DST_LE ? 2 : 0;       // byte of the 24-bit value that is memory byte 0, 1, 2
    // dword 0 = L0[0..2] R0[0]; for the L lane own = L (bytes 0-3 of the pair), got = R (bytes 4-7); for the R lane the reverse
    const uint32_t own_l = ch == 0 ? 0u : 4u, own_r = ch == 0 ? 4u : 0u;          // where L's and R's value of a frame sit in {got, own}
    const uint32_t sel_d0 = (own_l + kB0) | (own_l + kB1) << 8 | (own_l + kB2) << 16 | (own_r + kB0) << 24;
    const uint32_t sel_d2 = (own_l + kB2) | (own_r + kB0) << 8 | (own_r + kB1) << 16 | (own_r + kB2) << 24;
    // dword 1 = R_a[1..2] L_b[0..1] from {L_b, R_a} = perm(hi = L of the second frame, lo = R of the first)
    constexpr uint32_t sel_d1 = kB1 | kB2 << 8 | (4 + kB0) << 16 | (4 + kB1) << 24;

    const uint32_t first_claimed = gridDim.x * n_waves;
    uint32_t unit = blockIdx.x * n_waves + wave;
    while (unit < n_work) {
        const LeanUnit wk = units[unit];
        const uint32_t n_blocks = wk.n_blocks;
        const uint32_t kb = (wk.flags >> 8) & 0xffu;
        const uint32_t L_blk = L_blk1 * kb, M_blk = M_blk1 * kb;        // outputs / input frames per row
        const bool ramped = (wk.flags & kWorkRamped) != 0;             // wave-uniform
        const bool first = (wk.flags & kWorkFirst) != 0;
        const uint32_t n_steps = L_blk >> 4;
        const uint32_t c_last = (M_blk + 31u) >> 4;                     // the last chunk any output of a row needs
        const bool zero_history = first && rs == 0;                    // the stream's block 0: frames before it read as zeros

        uint32_t claim = 0;
        // (the unit's body, once for units whose every piece lies inside the arena and once, with out-of-line checked loads, for the few
        // at its ends: a load that is a load on one path and a call on the other would be waited for at the join)
        auto run_unit = [&](auto checked_c) __attribute__((always_inline)) {
        constexpr bool CHECKED = decltype(checked_c)::value;
        // ---- the transfers' addresses: a wave-uniform base and a 32-bit lane offset (a unit spans far less than 4 GiB) ----
        const uint8_t* const unit_src = src + wk.src_row0;
        uint8_t* const unit_dst = dst + wk.dst_row0;
        const uint32_t row_src_bytes = M_blk * 6u, row_dst_bytes = L_blk * 6u;
        uint32_t tr_src[3];
#pragma unroll
        for (int k = 0; k < 3; k++)                                     // (rows past the unit's last re-read row 0 and are not stored)
            tr_src[k] = (tr_row[k] < n_blocks ? tr_row[k] : 0u) * row_src_bytes + 16u * tr_piece[k];      // + 96 c
        uint32_t wb_dst[3];
#pragma unroll
        for (int k = 0; k < 3; k++) wb_dst[k] = wb_row[k] * row_dst_bytes + wb_piece16[k];                // + 96 t (+ 16 rows)
        auto load_chunk = [&](uint32_t c, u32x4 (&raw)[3]) __attribute__((always_inline)) {
            if constexpr (!CHECKED) {
#pragma unroll
#ifdef MF_LOAD_NT
                for (int k = 0; k < 3; k++) raw[k] = __builtin_nontemporal_load((const u32x4_u*)(unit_src + (tr_src[k] + c * 96u)));
#else
                for (int k = 0; k < 3; k++) raw[k] = *(const u32x4_u*)(unit_src + (tr_src[k] + c * 96u));
#endif
            } else {
                // a unit at an end of the arena: a 16-byte piece that is not wholly inside is fetched byte by byte (out of line)
#pragma unroll
                for (int k = 0; k < 3; k++) raw[k] = mf_load_piece_checked(src, wk.src_row0 + (int64_t)(tr_src[k] + c * 96u), src_arena_bytes);
            }
        };
        // the chunk's 32 x 96 bytes go through the stage; lane (row, half) takes its 48 bytes = 8 frames x {L, R} x 3 bytes and
        // makes six planes of 8 bytes of them; c0 = 3 * channel + byte position in the sample
        auto split_chunk = [&](uint32_t c, const u32x4 (&raw)[3], bool head) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 3; k++) *(u32x4*)(tr_stage + 1024 * k) = raw[k];
            u32x4 mine[3];
#pragma unroll
            for (int k = 0; k < 3; k++) mine[k] = *(const u32x4*)(split_stage + 16 * k);
            uint32_t w[12] = {mine[0].x, mine[0].y, mine[0].z, mine[0].w, mine[1].x, mine[1].y, mine[1].z, mine[1].w, mine[2].x, mine[2].y, mine[2].z, mine[2].w};
            if (head && first && c < 2) {
#pragma unroll
                for (int k = 0; k < 12; k++) w[k] = zero_history ? 0u : w[k];
            }
            uint32_t pl[6][2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t* v = w + 6 * q;
                // level 1: two frames of two planes per permute ({hi, lo} = bytes 7..0)
                const uint32_t x01 = mf_perm(v[1], v[0], 0x07010600u), x23 = mf_perm(v[2], v[0], 0x05030402u), x45 = mf_perm(v[2], v[1], 0x07010600u);
                const uint32_t y01 = mf_perm(v[4], v[3], 0x07010600u), y23 = mf_perm(v[5], v[3], 0x05030402u), y45 = mf_perm(v[5], v[4], 0x07010600u);
                // level 2: four frames of one plane
                pl[0][q] = mf_perm(y01, x01, 0x05040100u); pl[1][q] = mf_perm(y01, x01, 0x07060302u);
                pl[2][q] = mf_perm(y23, x23, 0x05040100u); pl[3][q] = mf_perm(y23, x23, 0x07060302u);
                pl[4][q] = mf_perm(y45, x45, 0x05040100u); pl[5][q] = mf_perm(y45, x45, 0x07060302u);
            }
            const uint32_t slot = (c & 3u) * 256u;
#pragma unroll
            for (int c0 = 0; c0 < 6; c0++) {
                const int chn = c0 / 3, bpos = c0 % 3, digit = SRC_LE ? bpos : 2 - bpos;
                const uint32_t flip = digit < 2 ? 0x80808080u : 0u;
                *(u32x2*)(split_lds + digit * 4096 + slot + chn * 16) = u32x2{pl[c0][0] ^ flip, pl[c0][1] ^ flip};
            }
        };

        const uint8_t* const mbase = (const uint8_t*)planes + (uint64_t)wk.plane * plane_stride;

        struct StepOps { v4i a[4]; uint32_t kc; };
        auto load_ops = [&](uint32_t t, StepOps& o) __attribute__((always_inline)) {
            const MfStep* const st = steps + t;
#pragma unroll
            for (int j = 0; j < 4; j++) o.a[j] = *(const v4i*)(amat + ((uint64_t)t * kMfStepImage + j * 1024u + lane * 16u));
            o.kc = st->kc;
        };
        // one step: the four column tiles' 16 output frames -> the stage (half `t & 1`)
        auto do_step = [&](uint32_t t, const StepOps& o) __attribute__((always_inline)) {
            u32x2 mm[4];
            if (ramped) {
                // RampApplicator's multipliers of the lane's four frames in every column tile (0xffff: the frame's message has no ramp)
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {
                    const uint32_t row = (uint32_t)ct * 8u + rr;
                    mm[ct] = *(const u32x2*)(mbase + ((uint64_t)(row < n_blocks ? row * L_blk : 0u) + 16u * t + 4u * g) * 2u);
                }
            }
            const uint32_t slot_g = ((o.kc + g) & 3u) * 256u;
            const uint8_t* const bl = smem + (t % spb) * 192u + 16u * g;
            const v4i bias0 = *(const v4i*)bl, bias1 = *(const v4i*)(bl + 64), bias2 = *(const v4i*)(bl + 128);
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                v4i bd[3];
#pragma unroll
                for (int d = 0; d < 3; d++) bd[d] = *(const v4i*)(b_lds + (d * 4 + ct) * 1024 + slot_g);
                v4i s0 = bias0, s1 = v4i{0, 0, 0, 0}, s2 = bias1, s3 = v4i{0, 0, 0, 0}, s4 = bias2, s5 = v4i{0, 0, 0, 0};
                s0 = MF_MFMA(o.a[0], bd[0], s0);
                s1 = MF_MFMA(o.a[1], bd[0], s1);
                s2 = MF_MFMA(o.a[2], bd[0], s2);
                s3 = MF_MFMA(o.a[3], bd[0], s3);
                s1 = MF_MFMA(o.a[0], bd[1], s1);
                s2 = MF_MFMA(o.a[1], bd[1], s2);
                s3 = MF_MFMA(o.a[2], bd[1], s3);
                s4 = MF_MFMA(o.a[3], bd[1], s4);
                s2 = MF_MFMA(o.a[0], bd[2], s2);
                s3 = MF_MFMA(o.a[1], bd[2], s3);
                s4 = MF_MFMA(o.a[2], bd[2], s4);
                s5 = MF_MFMA(o.a[3], bd[2], s5);
                // ---- recombine, round (the bias carries 2^27), clamp: the lane's four frames 16 t + 4 g + v of column n ----
                int y[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int t0 = (int)(((uint32_t)s1[v] << 8) + (uint32_t)s0[v]);
                    const int u = (int)(((uint32_t)s3[v] << 8) + (uint32_t)s2[v]) + (t0 >> 16);
                    const int w = (int)(((uint32_t)s5[v] << 8) + (uint32_t)s4[v]) + (u >> 16);
                    int yy = (int)(((uint32_t)w << 4) | (((uint32_t)u >> 12) & 15u));
                    yy = yy < -8388608 ? -8388608 : (yy > 8388607 ? 8388607 : yy);
                    y[v] = yy;
                }
                if (ramped) {
                    // RampApplicator::GetNextSample on the 24-bit value (Msg.cpp:840-895): top 16 bits * Q15 >> 15, low byte zero
                    const uint32_t mu[4] = {mm[ct].x & 0xffffu, mm[ct].x >> 16, mm[ct].y & 0xffffu, mm[ct].y >> 16};
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int top = (int)((uint32_t)y[v] << 8) >> 16;  // bits 8..23, signed
                        const int r = (int)((uint32_t)((top * (int)mu[v]) >> 15) << 8);
                        y[v] = mu[v] != 0xffffu ? r : y[v];
                    }
                }
                // ---- pack: exchange two values with the other channel's lane, three permutes, 12 bytes into the stage ----
                const int give_a = ch ? y[0] : y[2], give_b = ch ? y[1] : y[3];
                const int own_a = ch ? y[2] : y[0], own_b = ch ? y[3] : y[1];
                const uint32_t got_a = (uint32_t)__builtin_amdgcn_mov_dpp(give_a, 0xb1, 0xf, 0xf, true);    // quad_perm:[1,0,3,2]
                const uint32_t got_b = (uint32_t)__builtin_amdgcn_mov_dpp(give_b, 0xb1, 0xf, 0xf, true);
                const uint32_t r_first = ch ? (uint32_t)own_a : got_a;      // R of the lane's first frame
                const uint32_t l_second = ch ? got_b : (uint32_t)own_b;     // L of its second
                const uint32_t o0 = mf_perm(got_a, (uint32_t)own_a, sel_d0);
                const uint32_t o1 = mf_perm(l_second, r_first, sel_d1);
                const uint32_t o2 = mf_perm(got_b, (uint32_t)own_b, sel_d2);
                // (the 12 bytes start on an 8-byte boundary for channel 0 and 4 bytes past one for channel 1)
                uint8_t* const os = out_stage + ct * 768 + (t & 1u) * kMfStageHalf;
                *(u32x2*)(os + (ch ? 4 : 0)) = ch ? u32x2{o1, o2} : u32x2{o0, o1};
                *(uint32_t*)(os + (ch ? 0 : 8)) = ch ? o0 : o2;
            }
        };

        // ---- the unit's head: the first window's four chunks, the chunk step 1 may add, the first pair's operands: one round trip ----
        StepOps e, f;                                       // the pair's even and odd step
        u32x4 raw_b[3];                                    // the chunk the pair's odd step adds to the planes, if it does
        bool add_b;
        uint32_t cn = 4;                                   // chunks [0, cn) are in the planes (a step's window is chunks kc .. kc + 3)
        {
            u32x4 r4[4][3];
#pragma unroll
            for (int c = 0; c < 4; c++) load_chunk((uint32_t)c, r4[c]);
            load_ops(0, e);
            load_ops(1, f);                                 // (a row is a whole number of blocks = an even number of steps)
            add_b = f.kc + 3u >= cn && cn <= c_last;
            load_chunk(cn <= c_last ? cn : c_last, raw_b);
#pragma unroll
            for (int c = 0; c < 4; c++) split_chunk((uint32_t)c, r4[c], true);
        }
        for (uint32_t t = 0; t < n_steps; t += 2) {
            // ---- PAIRS of steps.  vmcnt counts loads and stores together, in issue order, and the compiler waits for all of them at
            // the first use of any: so the next pair's operands and the (at most two) chunks its windows add are requested HERE, a whole
            // pair ahead -- a step's arithmetic alone is shorter than a trip to memory -- and waited for once, at the pair's end, in
            // front of its stores ----
            // (No load sits in a conditional: a value that is loaded on one path and copied on the other meets it in a register COPY
            // behind the join, and the copy is a use -- the wait would land here, in front of the pair's arithmetic.  A pair that needs
            // no chunk, and the unit's last pair, fetch the unit's last chunk again instead: a hit in the L2.)
            const bool more = t + 2 < n_steps;
            const uint32_t t2 = more ? t + 2u : t;
            StepOps e2, f2;
            load_ops(t2, e2);
            load_ops(t2 + 1u, f2);
            const uint32_t c_a2 = cn + (add_b ? 1u : 0u);              // the chunks in the planes once this pair is through
            const bool add_a2 = more && e2.kc + 3u >= c_a2 && c_a2 <= c_last;       // (wave-uniform)
            const uint32_t c_b2 = c_a2 + (add_a2 ? 1u : 0u);
            const bool add_b2 = more && f2.kc + 3u >= c_b2 && c_b2 <= c_last;
            u32x4 raw_a2[3], raw_b2[3];
#if defined(MF_DIAG_IO_CONTIG)
            // (timing only: the pair's two chunks as ONE contiguous 6 KB run near the unit's input span, clamped into the arena)
            {
                int64_t a0 = (wk.src_row0 > 0 ? wk.src_row0 : 0) + (int64_t)((t >> 1) * 5616u);
                if (a0 + 6144 > (int64_t)src_arena_bytes) a0 = (int64_t)src_arena_bytes - 6144;
                if (a0 < 0) a0 = 0;
                // (the bench's arena is far larger than 6 KB: this build is for it alone)
#pragma unroll
                for (int k = 0; k < 3; k++) raw_a2[k] = *(const u32x4_u*)(src + (a0 + 1024 * k + 16 * (int)lane));
#pragma unroll
                for (int k = 0; k < 3; k++) raw_b2[k] = *(const u32x4_u*)(src + (a0 + 3072 + 1024 * k + 16 * (int)lane));
            }
#elif !defined(MF_DIAG_NO_RAW)
            load_chunk(add_a2 ? c_a2 : c_last, raw_a2);
            load_chunk(add_b2 ? c_b2 : c_last, raw_b2);
#else
#pragma unroll
            for (int k = 0; k < 3; k++) raw_a2[k] = raw_b2[k] = raw_b[k];
#endif
#ifndef MF_DIAG_IO_ONLY
            do_step(t, e);
#endif
            if (add_b) {
                // the even step has read its window: the slot of the window's oldest chunk takes the chunk the odd step adds
#ifndef MF_DIAG_NO_RAW
                split_chunk(cn, raw_b, false);
#endif
                cn++;
            }
#ifndef MF_DIAG_IO_ONLY
            do_step(t + 1, f);
#endif
            // ---- everything requested at the top of the pair has to be here before a store goes out; then the two steps' output
            // leaves as 384 lane-contiguous pieces, whole sectors, past the L2 (non-temporal: the lines are complete and nobody
            // reads them again) ----
            asm volatile("" : "+v"(e2.a[0]), "+v"(e2.a[1]), "+v"(e2.a[2]), "+v"(e2.a[3]), "+v"(f2.a[0]), "+v"(f2.a[1]), "+v"(f2.a[2]), "+v"(f2.a[3]));
            asm volatile("" : "+v"(raw_a2[0]), "+v"(raw_a2[1]), "+v"(raw_a2[2]), "+v"(raw_b2[0]), "+v"(raw_b2[1]), "+v"(raw_b2[2]));
            if (!more && lane == 0) claim = atomicAdd(unit_counter, 1u);            // the next unit, claimed in this one's last pair (behind the wait: its result is a use)
            {
                u32x4 op[6];
#pragma unroll
                for (int k = 0; k < 6; k++) op[k] = *(const u32x4*)(wb_stage[k % 3] + (k / 3) * 1536);
#pragma unroll
                for (int k = 0; k < 6; k++) {
#ifdef MF_DIAG_IO_CONTIG
                    // (timing only: the pair's output as ONE contiguous 6 KB run of the unit's own output span; a pair that would
                    // leave a partly filled unit's span is not stored)
                    const bool valid = ((t >> 1) + 1u) * 6144u <= n_blocks * row_dst_bytes;
                    u32x4_u* const at = (u32x4_u*)(unit_dst + ((t >> 1) * 6144u + 1024u * k + 16u * lane));
#else
                    const bool valid = wb_row[k % 3] + 16u * (k / 3) < n_blocks;
                    u32x4_u* const at = (u32x4_u*)(unit_dst + (wb_dst[k % 3] + (k / 3) * 16u * row_dst_bytes + 96u * t));
#endif
#if defined(MF_DIAG_NO_STORE)
                    if (valid && n_blocks > 1000000u) *at = op[k];
#elif defined(MF_STORE_PLAIN)
                    if (valid) *at = op[k];
#else
                    if (valid) __builtin_nontemporal_store(op[k], at);
#endif
                }
            }
            if (add_a2) {
#ifndef MF_DIAG_NO_RAW
                split_chunk(cn, raw_a2, false);             // (cn == c_a2 by now)
#endif
                cn++;
            }
            e = e2; f = f2;
            raw_b[0] = raw_b2[0]; raw_b[1] = raw_b2[1]; raw_b[2] = raw_b2[2];
            add_b = add_b2;
        }
        };
        if (wk.flags & kWorkChecked) run_unit(std::true_type{});
        else run_unit(std::false_type{});
        unit = first_claimed + (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
    }
    // The counters reset themselves: a wave reports in after its last claim, and the last wave of the grid zeroes both.
    if (lane == 0) {
        const uint32_t waves_total = gridDim.x * n_waves;
        if (atomicAdd(unit_counter + 1, 1u) == waves_total - 1) {
            __hip_atomic_store(unit_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(unit_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#endif   // OHGPU_LEGACY_KERNELS

// ---- host: the tables ----
// Balanced base-256 digits of a Q28 coefficient: c = e0 + e1 2^8 + e2 2^16 + e3 2^24, e0..e2 in [-128, 127].  False when the top
// digit -- what is left after the three balanced ones -- is no int8: the coefficients from -(128 << 24) - 0x808080 to
// (127 << 24) + 0x7f7f7f have digits (a value above that carries into a top digit of 128, which the cast would wrap to -128).
static bool coef_digits(int32_t c, int8_t e[4])
{
    int64_t v = c;
    for (int j = 0; j < 3; j++) {
        const int d = (int)(int8_t)(uint8_t)(v & 0xff);
        e[j] = (int8_t)d;
        v = (v - d) >> 8;
    }
    e[3] = (int8_t)v;
    return v >= -128 && v <= 127;
}

bool build_mfma_tables(uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, uint32_t L_blk, uint32_t kb_cap,
                       std::vector<uint8_t>* adig, std::vector<MfStep>* steps)
{
    if (T != 32 || L == 0 || M == 0 || L_blk == 0 || (L_blk % 16) != 0 || (L_blk % L) != 0 || kb_cap == 0) return false;
    for (size_t i = 0; i < (size_t)L * T; i++) {
        int8_t e[4];
        if (!coef_digits(coef_q28[i], e)) return false;                                           // the top digit is an int8
    }
    adig->assign((size_t)4 * L * 96, 0);
    std::vector<int64_t> bias(L);
    for (uint32_t p = 0; p < L; p++) {
        int64_t sum = 0;
        for (uint32_t k = 0; k < 32; k++) {
            const int32_t c = coef_q28[(size_t)p * T + k];
            sum += c;
            int8_t e[4];
            coef_digits(c, e);
            for (int j = 0; j < 4; j++) (*adig)[((size_t)j * L + p) * 96 + (63 - k)] = (uint8_t)e[j];
        }
        // the offset digits' constant (128 + 128 * 256 per sample) and the rounding
        bias[p] = 32896 * sum + ((int64_t)1 << 27);
        if (bias[p] < -((int64_t)1 << 44) || bias[p] > ((int64_t)1 << 44)) return false;      // (its bits 16.. ride in ONE accumulator: MfStep)
    }
    const uint32_t n_steps = (L_blk / 16) * kb_cap;
    steps->assign(n_steps, MfStep());
    for (uint32_t t = 0; t < n_steps; t++) {
        MfStep& s = (*steps)[t];
        memset(&s, 0, sizeof(s));
        const uint64_t n0_first = ((uint64_t)16 * t * M) / L;         // newest input frame of the step's first output (row-relative)
        const uint64_t k0 = ((n0_first + 1) / 16) * 16;               // the window: frames k0 .. k0 + 63 in a' = frame + 32
        s.kc = (uint32_t)(k0 / 16);
        for (uint32_t m = 0; m < 16; m++) {
            const uint64_t tm = ((uint64_t)16 * t + m) * M;
            const uint64_t n0 = tm / L;
            const uint32_t p = (uint32_t)(tm % L);
            const int64_t o = 31 + (int64_t)k0 - (int64_t)n0;         // row offset of K = 0
            if (o < 0 || o > 32) return false;                        // (a ratio this tiling does not hold: 15 M / L must stay below 17)
            s.aoff[m] = p * 96 + (uint32_t)o;
            s.b0[m] = (uint32_t)(bias[p] & 0xffff);
            s.b1[m] = (uint32_t)(int32_t)(bias[p] >> 16);
            s.b2[m] = 0;
        }
    }
    return true;
}

// The half-band 2:1 decimator (L = 1, M = 2, T = 64 stored taps of which the even ones and the centre tap 31 are not zero) on the
// same tiles.  y[j] = sum_m c[2 m] x[2 j - 2 m] + c[31] x[2 j - 31]: with the row's image starting 64 frames before the block
// (image frame a = frame + 64), its EVEN frames e = a / 2 and its ODD frames o = (a - 1) / 2 as two sample streams, output j
// meets e = 32 + j - m (m = 0..31) and o = j + 16.  A step of 16 outputs j = 16 s + n therefore reads even samples 16 s ..
// 16 s + 47 -- three aligned chunks, K groups 0..2 -- and the ONE odd chunk s + 1, sample n for output n -- K group 3, a diagonal.
// One coefficient image serves every step (the phase never changes): [digit 4][lane = 16 g + n][16 bytes], and one bias.
bool build_mfma_halfband(const int32_t* coef_q28, uint32_t L_blk, std::vector<MfStep>* steps, std::vector<uint8_t>* amat)
{
    if (L_blk == 0 || (L_blk % 16) != 0) return false;
    int64_t sum = 0;
    for (uint32_t k = 0; k < 64; k++) {
        const int32_t c = coef_q28[k];
        int8_t e4[4];
        if (!coef_digits(c, e4)) return false;                                                    // the top digit is an int8
        if ((k & 1u) && k != 31 && c != 0) return false;                                          // not a half-band filter
        sum += c;
    }
    const int64_t bias = 32896 * sum + ((int64_t)1 << 27);
    if (bias < -((int64_t)1 << 44) || bias > ((int64_t)1 << 44)) return false;
    amat->assign(kMfStepImage, 0);
    for (uint32_t gq = 0; gq < 4; gq++)
        for (uint32_t n = 0; n < 16; n++)
            for (uint32_t i = 0; i < 16; i++) {
                int32_t c = 0;
                if (gq < 3) {
                    const int m_tap = 32 + (int)n - 16 * (int)gq - (int)i;                          // even sample 16 (s + gq) + i against output 16 s + n
                    if (m_tap >= 0 && m_tap <= 31) c = coef_q28[2 * m_tap];
                } else if (i == n) {
                    c = coef_q28[31];
                }
                int8_t e[4];
                coef_digits(c, e);
                for (int j = 0; j < 4; j++) (*amat)[(size_t)j * 1024 + (gq * 16 + n) * 16 + i] = (uint8_t)e[j];
            }
    steps->assign(L_blk / 16, MfStep());
    for (uint32_t t = 0; t < L_blk / 16; t++) {
        MfStep& st = (*steps)[t];
        memset(&st, 0, sizeof(st));
        st.kc = t;                                                                                // even chunks t .. t + 2, odd chunk t + 1
        for (uint32_t m = 0; m < 16; m++) {
            st.b0[m] = (uint32_t)(bias & 0xffff);
            st.b1[m] = (uint32_t)(int32_t)(bias >> 16);
            st.b2[m] = 0;
        }
    }
    return true;
}

// The kernel's A operands, lane-linear: [step][coefficient digit][lane = 16 g + m][16 bytes] = the 16 bytes of output m's padded
// coefficient row that meet frames 16 (kc + g) .. + 15 of the step's window (MfStep::aoff).
void build_mfma_images(const std::vector<uint8_t>& adig, const std::vector<MfStep>& steps, uint32_t L, std::vector<uint8_t>* amat)
{
    amat->assign(steps.size() * (size_t)kMfStepImage, 0);
    for (size_t t = 0; t < steps.size(); t++)
        for (uint32_t j = 0; j < 4; j++)
            for (uint32_t gq = 0; gq < 4; gq++)
                for (uint32_t m = 0; m < 16; m++)
                    memcpy(amat->data() + t * kMfStepImage + j * 1024 + (gq * 16 + m) * 16,
                           adig.data() + (size_t)j * L * 96 + steps[t].aoff[m] + 16 * gq, 16);
}

bool src_mfma_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t db)
{
    return T == 32 && ch == 2 && sb == 3 && db == 3;
}

void src_mfma_geometry(uint32_t* rows, uint32_t* wave_lds_bytes, uint32_t* max_waves)
{
    *rows = 32;
    *wave_lds_bytes = kMfWaveLds;
    *max_waves = OHGPU_MFMA_WAVES;
}

#ifdef OHGPU_LEGACY_KERNELS
template <bool SRC_LE, bool DST_LE>
static hipError_t launch_mfma_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s, uint32_t first_unit)
{
    auto kernel = src_mfma_kernel<SRC_LE, DST_LE>;
    const SrcFastPlan& f = b->fast;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    if (first_unit >= f.n_lean) return hipSuccess;
    const uint32_t n_units = f.n_lean - first_unit;
    uint32_t w = (n_units + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > OHGPU_MFMA_WAVES) w = OHGPU_MFMA_WAVES;
    uint32_t gsz = (n_units + w - 1) / w;
    if (gsz > cus) gsz = cus;
    if ((p.L_blk >> 4) > kMfBiasSteps) return hipErrorInvalidValue;            // (src_mfma_supported keeps such a filter off this kernel)
    const uint32_t lds = kMfBiasBytes + w * kMfWaveLds;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(gsz), dim3(w * 64), lds, s,
                       (const LeanUnit*)f.d_lean_units + first_unit, n_units, (const uint8_t*)f.d_mf_amat, (const MfStep*)f.d_mf_steps,
                       (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst, p.src_arena_bytes, p.L_blk, p.M_blk, (uint32_t*)f.d_counter);
    return hipGetLastError();
}

hipError_t launch_src_mfma(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s, uint32_t first_unit)
{
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    if (prm.src_le) return prm.dst_le ? launch_mfma_one<true, true>(ctx, b, prm, s, first_unit) : launch_mfma_one<true, false>(ctx, b, prm, s, first_unit);
    return prm.dst_le ? launch_mfma_one<false, true>(ctx, b, prm, s, first_unit) : launch_mfma_one<false, false>(ctx, b, prm, s, first_unit);
}

#endif   // OHGPU_LEGACY_KERNELS

}  // namespace ohgpu
