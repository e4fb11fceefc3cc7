// src_mfma_kernel.hip -- round 4's resample -> ramp -> pack kernel for 24-bit stereo: the taps on the MATRIX pipe.
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
//     column n -- conflict-free, the hardware's 16-lane groups take complementary columns from two chunks.
//   * THE SPLIT: lane (row = lane / 2, half = lane % 2) loads the 48 bytes of its row's 8 frames of the chunk straight from
//     memory (three unaligned 16-byte loads; no staging buffer), transposes them into six plane dwords pairs with a two-level
//     v_perm_b32 network (2 permutes per plane dword, compile-time selectors), flips the offset digits' top bits and writes six
//     ds_write_b64.  The next chunk's loads are issued before the permutes of this one.
//   * COEFFICIENT DIGITS: per digit and phase a 96-byte row [32 zeros | the phase's 32 digit bytes, oldest tap last | 32 zeros]
//     (global memory, 61 KB for L = 160: L1/L2 resident).  Output m of a step meets input frame k0 + k with tap
//     n0(m) + 32 - k0 - k, so its A row is the 64 bytes of its phase's row from offset 31 + k0 - n0(m) on: lane (m, g) makes
//     ONE unaligned 16-byte load per digit and step, shared by the four column tiles.
//   * RECOMBINATION in 32-bit integers: T0 = S0 + (S1 << 8), T1 = S2 + (S3 << 8), T2 = S4 + (S5 << 8) (each below 2^30),
//     U = T1 + (T0 >> 16), W = T2 + (U >> 16), y = (W << 4) | bits 12..15 of U = floor(acc / 2^28) exactly (the low 16 bits of
//     T0 and the low 12 of U cannot carry into bit 28), then one v_med3_i32.  Ten instructions per output.
//   * PACK: the two channel lanes of a row hold four frames each; they exchange two values (DPP quad_perm), three v_perm_b32
//     make the lane's 12 bytes, ONE global_store_dwordx3 per lane and tile writes them: a row's 16 frames are 96 contiguous
//     bytes, consecutive steps complete the lines in the L2.
// Bit-exact against the integer model (oracle/ohp_pipeline.c) like the kernels before it: same sum, same rounding.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_block_common.h"

namespace ohgpu {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef v4i v4i_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_u __attribute__((aligned(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#ifndef OHGPU_MFMA_WAVES
#define OHGPU_MFMA_WAVES 8                         // waves per workgroup = per CU (LDS: 12 KB each)
#endif
constexpr uint32_t kMfPlaneBytes = 3 * 4 * 4 * 256; // [digit][column tile][chunk slot][column][16]

__device__ __forceinline__ uint32_t mf_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

template <bool SRC_LE, bool DST_LE>
__global__ __launch_bounds__(OHGPU_MFMA_WAVES * 64)
void src_mfma_kernel(const LeanUnit* __restrict__ units, const uint32_t n_work,
                     const uint8_t* __restrict__ adig, const uint32_t adig_stride,
                     const MfStep* __restrict__ steps,
                     const uint16_t* __restrict__ planes, const uint32_t plane_stride,
                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const uint64_t src_arena_bytes,
                     const uint32_t L_blk1, const uint32_t M_blk1, uint32_t* __restrict__ unit_counter)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t n_waves = blockDim.x >> 6;
    uint8_t* const wl = smem + wave * kMfPlaneBytes;

    // lane roles.  Matrix operands and results: g = K group / output quad, n = A row (output of the step) / column of the tile.
    const uint32_t g = lane >> 4, n = lane & 15;
    const uint32_t rr = n >> 1, ch = n & 1;                // the column's row within its tile, its channel
    // the split: row and half chunk
    const uint32_t rs = lane >> 1, hs = lane & 1;
    uint8_t* const split_lds = wl + (rs >> 3) * 1024 + (rs & 7) * 32 + hs * 8;   // + digit * 4096 + slot * 256 + channel * 16
    const uint8_t* const b_lds = wl + n * 16;                                     // + (digit * 4 + tile) * 1024 + slot * 256

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
        const bool checked = (wk.flags & kWorkChecked) != 0;
        const bool first = (wk.flags & kWorkFirst) != 0;
        const uint32_t n_steps = L_blk >> 4;
        const uint32_t c_last = (M_blk + 31u) >> 4;                     // the last chunk any output of a row needs

        // the split's source: frame 16 c + 8 hs of row rs (rows past the unit's last re-read row 0)
        const int64_t split_off = wk.src_row0 + (int64_t)((rs < n_blocks ? rs : 0u) * M_blk) * 6 + 48 * (int64_t)hs;
        const uint8_t* const split_src = src + split_off;
        const bool zero_history = first && rs == 0;                    // the stream's block 0: frames before it read as zeros

        u32x4 raw[3];
        auto load_chunk = [&](uint32_t c) __attribute__((always_inline)) {
            if (!checked) {
#pragma unroll
                for (int k = 0; k < 3; k++) raw[k] = *(const u32x4_u*)(split_src + (uint64_t)c * 96 + 16 * k);
            } else {
                // a unit at an end of the arena: a 16-byte piece that is not wholly inside is fetched byte by byte, bytes outside
                // read as zero (they are frames before a stream's first or beyond the last frame any output needs)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int64_t a = split_off + (int64_t)c * 96 + 16 * k;
                    if (a >= 0 && (uint64_t)a + 16 <= src_arena_bytes) raw[k] = *(const u32x4_u*)(src + a);
                    else {
                        uint32_t w[4] = {0, 0, 0, 0};
                        for (int bb = 0; bb < 16; bb++) {
                            const int64_t a1 = a + bb;
                            if (a1 >= 0 && (uint64_t)a1 < src_arena_bytes) w[bb >> 2] |= (uint32_t)src[a1] << (8 * (bb & 3));
                        }
                        raw[k] = u32x4{w[0], w[1], w[2], w[3]};
                    }
                }
            }
        };
        // 48 bytes = 8 frames x {L, R} x 3 bytes -> six planes of 8 bytes; c0 = 3 * channel + byte position in the sample
        auto split_chunk = [&](uint32_t c, bool prefetch) __attribute__((always_inline)) {
            uint32_t w[12] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w, raw[2].x, raw[2].y, raw[2].z, raw[2].w};
            if (first && c < 2) {
#pragma unroll
                for (int k = 0; k < 12; k++) w[k] = zero_history ? 0u : w[k];
            }
            if (prefetch) load_chunk(c + 1);
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

        // the epilogue's addresses
        const uint64_t dst_lane = (uint64_t)wk.dst_row0 + 24u * g + 12u * ch;                 // + row * L_blk * 6 + 96 t
        const uint8_t* const mbase = (const uint8_t*)planes + (uint64_t)wk.plane * plane_stride;

        uint32_t cn = 0;                                   // next chunk to split
        load_chunk(0);
        // step 0's operands
        MfStep const* st = steps;
        uint32_t kc = st->kc;
        uint32_t aoff = st->aoff[n] + 16u * g;
        v4i bias0 = *(const v4i*)(st->b0 + 4 * g), bias1 = *(const v4i*)(st->b1 + 4 * g), bias2 = *(const v4i*)(st->b2 + 4 * g);
        uint32_t claim = 0;
        for (uint32_t t = 0; t < n_steps; t++) {
            // every chunk of the step's window is in the planes
            uint32_t c_need = kc + 3u;
            if (c_need > c_last) c_need = c_last;
            while (cn <= c_need) {
                split_chunk(cn, cn < c_last);
                cn++;
            }
            // the coefficient digits of the step's 16 outputs: lane (n, g) holds output n's taps against frames 16 (kc + g) ..
            v4i a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = *(const v4i_u*)(adig + (uint64_t)j * adig_stride + aoff);
            const v4i b0 = bias0, b1 = bias1, b2 = bias2;
            const uint32_t slot_g = ((kc + g) & 3u) * 256u;
            // the next step's table entries ride under this step's work
            if (t + 1 < n_steps) {
                st = steps + (t + 1);
                kc = st->kc;
                aoff = st->aoff[n] + 16u * g;
                bias0 = *(const v4i*)(st->b0 + 4 * g); bias1 = *(const v4i*)(st->b1 + 4 * g); bias2 = *(const v4i*)(st->b2 + 4 * g);
            } else if (lane == 0) {
                claim = atomicAdd(unit_counter, 1u);       // the next unit, claimed at this one's last step
            }
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                v4i bd[3];
#pragma unroll
                for (int d = 0; d < 3; d++) bd[d] = *(const v4i*)(b_lds + (d * 4 + ct) * 1024 + slot_g);
                v4i s0 = b0, s1 = v4i{0, 0, 0, 0}, s2 = b1, s3 = v4i{0, 0, 0, 0}, s4 = b2, s5 = v4i{0, 0, 0, 0};
                s0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bd[0], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bd[0], s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bd[0], s2, 0, 0, 0);
                s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], bd[0], s3, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bd[1], s1, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bd[1], s2, 0, 0, 0);
                s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bd[1], s3, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], bd[1], s4, 0, 0, 0);
                s2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0], bd[2], s2, 0, 0, 0);
                s3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1], bd[2], s3, 0, 0, 0);
                s4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[2], bd[2], s4, 0, 0, 0);
                s5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[3], bd[2], s5, 0, 0, 0);
                // ---- recombine, round (the bias carries 2^27), clamp: the lane's four frames 16 t + 4 g + v of column n ----
                const uint32_t row = (uint32_t)ct * 8u + rr;
                const bool valid = row < n_blocks;
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
                    // RampApplicator::GetNextSample on the 24-bit value (Msg.cpp:840-895): top 16 bits * Q15 >> 15, low byte zero;
                    // the multipliers of the row's four frames come from the unit's plane (0xffff: the frame's message has no ramp)
                    const u32x2 mm = *(const u32x2*)(mbase + ((uint64_t)(valid ? row * L_blk : 0u) + 16u * t + 4u * g) * 2u);
                    const uint32_t mu[4] = {mm.x & 0xffffu, mm.x >> 16, mm.y & 0xffffu, mm.y >> 16};
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int top = (int)((uint32_t)y[v] << 8) >> 16;  // bits 8..23, signed
                        const int r = (int)((uint32_t)((top * (int)mu[v]) >> 15) << 8);
                        y[v] = mu[v] != 0xffffu ? r : y[v];
                    }
                }
                // ---- pack: exchange two values with the other channel's lane, three permutes, one 12-byte store ----
                const int give_a = ch ? y[0] : y[2], give_b = ch ? y[1] : y[3];
                const int own_a = ch ? y[2] : y[0], own_b = ch ? y[3] : y[1];
                const uint32_t got_a = (uint32_t)__builtin_amdgcn_mov_dpp(give_a, 0xb1, 0xf, 0xf, true);    // quad_perm:[1,0,3,2]
                const uint32_t got_b = (uint32_t)__builtin_amdgcn_mov_dpp(give_b, 0xb1, 0xf, 0xf, true);
                const uint32_t r_first = ch ? (uint32_t)own_a : got_a;      // R of the lane's first frame
                const uint32_t l_second = ch ? got_b : (uint32_t)own_b;     // L of its second
                u32x3 o;
                o.x = mf_perm(got_a, (uint32_t)own_a, sel_d0);
                o.y = mf_perm(l_second, r_first, sel_d1);
                o.z = mf_perm(got_b, (uint32_t)own_b, sel_d2);
                if (valid) *(u32x3_u*)(dst + dst_lane + (uint64_t)row * L_blk * 6u + 96u * (uint64_t)t) = o;
            }
        }
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

// ---- host: the tables ----
// Balanced base-256 digits of a Q28 coefficient: c = e0 + e1 2^8 + e2 2^16 + e3 2^24, e0..e2 in [-128, 127].
static void coef_digits(int32_t c, int8_t e[4])
{
    int64_t v = c;
    for (int j = 0; j < 4; j++) {
        int d = (int)(int8_t)(uint8_t)(v & 0xff);
        if (j == 3) d = (int)v;
        e[j] = (int8_t)d;
        v = (v - d) >> 8;
    }
}

bool build_mfma_tables(uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, uint32_t L_blk, uint32_t kb_cap,
                       std::vector<uint8_t>* adig, std::vector<MfStep>* steps)
{
    if (T != 32 || L == 0 || M == 0 || L_blk == 0 || (L_blk % 16) != 0 || (L_blk % L) != 0 || kb_cap == 0) return false;
    for (size_t i = 0; i < (size_t)L * T; i++)
        if (coef_q28[i] > (127 << 24) + 0x7fffff || coef_q28[i] < -(127 << 24)) return false;     // the top digit is an int8
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
        if (bias[p] < -((int64_t)1 << 46) || bias[p] > ((int64_t)1 << 46)) return false;
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
            s.b1[m] = (uint32_t)((bias[p] >> 16) & 0xffff);
            s.b2[m] = (uint32_t)(int32_t)(bias[p] >> 32);
        }
    }
    return true;
}

bool src_mfma_supported(uint32_t T, uint32_t ch, uint32_t sb, uint32_t db)
{
    return T == 32 && ch == 2 && sb == 3 && db == 3;
}

void src_mfma_geometry(uint32_t* rows, uint32_t* wave_lds_bytes, uint32_t* max_waves)
{
    *rows = 32;
    *wave_lds_bytes = kMfPlaneBytes;
    *max_waves = OHGPU_MFMA_WAVES;
}

template <bool SRC_LE, bool DST_LE>
static hipError_t launch_mfma_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s)
{
    auto kernel = src_mfma_kernel<SRC_LE, DST_LE>;
    const SrcFastPlan& f = b->fast;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint32_t w = (f.n_lean + cus - 1) / cus;
    if (w < 1) w = 1;
    if (w > OHGPU_MFMA_WAVES) w = OHGPU_MFMA_WAVES;
    uint32_t gsz = (f.n_lean + w - 1) / w;
    if (gsz > cus) gsz = cus;
    const uint32_t lds = w * kMfPlaneBytes;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(gsz), dim3(w * 64), lds, s,
                       (const LeanUnit*)f.d_lean_units, f.n_lean, (const uint8_t*)f.d_mf_adig, f.mf_adig_stride, (const MfStep*)f.d_mf_steps,
                       (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst, p.src_arena_bytes, p.L_blk, p.M_blk, (uint32_t*)f.d_counter);
    return hipGetLastError();
}

hipError_t launch_src_mfma(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    if (prm.src_le) return prm.dst_le ? launch_mfma_one<true, true>(ctx, b, prm, s) : launch_mfma_one<true, false>(ctx, b, prm, s);
    return prm.dst_le ? launch_mfma_one<false, true>(ctx, b, prm, s) : launch_mfma_one<false, false>(ctx, b, prm, s);
}

}  // namespace ohgpu
