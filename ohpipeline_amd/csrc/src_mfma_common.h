// src_mfma_common.h -- what the two matrix-pipe resampler kernels (src_mfma_kernel.hip: one unit per wave; src_mfma_wg_kernel.hip:
// one unit per workgroup, one output step per wave) share: vector types, the byte permute, the size of a step's A image.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "src_block_common.h"

#ifdef MF_DIAG_NO_MFMA
#define MF_MFMA(a, b, c) ((c) + (a) + (b))
#else
#define MF_MFMA(a, b, c) __builtin_amdgcn_mfma_i32_16x16x64_i8((a), (b), (c), 0, 0, 0)
#endif

namespace ohgpu {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

constexpr uint32_t kMfStepImage = 4 * 1024;         // a step's A operands: [coefficient digit 4][lane 64][16 bytes]

// A value the compiler has to take as it stands, here: keeps a lane offset's widening to 64 bits next to the access that uses it
// (instruction selection works block by block and finds the scalar-base form only when it sees the widening).
__device__ __forceinline__ uint32_t mf_here(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ uint32_t mf_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// Lanes 2 k and 2 k + 1 each hold two pairs of values, (y0, y1) and (y2, y3): the even lane ends with both lanes' (y0, y1), its own
// first, the odd lane with both lanes' (y2, y3), the even lane's first -- four v_cndmask_b32_dpp, the neighbour's value read through
// the select's DPP operand (quad_perm [0, 0, 2, 2]: the even lane's; [1, 1, 3, 3]: the odd lane's).  Written out: the compiler makes a
// copy, a v_mov_b32_dpp and a select of each.  EXEC must be all ones (the tiles' phase); the s_nop is the two wait states a DPP read
// needs behind the vector instruction that wrote its register.
__device__ __forceinline__ void mf_pair_gather(int y0, int y1, int y2, int y3, uint32_t& f0, uint32_t& f1, uint32_t& f2, uint32_t& f3)
{
    const uint64_t even = 0x5555555555555555ull;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %8\n\t"
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"      // even ? y0 : the even lane's y2
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"      // odd ? y2 : the odd lane's y0
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)
        : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "s"(even)
        : "vcc", "scc");
}

// 48 bytes = 8 frames x {L, R} x 3 bytes (twelve dwords as they lie in memory) -> six planes of 8 bytes: pl[3 * channel + byte
// position in the sample][two dwords of four frames each].  A two-level v_perm_b32 network, two permutes per plane dword.
__device__ __forceinline__ void mf_split48(const uint32_t (&w)[12], uint32_t (&pl)[6][2])
{
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
}

// The same planes from eight frames that do NOT lie side by side (a channel pair of a six- or eight-channel stream): frame n's six
// bytes are the low 48 bits of {hi[n], lo[n]}.  Level 1 interleaves two frames' bytes, level 2 two such pairs: 24 permutes again.
__device__ __forceinline__ void mf_split_frames(const uint32_t (&lo)[8], const uint32_t (&hi)[8], uint32_t (&pl)[6][2])
{
    uint32_t t01[4], t23[4], t45[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        t01[m] = mf_perm(lo[2 * m + 1], lo[2 * m], 0x05010400u);     // {x0 y0 x1 y1}: bytes 0 and 1 of frames 2 m (x) and 2 m + 1 (y)
        t23[m] = mf_perm(lo[2 * m + 1], lo[2 * m], 0x07030602u);     // bytes 2 and 3
        t45[m] = mf_perm(hi[2 * m + 1], hi[2 * m], 0x05010400u);     // bytes 4 and 5
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        pl[0][q] = mf_perm(t01[2 * q + 1], t01[2 * q], 0x05040100u); pl[1][q] = mf_perm(t01[2 * q + 1], t01[2 * q], 0x07060302u);
        pl[2][q] = mf_perm(t23[2 * q + 1], t23[2 * q], 0x05040100u); pl[3][q] = mf_perm(t23[2 * q + 1], t23[2 * q], 0x07060302u);
        pl[4][q] = mf_perm(t45[2 * q + 1], t45[2 * q], 0x05040100u); pl[5][q] = mf_perm(t45[2 * q + 1], t45[2 * q], 0x07060302u);
    }
}

// The six accumulators of an output -> floor(acc / 2^28), clamped to 24 bits.  T0 = S0 + (S1 << 8), T1 = S2 + (S3 << 8),
// T2 = S4 + (S5 << 8) (each below 2^30), U = T1 + (T0 >> 16), W = T2 + (U >> 16), y = (W << 4) | bits 12..15 of U: the low 16 bits of
// T0 and the low 12 of U cannot carry into bit 28.  The rounding 2^27 rides in the accumulators' initial values.
__device__ __forceinline__ int mf_recombine(int s0, int s1, int s2, int s3, int s4, int s5)
{
#ifndef MF_RECOMBINE_PLAIN
    // (the two "+ (x >> 16)" as ONE instruction each: an SDWA add whose second source is the sign-extended high word of x -- eight
    // instructions an output instead of the ten the compiler makes of the C below it (shift, three-operand add): 0.3081 -> 0.3022 ms on
    // the headline, same box, three alternating pairs, round 5.  The vector pipe's instruction count is what the tiles' phase is made of)
    const int t0 = (int)(((uint32_t)s1 << 8) + (uint32_t)s0);
    const int t1 = (int)(((uint32_t)s3 << 8) + (uint32_t)s2);
    const int t2 = (int)(((uint32_t)s5 << 8) + (uint32_t)s4);
    int u, w;
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(u) : "v"(t1), "v"(t0));
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(w) : "v"(t2), "v"(u));
    const int yy2 = (int)(((uint32_t)w << 4) | (((uint32_t)u >> 12) & 15u));
    return yy2 < -8388608 ? -8388608 : (yy2 > 8388607 ? 8388607 : yy2);
#else
    const int t0 = (int)(((uint32_t)s1 << 8) + (uint32_t)s0);
    const int u = (int)(((uint32_t)s3 << 8) + (uint32_t)s2) + (t0 >> 16);
    const int w = (int)(((uint32_t)s5 << 8) + (uint32_t)s4) + (u >> 16);
    const int yy = (int)(((uint32_t)w << 4) | (((uint32_t)u >> 12) & 15u));
    return yy < -8388608 ? -8388608 : (yy > 8388607 ? 8388607 : yy);
#endif
}

// ... the same in two parts (the workgroup kernel's pipeline of tiles): U from the first four accumulators, y from U and the last two
__device__ __forceinline__ int mf_recombine_head(int s0, int s1, int s2, int s3)
{
    const int t0 = (int)(((uint32_t)s1 << 8) + (uint32_t)s0);
    const int t1 = (int)(((uint32_t)s3 << 8) + (uint32_t)s2);
    int u;
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(u) : "v"(t1), "v"(t0));
    return u;
}
__device__ __forceinline__ int mf_recombine_tail(int u, int s4, int s5)
{
    const int t2 = (int)(((uint32_t)s5 << 8) + (uint32_t)s4);
    int w;
    asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(w) : "v"(t2), "v"(u));
    const int yy = (int)(((uint32_t)w << 4) | (((uint32_t)u >> 12) & 15u));
    return yy < -8388608 ? -8388608 : (yy > 8388607 ? 8388607 : yy);
}

}  // namespace ohgpu
