// src_block_common.h -- what the two block resampler kernels (src_lean_kernel.hip, src_block_kernel.hip) and their host
// glue share: address-space typedefs, the compile-time loop, the output ring's geometry, the instantiation list.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace ohgpu {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* global_ptr_t;
typedef __attribute__((address_space(3))) uint8_t* lds_u8_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// calls f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a compile-time unrolled loop
template <typename F, int... S>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, S...>)
{
    (f(std::integral_constant<int, S>{}), ...);
}

// Output ring: every block row owns `ring_bytes` of LDS that hold its packed output byte stream modulo ring_bytes.
// ring_bytes is a multiple of the frame size (a frame never wraps) and of 16 (a 16-byte piece of a line never
// wraps).  After a drain fewer than 64 bytes (a multiple of g = gcd(64, fb_dst)) are pending, and `out_per_drain`
// more frames arrive before the next drain.  Rows are 4 bytes further apart so that they start in different banks.
static constexpr uint32_t gcd_c(uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; }
static constexpr uint32_t ring_bytes_for(uint32_t fb_dst, uint32_t out_per_drain, bool pair)
{
    // pair mode (24-bit stereo): frames enter the ring two at a time, so one more frame can be waiting and the
    // pending bytes are a multiple of gcd(64, 2 * fb_dst)
    const uint32_t step = pair ? 2 * fb_dst : fb_dst;
    const uint32_t unit = step * 16 / gcd_c(step, 16);
    const uint32_t need = (64 - gcd_c(64, step)) + (out_per_drain + (pair ? 1 : 0)) * fb_dst;
    return unit * ((need + unit - 1) / unit);
}
static constexpr bool ring_pair_mode(uint32_t ch, uint32_t db) { return ch == 2 && db == 3; }
static constexpr uint32_t kRampLdsBytes = 1024;         // RampArray.h's 512 Q15 multipliers, kept after the coefficient table

// lean kernel: 16-byte pieces per staged row.  The frames of a stage start anywhere in their first piece, so
// the bytes that are USED span at most 15 + frames * fb_src bytes.  Stereo rows drift against the banks by themselves (the
// rows of a unit are M_blk frames apart, never a whole number of pieces for the stereo layouts); wider frames get an
// odd count so that the rows start in different banks.
#ifndef OHGPU_LEAN_STAGE_FRAMES
#define OHGPU_LEAN_STAGE_FRAMES 16
#endif
// frames per stage: stereo rows take OHGPU_LEAN_STAGE_FRAMES at a time (each row's request then covers most of a 128-byte
// line: with 8 frames -- 48 bytes of a 6-byte-frame row -- every line was requested by three or four stages and fetched
// from memory 1.8 times), wider frames 8
// (mono and the odd channel counts: 16 frames too -- a stage must advance every row by whole 16-byte pieces, and 8 frames of
// 3, 9, 15 or 21 bytes do not)
static constexpr int lean_stage_frames(int ch) { return (ch <= 2 || (ch & 1)) ? OHGPU_LEAN_STAGE_FRAMES : 8; }
static constexpr int lean_in_blocks(int ch, int sb)
{
    const int n = (lean_stage_frames(ch) * ch * sb + 14) / 16 + 1;
    return ch == 2 ? n : (n | 1);
}

static constexpr int lean_in_blocks_planar(int ch) { return ((lean_stage_frames(ch) * 4 + 14) / 16 + 1) | 1; }   // a staged row = one channel's 4-byte frames

// SrcWork::flags
enum { kWorkRamped = 1u,      // a ramped message overlaps the unit's output range
       kWorkChecked = 2u,     // some staging piece of the unit lies outside the source arena (ends of the arena)
       kWorkFirst = 4u,       // LeanUnit only: row 0 is its stream's block 0 (the frames before it read as zeros)
       kWorkEdge = 8u };      // LeanUnit of a src_mfma_wg_kernel plan: the unit's 32-row input image leaves the arena -- that kernel fetches its pieces through its checked, out-of-line load

}  // namespace ohgpu

// ---- instantiations: (T, channels, source bytes, source LE, destination bytes, destination LE) ----
// The list is compiled in parts so that the build can run them side by side (ohpipeline_amd/build.py compiles each kernel
// file once per part with -DOHGPU_BLOCK_PART=k): part 1 also holds the host code and only DECLARES the other parts' kernels;
// parts 2.. hold nothing but their kernels.  Without the macro (tools, tests) a file is one translation unit.
#ifdef OHGPU_DIAG_ONE_KERNEL
#define OHGPU_BLOCK_KERNELS_1(X) X(32, 2, 3, true, 3, false)
#define OHGPU_BLOCK_KERNELS_2(X)
#define OHGPU_BLOCK_KERNELS_3(X)
#else
#define OHGPU_BLOCK_KERNELS_1(X)    \
    X(32, 2, 3, true, 3, false)     \
    X(32, 2, 3, true, 3, true)      \
    X(32, 2, 3, false, 3, false)    \
    X(32, 2, 3, true, 4, false)     \
    X(32, 2, 3, true, 2, false)     \
    X(32, 2, 2, true, 3, false)     \
    X(32, 2, 2, true, 2, true)      \
    X(32, 2, 2, true, 2, false)
#define OHGPU_BLOCK_KERNELS_2(X)    \
    X(32, 6, 3, true, 3, false)     \
    X(32, 6, 3, false, 3, false)    \
    X(32, 8, 3, true, 3, false)     \
    X(32, 8, 3, false, 3, false)    \
    X(32, 4, 3, false, 3, false)    \
    X(32, 6, 2, false, 3, false)
#define OHGPU_BLOCK_KERNELS_3(X)    \
    X(64, 2, 3, true, 3, false)     \
    X(64, 6, 3, true, 3, false)     \
    X(64, 8, 3, true, 3, false)     \
    X(32, 2, 2, false, 3, false)    \
    X(32, 2, 2, false, 2, false)
#endif
#define OHGPU_BLOCK_KERNELS(X) OHGPU_BLOCK_KERNELS_1(X) OHGPU_BLOCK_KERNELS_2(X) OHGPU_BLOCK_KERNELS_3(X)
// Round 1's block kernel in the SHIPPED library (round 5: retired as a selectable variant, legacy builds have the whole list above):
// only as the fallback for a filter whose phase sums reach 2^29 -- beyond the lean kernel's rounding bias, which is what keeps its
// fp64 sums exact -- and only for the stereo layouts such a filter is likely to meet.  ohgpu_src_design's own 48 -> 44.1 kHz and
// 32 -> 48 kHz filters are such filters (sum|c| = 2.02 and 2.40 x 2^28); without these they would run on the generic kernel.
#define OHGPU_BLOCK_FALLBACK_KERNELS(X) \
    X(32, 2, 3, true, 3, false)         \
    X(32, 2, 3, true, 3, true)          \
    X(32, 2, 3, false, 3, false)        \
    X(32, 2, 2, true, 3, false)         \
    X(32, 2, 2, false, 3, false)
// the lean kernel's planar-source instantiations (source bytes 0 = the TInt32 planes of OHGPU_FLAG_SRC_PLANAR32), compiled with part 3
#ifdef OHGPU_DIAG_ONE_KERNEL
#define OHGPU_LEAN_PLANAR_KERNELS(X)
#else
#define OHGPU_LEAN_PLANAR_KERNELS(X) \
    X(32, 2, 0, true, 3, false)      \
    X(32, 2, 0, true, 3, true)
#endif
// the lean kernel's half-band instantiations (a filter with ohgpu_src::halfband: T = 64 stored, 33 products per output), part 4.
// Each has a plain T = 64 twin in the list above, which serves every other 64-tap filter of the same layout.
#ifdef OHGPU_DIAG_ONE_KERNEL
#define OHGPU_LEAN_HB_KERNELS(X)
#else
#define OHGPU_LEAN_HB_KERNELS(X)     \
    X(64, 2, 3, true, 3, false)      \
    X(64, 6, 3, true, 3, false)      \
    X(64, 8, 3, true, 3, false)
#endif
// layouts only the lean kernel is instantiated for (round 1's kernel, variant 2, leaves them to the generic one): packed 32-bit
// stereo sources, mono (64 blocks per wave), wide little-endian outputs, and the channel counts that complete 1..8 for S24
// little-endian sources (Msg.h:171 admits 1 to 8 channels): 3, 4, 5, 7.  Part 5.
#ifdef OHGPU_DIAG_ONE_KERNEL
#define OHGPU_LEAN_ONLY_KERNELS(X)
#else
#define OHGPU_LEAN_ONLY_KERNELS(X)   \
    X(32, 2, 4, true, 3, false)      \
    X(32, 2, 4, false, 3, false)     \
    X(32, 1, 3, true, 3, false)      \
    X(32, 1, 3, false, 3, false)     \
    X(32, 1, 2, true, 3, false)      \
    X(32, 1, 2, false, 3, false)     \
    X(32, 6, 3, true, 3, true)       \
    X(32, 8, 3, true, 3, true)       \
    X(32, 3, 3, true, 3, false)      \
    X(32, 4, 3, true, 3, false)      \
    X(32, 5, 3, true, 3, false)      \
    X(32, 7, 3, true, 3, false)
#endif
// ... and (round 4, part 6) the layouts that were the generic kernel's: odd channel counts from big-endian S24, 16- and 32-bit big-endian
// destinations for six and eight channels, 8-bit stereo sources.
#ifdef OHGPU_DIAG_ONE_KERNEL
#define OHGPU_LEAN_MORE_KERNELS(X)
#else
#define OHGPU_LEAN_MORE_KERNELS(X)   \
    X(32, 3, 3, false, 3, false)     \
    X(32, 5, 3, false, 3, false)     \
    X(32, 7, 3, false, 3, false)     \
    X(32, 6, 3, true, 2, false)      \
    X(32, 8, 3, true, 2, false)      \
    X(32, 6, 3, true, 4, false)      \
    X(32, 8, 3, true, 4, false)      \
    X(32, 2, 1, false, 3, false)
#endif
#define OHGPU_BLOCK_PARTS 6
