// src_mfma_wg_kernel.hip -- the matrix-pipe resampler (src_mfma_kernel.hip has the arithmetic: int8 digit planes, twelve
// v_mfma_i32_16x16x64_i8 per tile of 16 outputs x 16 columns, 32-bit recombination) with the work cut for the MEMORY system:
// one unit per WORKGROUP, five tiles per WAVE and pass.
//
// Why.  With a unit per wave (src_mfma_kernel.hip) a wave reads 96 bytes of each of its 32 rows per step and writes 192 bytes of
// each per pair of steps.  Every byte is fetched once and every sector written whole (1.09 x the algorithmic traffic), and still
// the launch is bound by exactly that access pattern: with the arithmetic compiled out it takes 0.41 ms of the 0.42, because
// 65 000 row streams advancing a hundred bytes at a time leave no DRAM page open for its next visitor.  The same kernel moving
// the same bytes as one contiguous 6 KB run per wave and pair of steps takes 0.30 ms (`tools/exp_mfma.sh`, MF_DIAG_IO_CONTIG).
// So a unit has to arrive and leave in ONE piece.
//
// How.  A planner unit is 30 or 32 CONSECUTIVE blocks of a stream (src_plan.cpp: rows of one block), its input and its output each
// contiguous in memory; a workgroup takes sixteen pair-rows of one at a time -- a PASS: 14 KB in, 15 KB out.  A block is 10 steps of
// 16 output frames, a tile 16 outputs x 16 columns = 8 channel PAIRS; a step's coefficient image is the same for every block: a
// wave owns five of the pass's 20 tiles in step-major order and keeps the A operands of the three steps they touch in registers for
// the whole launch -- no table is read in the loop.  Per pass the workgroup
//   (A) has the rows' input in an LDS image: packed sources through the polyphase filters as ONE run of 16-byte pieces, the union of
//       the overlapping rows, fetched a pass ahead STRAIGHT into a buffer of its own (global_load_lds_dwordx4: WgGeom::kDma, round 5);
//       the planar and half-band forms row by row through registers (loaded a pass ahead) and 16-byte LDS stores,
//   (S) splits it into the digit planes (lane = eight frames of one channel pair: src_mfma_common.h),
//   (C) runs the 20 tiles -- a wave's five as a software pipeline, a tile's last vector instructions between the next tile's matrix
//       instructions -- packed results into an LDS image of the pass's output, twelve bytes a lane and tile in two stores,
//   (D) writes that image out: 15 KB contiguous, whole 16-byte pieces, non-temporal,
// with a workgroup barrier behind (S) and behind (C) -- (D) and (S) touch nothing of each other's, so kDma needs none between them;
// the row-by-row forms, whose input image and output image share their LDS, have one more behind (A) -- three workgroups per CU
// (kDma: 50-52 KB of LDS each), the next pass's input in flight during (C).
// Six and eight channels (PAIRS = 3, 4) are the same sixteen pair-rows cut differently: 5 stream rows x 3 pairs (the sixteenth
// column pair idles) or 4 x 4; only the addresses of the split's reads and of the tiles' stores know.  HB is the 96 -> 48 kHz
// half-band decimator on the same tiles (WgGeom below), PLANAR the FLAC decoder's TInt32 planes as the source.
// Units whose input image does not lie wholly inside the source arena (kWorkEdge: the first of the first stream, the last of
// the last) fetch their pieces through a checked, out-of-line load.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ohgpu_internal.h"
#include "pcm_device.h"
#include "src_mfma_common.h"

#ifndef OHGPU_WG_ROWS
#define OHGPU_WG_ROWS 16                            // rows (blocks) a workgroup takes at a time: 16 (three workgroups per CU) or 32 (two)
#endif

namespace ohgpu {

constexpr uint32_t kWgPlaneIn = 12 * 64;            // planar TInt32 source: bytes of one channel's frames of a row's image (two of them side by side)

// A workgroup's pass is ROWS pair-rows (a pair-row = one channel pair of one block of a stream) = CT = ROWS / 8 column tiles per
// step, STEPS x CT tiles in all; a wave takes FIVE (four: half-band) of them in step-major order (2 CT waves: two per SIMD for 32
// rows, one for 16 -- whole numbers per SIMD whatever SIMD the first wave lands on; ten waves, one per step, would be 3, 3, 2, 2,
// and the phase between two barriers lasts as long as its slowest wave).  PAIRS = channels / 2: the pass holds ROWS / PAIRS whole
// stream rows.
// HB: the half-band 2:1 decimator (build_mfma_halfband): blocks of 128 outputs = 8 steps from 256 + 64 input frames, a row's image
// is 20 input chunks and its planes 10 chunks of EVEN frames (ids 0..9) and 10 of ODD ones (ids 10..19); a step's K groups 0..2
// read even chunks step .. step + 2, group 3 the odd chunk step + 1.  The image is staged in TWO sections (rounds 0..4 of the
// lanes' pieces = input chunks 0..12, then rounds 4..7 = chunks 13..19), so that three workgroups still share a CU's LDS.
template <int ROWS, int PLANAR = 0, int PAIRS = 1, bool HB = false>
struct WgGeom {
    static_assert(ROWS == 16 || ROWS == 32, "pair-rows per workgroup");
    static_assert(PAIRS == 1 || ((PAIRS == 3 || PAIRS == 4) && PLANAR == 0 && ROWS == 16), "six and eight channels: packed, sixteen pair-rows");
    static_assert(!HB || (PLANAR == 0 && ROWS == 16), "half-band: packed, sixteen pair-rows");
    static_assert(PLANAR >= 0 && PLANAR <= 4, "source format: 0 packed S24, 1..3 the TInt32 planes, 4 packed S16");
    static constexpr bool kPlanes = PLANAR >= 1 && PLANAR <= 3;       // the decoder's TInt32 planes
    static constexpr bool kS16 = PLANAR == 4;                        // packed 16-bit frames: the sample is the 16 bits shifted up a byte
    static constexpr uint32_t kSteps = HB ? 8 : 10;                   // steps (16 output frames) per block
    static constexpr uint32_t kOutFrames = 16u * kSteps;              // a block's outputs: 160, or 128
    static constexpr uint32_t kImgChunks = HB ? 20 : 12;              // chunks (16 frames) of a row's input image: frames -32 .. 159 of the row, or -64 .. 255
    static constexpr uint32_t kPlaneChunks = kImgChunks;              // ... and of its digit planes (half-band: ten of even frames, then ten of odd ones)
    static constexpr uint32_t kFb = 6u * PAIRS;                       // bytes of an output frame, and of a packed S24 input frame
    static constexpr uint32_t kFbIn = kS16 ? 4u * PAIRS : kFb;        // bytes of a packed input frame
    static constexpr uint32_t kSR = ROWS / PAIRS;                     // stream rows of a pass: 16, 5 (fifteen of the sixteen pair-rows) or 4
    static constexpr uint32_t kUnitRows = PAIRS == 3 ? 30u : 32u;     // rows of a planner unit (LeanUnit: src_plan.cpp cuts them this long for this kernel): a whole number of passes
    static constexpr uint32_t kRowIn = kImgChunks * 16u * kFbIn;      // bytes of a row's input image, packed
    static constexpr uint32_t kRowLanes = 16u * PAIRS;                // lanes that move one row of the input image, 4.5 (7.5) pieces each (planar: 6)
    static constexpr uint32_t kRound = 16u * kRowLanes;               // ... and the bytes of it one round of them moves
    static constexpr uint32_t kFullRounds = HB ? 7 : 4;               // whole rounds; then half a round
    // Packed sources (but the half-band form): a pass's rows overlap -- row r + 1 starts M_blk frames after row r and an image is 192
    // frames long -- so the pass fetches their UNION, one run of at most 1024 16-byte pieces from the first row's start (four rounds
    // of the workgroup's lanes, every instruction 4 KB contiguous), and the split finds a row at r * row_src_bytes in it: a fifth
    // fewer loads and stage writes than row by row, and 5 % off the headline launch (0.317 -> 0.300 ms, same box, before the split paid for it).
    static constexpr bool kSpan = !kPlanes && !HB;
    static constexpr uint32_t kSpanRounds = kS16 ? 3 : 4;             // (16-bit stereo: at most 648 pieces)
    static constexpr uint32_t kSpanBytes = kSpanRounds * 256u * 16u;
    // The half-band form's rows overlap too -- a row starts 256 frames after its predecessor, its image is 320 long -- and can be fetched
    // as ONE run as well (kHbRun): a fifth fewer pieces loaded, and staged in two sections cut BY ROWS (the run's first kHbRows0 rows,
    // then the rest; the 64 frames the two share are staged twice), since the run and the planes do not fit the LDS side by side.
    // Round 5, same box, turn and turn about (gpurun_out/r5/exp_hbrun.log, ms): six channels 1.825 -> 1.752 (-4 %), stereo 0.574 -> 0.583
    // (+1.5 %: sixteen rows, and the split's sixteen lanes a row apart pay for the run's fixed row distance even with the spare bytes
    // below), eight channels 2.20-2.30 -> 2.31-2.33 (its first section is one row: 160 tasks for 256 lanes).  So: six channels.
#if defined(MF_WG_HB_ROWS)
    static constexpr bool kHbRun = false;               // (A/B: every row's image fetched on its own, staged in two sections cut by chunks)
#elif defined(MF_WG_HB_RUN_ALL)
    static constexpr bool kHbRun = HB;                  // (A/B: every channel count)
#else
    static constexpr bool kHbRun = HB && PAIRS == 3;
#endif
    static constexpr uint32_t kHbAdv = 256u * kFb;                                   // bytes from a row's image to the next row's
    static constexpr uint32_t kHbRunBytes = (kSR - 1u) * kHbAdv + kRowIn;            // the rows' union: 24960 (stereo), 24192, 26112
    static constexpr uint32_t kHbRunPieces = kHbRunBytes / 16u;
    static constexpr uint32_t kHbRunRounds = (kHbRunPieces + 255u) / 256u;
    static constexpr uint32_t kHbRows0 = PAIRS == 1 ? 6u : (PAIRS == 3 ? 2u : 1u);   // (240 / 240 / 160 split tasks: one round of the lanes; the rest: two)
    static constexpr uint32_t kHbSec0Hi = (kHbRows0 - 1u) * kHbAdv + kRowIn;         // section 0 = the run's bytes [0, kHbSec0Hi)
    static constexpr uint32_t kHbSec1Lo = kHbRows0 * kHbAdv;                         // section 1 = [kHbSec1Lo, kHbRunBytes)
    // (in LDS every 256 frames of the run are followed by 16 spare bytes: a row's image starts 256 frames x 6, 18 or 24 bytes after its
    // predecessor's -- a multiple of 256 bytes, every row on the same banks -- and sixteen lanes of the split read sixteen rows at once:
    // without the spare bytes the stereo half-band group took 0.82 ms instead of 0.57.  A row's image then has its hole behind chunk 15)
    static constexpr uint32_t kHbStage = (kHbSec0Hi > kHbRunBytes - kHbSec1Lo ? kHbSec0Hi : kHbRunBytes - kHbSec1Lo) + 16u * kSR;
    static_assert(!kHbRun || (kHbRunBytes % 16u == 0 && kHbSec0Hi % 16u == 0 && kHbSec1Lo % 16u == 0), "whole pieces");
    static constexpr uint32_t kInRounds = kPlanes ? 6 : (kSpan ? kSpanRounds : (kHbRun ? kHbRunRounds : kFullRounds + 1));
    static constexpr uint32_t kRoundsA = HB ? 5 : kInRounds;          // half-band: the rounds of the first staging (the second: kRoundsA - 1 ..)
    static constexpr uint32_t kChunksA = HB ? 13 : kImgChunks;        // ... and the input chunks it holds whole
    static constexpr uint32_t kRowInPitch = kPlanes ? 2 * kWgPlaneIn + 16 : (HB ? kRoundsA * kRound + 16 : kRowIn + 16);   // (sixteen stereo rows, 16 bytes each, then meet all 64 banks once)
    static constexpr uint32_t kRowOut = kOutFrames * kFb;             // bytes of a row's output
    // ... and the distance between two rows of the OUTPUT IMAGE in LDS: the same, unless MF_WG_OUT_PAD is set.  Six channels: frames 18
    // bytes apart put two of a row's sixteen frames of a step into every bank, and with rows 2880 bytes = 720 dwords apart (16 mod 32)
    // the two pair-rows a 32-lane half of the epilogue's 2-byte stores serves land on the same banks again -- 5.25 LDS cycles per
    // store by the bank model where stereo's take 4 and eight channels' 2; rows 64 bytes further apart meet them at 4
    // (tools/micro/lds_conflicts.py).  Built and measured in round 5 (the image then is no longer the output piece for piece: phase (D)
    // finds piece f at (f / 180) * pitch + 16 (f % 180) and the barrier between (D) and (A) is back): bit-exact, and config 4's
    // six-channel 44.1 kHz group took 1.2965 / 1.2822 / 1.2840 ms with the pad against 1.2770 / 1.2782 / 1.2828 without, same box, turn
    // and turn about -- a 2- to 3-way conflict on a 2-byte store costs no time (the store's four cycles are its data's way to the LDS,
    // MI355X_MICROARCH.md), the barrier costs what it cost before.  Not the default.
#ifdef MF_WG_OUT_PAD
    static constexpr uint32_t kRowOutPitch = kRowOut + (PAIRS == 3 && !HB ? 64u : 0u);
#else
    static constexpr uint32_t kRowOutPitch = kRowOut;
#endif
    static constexpr bool kOutLinear = kRowOutPitch == kRowOut;
    static constexpr uint32_t kCt = ROWS / 8;
    static constexpr uint32_t kWaves = 2 * kCt;
    static constexpr uint32_t kThreads = 64 * kWaves;                 // = 16 * ROWS
    static constexpr uint32_t kTilesPerWave = kSteps * kCt / kWaves;  // 5, or 4
    static constexpr uint32_t kKcSets = HB ? 2 : (kCt == 4 ? 2 : 3);  // steps a wave's tiles touch
    static constexpr uint32_t kASets = HB ? 1 : kKcSets;              // ... and the coefficient images they need (half-band: the one there is)
    static constexpr uint32_t kHalf = kCt * 128;                      // a digit plane's chunk: [half of its frames 2][column tile][column 16][8 frames]
    static constexpr uint32_t kChunk = 2 * kHalf;
    static constexpr uint32_t kDigit = kPlaneChunks * kChunk;
    static constexpr uint32_t kPlaneBytes = 3 * kDigit;
    // [step][b0, b1][output 16][4 copies] dwords: an output's two initial values as the four-register C operands of its tile (bits 16.. of
    // the constant ride in ONE accumulator, MfStep: two 16-byte reads per tile, not three; half-band: the same for every output and step --
    // they stay in eight registers, no table)
    // Packed sources through the polyphase filters (kDma): the pass's run goes from memory STRAIGHT into an LDS buffer of its own
    // (global_load_lds_dwordx4), not through registers and four 16-byte LDS stores a lane: phase (A) and its barrier are gone, and so
    // are sixteen registers.  The buffer is the longest run the geometry admits (stereo 972 pieces, six channels 936, eight 1008); that
    // three workgroups still share a CU -- six channels: 52 KB each to the byte -- the table of initial values holds two copies of a
    // value instead of four (read as ds_read2_b64 of the same 8 bytes twice).
#ifdef MF_WG_NO_DMA
    static constexpr bool kDma = false;
#else
    static constexpr bool kDma = kSpan;
#endif
    static constexpr uint32_t kBiasSteps = HB ? 0 : kSteps;
    static constexpr uint32_t kBiasCopies = kDma ? 2 : 4;
    static constexpr uint32_t kBiasStep = 128 * kBiasCopies;          // [b0, b1][output 16][copies] dwords
    static constexpr uint32_t kBiasBytes = kBiasSteps * kBiasStep;
    static constexpr uint32_t kInBytes = kSpan ? kSpanBytes : (kHbRun ? kHbStage : kSR * kRowInPitch);   // the input image ...
    static constexpr uint32_t kOutBytes = (ROWS + PAIRS - 1) / PAIRS * kRowOutPitch;     // ... and the output image that lies over it (six channels: the idle pair-row's stores land behind the fifth row)
    static constexpr uint32_t kStageBytes = kDma ? kOutBytes : (kInBytes > kOutBytes ? kInBytes : kOutBytes);
    static constexpr uint32_t kDmaBytes = kDma ? (((kSR - 1) * kOutFrames * kFbIn + kRowIn + 15u) / 16u) * 16u : 0u;
    static constexpr uint32_t kLdsBytes = kPlaneBytes + kStageBytes + kBiasBytes + kDmaBytes;
    static constexpr uint32_t kGroupsPerCu = ROWS == 32 ? 2 : 3;
    static constexpr uint32_t kSubUnits = kUnitRows / kSR;            // passes per planner unit
    static constexpr uint32_t kOutPieces = kSR * (kRowOut / 16);
    static constexpr uint32_t kStoreRounds = (kOutPieces + kThreads - 1) / kThreads;
    static_assert(kGroupsPerCu * kLdsBytes <= 160 * 1024 && (!kDma || kLdsBytes <= 52 * 1024), "workgroups per CU (the LDS is handed out in granules)");
    static_assert(kTilesPerWave * kWaves == kSteps * kCt, "whole tiles per wave");
    static_assert(kSR * kRowLanes <= kThreads && (kSpan || kRowIn == (2 * kFullRounds + 1) * 8 * 16 * PAIRS) && kSubUnits * kSR == kUnitRows, "lanes per row of the input image, 4.5 (7.5) pieces each");
    // (the union of a pass's rows: at most kOutFrames input frames from one row to the next -- the block geometry src_mfma_wg_supported
    // admits -- and at least 145, so that only the fourth round has pieces past its end)
    static_assert(!kSpan || ((kSR - 1) * kOutFrames * kFbIn + kRowIn <= kSpanBytes && (kSR - 1) * 145 * kFbIn + kRowIn > (kSpanRounds - 1) * 256 * 16 && kThreads == 256), "a pass's rows in kSpanRounds rounds, only the last with pieces past the end");
    static_assert(!HB || (kChunksA * 16 * kFb <= kRoundsA * kRound && (kChunksA * 16 * kFb) >= (kRoundsA - 1) * kRound), "the two stagings meet in chunk 13");
};

#ifdef MF_WG_SPLIT_GENERIC
constexpr bool kDiagSplitGeneric = true;            // (timing experiments: stereo through the any-stride split)
#else
constexpr bool kDiagSplitGeneric = false;
#endif
#ifdef MF_WG_OUT_B16
constexpr bool kWideOut = false;                    // (A/B: round 4's epilogue, six 2-byte LDS stores per tile for every layout)
#else
constexpr bool kWideOut = true;                     // stereo and eight channels: a lane's twelve output bytes of a tile in two LDS stores
#endif
#ifdef MF_WG_NO_PIPE
constexpr bool kPipe = false;                       // (A/B: a tile's matrix instructions, then its vector instructions, tile by tile)
#else
constexpr bool kPipe = true;                        // a tile's last vector instructions between the next tile's matrix instructions
#endif
// wave priorities by phase: the tiles (C), the output's stores and the next input's staging and loads (D, A), the split (S)
#ifndef MF_WG_PRIO_C
#define MF_WG_PRIO_C 3
#endif
#ifndef MF_WG_PRIO_D
#define MF_WG_PRIO_D 0
#endif
#ifndef MF_WG_PRIO_S
#define MF_WG_PRIO_S MF_WG_PRIO_D
#endif
constexpr int kPrioC = MF_WG_PRIO_C, kPrioD = MF_WG_PRIO_D, kPrioS = MF_WG_PRIO_S;
#ifdef MF_WG_LATE_LOADS
constexpr bool kLateLoads = true;                   // (A/B: the next unit's loads issued behind the split, as in round 4)
#else
constexpr bool kLateLoads = false;
#endif
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sixteen bytes from arena offset a, bytes outside the arena read as zero.  Only units at an end of the arena come here (kWorkEdge).
__device__ __noinline__ u32x4 wg_load_piece_checked(const uint8_t* __restrict__ src, int64_t a, uint64_t arena_bytes)
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

// PLANAR (the source format): 0 = packed 24-bit frames, 4 = packed 16-bit stereo frames (SRC_LE: their byte order); 1 + k = the TInt32 planes of OHGPU_FLAG_SRC_PLANAR32 (one per
// channel, `src_plane_stride` apart, host byte order), the sample being the low 24 - 8 k bits of a plane's value shifted up k bytes
// (k = 0, 1, 2 for 24-, 16- and 8-bit streams: CodecFlac::CallbackWrite's pack, Flac.cpp:379-417, folded into the load).
template <int ROWS, int PLANAR, int PAIRS, bool HB, bool SRC_LE, bool DST_LE>
__global__ __launch_bounds__(WgGeom<ROWS>::kThreads) __attribute__((amdgpu_waves_per_eu(ROWS == 32 ? 4 : 3, ROWS == 32 ? 4 : 3)))
void src_mfma_wg_kernel(const LeanUnit* __restrict__ units, const uint32_t n_work,
                        const uint8_t* __restrict__ amat, const MfStep* __restrict__ steps,
                        const uint16_t* __restrict__ planes, const uint32_t plane_stride,
                        const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                        const uint32_t row_src_bytes, const uint64_t src_arena_bytes)
{
    using G = WgGeom<ROWS, PLANAR, PAIRS, HB>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const pl_lds = smem;
    uint8_t* const stage = smem + G::kPlaneBytes;
    uint8_t* const bias_lds = stage + G::kStageBytes;
    uint8_t* const dma_lds = bias_lds + G::kBiasBytes;                // (kDma: the pass's input run, piece for piece)
    uint8_t* const in_img = G::kDma ? dma_lds : stage;                // where the split finds a packed pass's run
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // this wave's tiles: tile0 .. tile0 + 4 of the unit's (step, column tile) pairs; its first step, and the A operands of the steps it touches
    const uint32_t tile0 = G::kTilesPerWave * wave;
    const uint32_t step0 = tile0 / G::kCt;

    // the accumulators' initial values (MfStep::b0..b2) of the block's steps
    for (uint32_t i = tid; i < G::kBiasSteps * 32u; i += G::kThreads) {
        const uint32_t t = i / 32u, r = i - 32u * t;
        const uint32_t v = steps[t].b0[r];                 // (b0 and b1 lie one after the other)
        if constexpr (G::kDma) ((u32x2*)bias_lds)[i] = u32x2{v, v};
        else ((u32x4*)bias_lds)[i] = u32x4{v, v, v, v};
    }
    // this wave's A operands, for good
    v4i a[G::kASets][4];
    uint32_t kc[G::kKcSets];
#pragma unroll
    for (int q = 0; q < (int)G::kKcSets; q++) {
        const uint32_t st = step0 + q < G::kSteps ? step0 + q : G::kSteps - 1u;   // (a set past the last step is never used)
        if (q < (int)G::kASets) {
            const uint32_t image = HB ? 0u : st;                                   // (half-band: one image for every step)
#pragma unroll
            for (int j = 0; j < 4; j++) a[q < (int)G::kASets ? q : 0][j] = *(const v4i*)(amat + ((uint64_t)image * kMfStepImage + j * 1024u + lane * 16u));
        }
        kc[q] = steps[st].kc;
    }

    // ---- lane roles ----
    // matrix operands: the SAMPLES are the A operand (lane = column n of the tile, K group g), the coefficients the B operand (lane =
    // output n of the step, K group g); the result's lane (g, n) then holds output frame n of the four columns 4 g .. 4 g + 3 =
    // rows 2 g and 2 g + 1 of the tile, both channels: a whole frame of each in one lane, nothing to exchange
    const uint32_t g = lane >> 4, n = lane & 15;
    // (half-band: K groups 0..2 are the even-frame chunks kc .. kc + 2, group 3 the odd-frame chunk kc + 1 = plane chunk 10 + kc + 1)
    const uint32_t g_chunk = HB ? (g < 3u ? g : 11u) : g;
    const uint8_t* const b_lds = pl_lds + g_chunk * G::kChunk + n * 8u;           // + kc * chunk + digit * kDigit + ((half * kHalf + tile * 128) ^ parity of the chunk * 128)
    const uint8_t* const my_bias = bias_lds + 4u * G::kBiasCopies * n;            // + step * kBiasStep: b0; b1 half a step further
    // (half-band: the two values, once)
    const int hb0 = HB ? (int)steps[0].b0[n] : 0, hb1 = HB ? (int)steps[0].b1[n] : 0;
    const v4i hb_s0 = v4i{hb0, hb0, hb0, hb0}, hb_s2 = v4i{hb1, hb1, hb1, hb1};
    // where the lane's frame of pair-row 2 g lies in the output image (+ 16 frames per step; + 8 pair-rows per column tile; the pair-row
    // 2 g + 1 is the next row (stereo) or the next pair of the same row (eight channels); six channels: out_at below)
    uint8_t* const out_lds = stage + ((2u * g) / (uint32_t)PAIRS) * G::kRowOutPitch + 6u * ((2u * g) % (uint32_t)PAIRS) + G::kFb * n;
    auto pair_row_srow = [&](uint32_t ct, uint32_t q) __attribute__((always_inline)) { return (ct * 8u + 2u * g + q) / (uint32_t)PAIRS; };   // the stream row of a tile's pair-row
    // the input image row by row (planar sources and the half-band form; packed sources otherwise: the pass's one run, span_* below):
    // 16 PAIRS lanes per row; lane `sub` of a row moves its pieces sub, sub + 16 PAIRS, .. and, the first half of them, one more --
    // an instruction reads 256 PAIRS contiguous bytes of every row, and a lane's addresses differ by constants (planar: a channel's
    // 768 bytes are three rounds exactly; the second channel's come from `src_plane_stride` further on; six channels: 240 lanes
    // move the five rows, the last sixteen repeat pieces of the fifth)
    const uint32_t in_row_of = tid / G::kRowLanes, in_sub = tid - in_row_of * G::kRowLanes;
    const uint32_t in_row = in_row_of < G::kSR ? in_row_of : G::kSR - 1u;
    const uint32_t in_src = in_row * row_src_bytes + 16u * in_sub;                 // + kRound k
    const uint32_t in_lds = in_row * G::kRowInPitch + 16u * in_sub;
    // (packed: the half round's spare lanes repeat an earlier piece of theirs -- the first; half-band: the fifth, which is staged with it)
    const uint32_t in_last = in_sub < G::kRowLanes / 2u ? G::kFullRounds * G::kRound : (HB ? (G::kRoundsA - 1u) * G::kRound : 0u);
    // the split: task q = threads * k + tid (k = 0, 1; q < 24 ROWS) is half chunk q / ROWS of pair-row q % ROWS (the pair-rows side by side)
    const uint32_t sp_row = tid % (uint32_t)ROWS, sp_hc0 = tid / (uint32_t)ROWS;   // (0..15; second round: half chunk + 16 while < 24)
    const uint32_t sp_used = sp_row < G::kSR * (uint32_t)PAIRS ? sp_row : G::kSR * (uint32_t)PAIRS - 1u;    // (six channels: the idle pair-row repeats the fifteenth)
    const uint32_t sp_srow = sp_used / (uint32_t)PAIRS, sp_pair = sp_used % (uint32_t)PAIRS;
    const uint32_t sp_span0 = sp_srow * row_src_bytes + 6u * sp_pair;               // (kSpan: the pair-row's first frame in the pass's run)
    // kSpan: the run's pieces -- the rows' union, rounded up to whole pieces (src_mfma_wg_unit_inside leaves the 15 bytes that can add)
    const uint32_t span_pieces = ((G::kSR - 1u) * row_src_bytes + G::kRowIn + 15u) >> 4;
    constexpr uint32_t kSpanLast = (G::kSpan ? G::kSpanRounds - 1u : 0u) * 256u;
    const uint32_t span_p3 = kSpanLast + tid < span_pieces ? kSpanLast + tid : tid;  // (the last round's lanes past the end repeat their first piece)
    constexpr uint32_t kHbLast = (G::kHbRun ? G::kHbRunRounds - 1u : 0u) * 256u;
    const uint32_t hb_p_last = kHbLast + tid < G::kHbRunPieces ? kHbLast + tid : tid;  // (kHbRun: the same for the half-band run)

    // pack: a frame's six bytes from its two 24-bit values, L then R, each most significant byte first (big endian) or last
    constexpr uint32_t kB0 = DST_LE ? 0 : 2, kB1 = 1, kB2 = DST_LE ? 2 : 0;       // byte of the 24-bit value that is memory byte 0, 1, 2
    constexpr uint32_t sel_lo = kB0 | kB1 << 8 | kB2 << 16 | (4 + kB0) << 24;     // {R, L} -> L's three bytes, R's first
    constexpr uint32_t sel_hi = (4 + kB1) | (4 + kB2) << 8 | 0x0c0c0000u;          // {R, L} -> R's other two
    // ... and twelve bytes from FOUR values {a, b, c, d} (kWideOut): dword 0 = sel_lo of {b, a}, dword 1 = b's other two bytes and c's
    // first two, dword 2 = c's last and d's three
    constexpr uint32_t sel_mid = kB1 | kB2 << 8 | (4 + kB0) << 16 | (4 + kB1) << 24;
    constexpr uint32_t sel_top = kB2 | (4 + kB0) << 8 | (4 + kB1) << 16 | (4 + kB2) << 24;
    const bool n_odd = (n & 1u) != 0;
    // (kWideOut, stereo: the lane's twelve bytes -- the even lane's in row 2 g from its own frame on, the odd lane's in row 2 g + 1 from its even neighbour's frame on)
    const uint32_t wide_off = n_odd ? G::kRowOutPitch - G::kFb : 0u;

    // a workgroup unit = sub-unit `u % kSubUnits` of planner unit `u / kSubUnits`
    struct Unit { int64_t src0, dst0; uint32_t n_blocks, plane, plane_stride; bool ramped, first, edge; };
    auto fetch_unit = [&](uint32_t u) __attribute__((always_inline)) {
        const LeanUnit w = units[u / G::kSubUnits];
        const uint32_t sub = u % G::kSubUnits, r0 = sub * G::kSR;
        Unit o;
        o.src0 = w.src_row0 + (int64_t)(r0 * row_src_bytes);
        o.dst0 = w.dst_row0 + (int64_t)(r0 * G::kRowOut);
        o.n_blocks = w.n_blocks > r0 ? (w.n_blocks - r0 < G::kSR ? w.n_blocks - r0 : G::kSR) : 0u;
        o.plane = w.plane + r0 * (G::kOutFrames / 8u);     // (a plane row is 160 (128) entries of 2 bytes = 20 (16) of the plane stride's 16)
        o.plane_stride = w.src_plane_stride;
        o.ramped = (w.flags & kWorkRamped) != 0;
        o.first = (w.flags & kWorkFirst) != 0 && sub == 0;
        o.edge = (w.flags & kWorkEdge) != 0;
        return o;
    };
    auto issue_input = [&](const Unit& w, u32x4 (&raw)[G::kInRounds]) __attribute__((always_inline)) {
        // (scalar base + 32-bit lane offset is the load's scalar-base form, but only if the offset is widened in THIS block: hoisted out
        // of the loop as a 64-bit pair it costs eight registers for the whole launch and a 64-bit add per load -- mf_here pins it)
        const uint8_t* const base = src + w.src0;
#ifdef MF_DIAG_NO_LOAD
        (void)base;
#pragma unroll
        for (int k = 0; k < (int)G::kInRounds; k++) raw[k] = u32x4{tid, in_src, (uint32_t)w.n_blocks, (uint32_t)k};
#else
        if (w.edge) {
            // a unit at an end of the arena (the first of the first stream, the last of the last): its pieces one by one, out of line,
            // bytes outside the arena read as zero -- they are history before a stream's first frame, rows the unit does not hold, or
            // the slack behind a row's last frame
#pragma unroll
            for (int k = 0; k < (int)G::kInRounds; k++) {
                const int64_t at = G::kPlanes ? (int64_t)(in_src + 256u * (uint32_t)(k % 3)) + (int64_t)(k / 3) * w.plane_stride
                                 : G::kHbRun ? (int64_t)(16u * (k + 1 < (int)G::kInRounds ? tid + 256u * (uint32_t)k : hb_p_last))
                                 : G::kSpan ? (int64_t)(16u * (k + 1 < (int)G::kInRounds ? tid + 256u * (uint32_t)k : span_p3))
                                          : (int64_t)(k < (int)G::kFullRounds ? in_src + G::kRound * (uint32_t)k : in_src + in_last);
                raw[k] = wg_load_piece_checked(src, w.src0 + at, src_arena_bytes);
            }
        } else if constexpr (G::kPlanes) {
            const uint32_t o = mf_here(in_src);
            const uint8_t* const base1 = base + w.plane_stride;
#pragma unroll
            for (int k = 0; k < 3; k++) raw[k] = *(const u32x4_u*)(base + o + 256 * k);
#pragma unroll
            for (int k = 0; k < 3; k++) raw[3 + k] = *(const u32x4_u*)(base1 + o + 256 * k);
        } else {
            if constexpr (G::kSpan) {
                const uint32_t o = mf_here(16u * tid);
                // (stereo: non-temporal -- a piece is read once, but for the 32 frames of history two passes share, 1.4 % of a pass's 2352,
                // and need not stay in a cache: -0.7 % on the headline.  Six and eight channels, whose passes are five and four rows, share
                // 4-5 %: there the plain load is the better one, by as much)
                if constexpr (PAIRS == 1) {
#pragma unroll
                    for (int k = 0; k + 1 < (int)G::kInRounds; k++) raw[k] = __builtin_nontemporal_load((const u32x4_u*)(base + o + 4096 * k));
                    raw[G::kInRounds - 1] = __builtin_nontemporal_load((const u32x4_u*)(base + mf_here(16u * span_p3)));
                } else {
#pragma unroll
                    for (int k = 0; k + 1 < (int)G::kInRounds; k++) raw[k] = *(const u32x4_u*)(base + o + 4096 * k);
                    raw[G::kInRounds - 1] = *(const u32x4_u*)(base + mf_here(16u * span_p3));
                }
            } else if constexpr (G::kHbRun) {
                // (the rows' union, piece for piece; the last round's lanes past its end repeat their first piece)
                const uint32_t o = mf_here(16u * tid);
#pragma unroll
                for (int k = 0; k + 1 < (int)G::kInRounds; k++) raw[k] = *(const u32x4_u*)(base + o + 4096 * k);
                raw[G::kInRounds - 1] = *(const u32x4_u*)(base + mf_here(16u * hb_p_last));
            } else {
                const uint32_t o = mf_here(in_src);
#pragma unroll
                for (int k = 0; k < (int)G::kFullRounds; k++) raw[k] = *(const u32x4_u*)(base + o + (int)G::kRound * k);
                raw[G::kFullRounds] = *(const u32x4_u*)(base + mf_here(in_src + in_last));
            }
        }
#endif
    };
    // kDma: the pass's run, memory -> LDS.  A wave's lanes fill a contiguous KB per instruction (LDS address = M0 + 16 lane); lanes past
    // the run's end sit out.  A unit at an end of the arena brings its pieces through registers, checked, as ever.
    auto issue_dma = [&](const Unit& w) __attribute__((always_inline)) {
#ifndef MF_DIAG_NO_LOAD
        if (w.edge) {
#pragma unroll
            for (int k = 0; k < (int)G::kInRounds; k++) {
                const uint32_t p = k + 1 < (int)G::kInRounds ? tid + 256u * (uint32_t)k : span_p3;
                *(u32x4*)(dma_lds + 16u * p) = wg_load_piece_checked(src, w.src0 + (int64_t)(16u * p), src_arena_bytes);
            }
        } else {
            const uint8_t* const base = src + w.src0;
            const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u8_t)dma_lds + 1024u * wave;
#pragma unroll
            for (int k = 0; k < (int)G::kInRounds; k++) {
                const uint32_t p = tid + 256u * (uint32_t)k;
                if (k + 1 < (int)G::kInRounds || p < span_pieces) {
                    // (stereo: non-temporal, six and eight channels plain -- issue_input has why)
                    if constexpr (PAIRS == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(16u * p), "s"(base), "s"(lds0 + 4096u * (uint32_t)k) : "memory", "m0");
                    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(16u * p), "s"(base), "s"(lds0 + 4096u * (uint32_t)k) : "memory", "m0");
                }
            }
        }
#endif
    };
    auto stage_input = [&](const u32x4 (&raw)[G::kInRounds]) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_STAGE
        return;
#endif
        if constexpr (G::kPlanes) {
#pragma unroll
            for (int k = 0; k < 6; k++) *(u32x4*)(stage + in_lds + (k / 3) * kWgPlaneIn + 256 * (k % 3)) = raw[k];
        } else if constexpr (G::kSpan) {
#pragma unroll
            for (int k = 0; k < (int)G::kInRounds; k++) *(u32x4*)(stage + 16u * tid + 4096 * k) = raw[k];      // (piece for piece: a wave writes a contiguous KB)
        }
    };
    // half-band: the image in two sections over the same LDS -- section 0 = the lanes' rounds 0..4 (bytes 0 .. 5 kRound of every row:
    // input chunks 0..12 whole), section 1 = rounds 4..7 (bytes 4 kRound .. 7.5 kRound, stored from the row's start: chunks 13..19)
    auto stage_section = [&](auto sec_c, const u32x4 (&raw)[G::kInRounds]) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_STAGE
        return;
#endif
        constexpr int SEC = decltype(sec_c)::value;
        if constexpr (G::kHbRun) {
            // section SEC of the run: its pieces to where the section starts at 0 (a round wholly outside it is no code at all)
            constexpr uint32_t kLo = SEC == 0 ? 0u : G::kHbSec1Lo, kHi = SEC == 0 ? G::kHbSec0Hi : G::kHbRunBytes;
#pragma unroll
            for (int k = 0; k < (int)G::kInRounds; k++) {
                if (4096u * (uint32_t)k + 4096u <= kLo || 4096u * (uint32_t)k >= kHi) continue;
                const uint32_t p = k + 1 < (int)G::kInRounds ? tid + 256u * (uint32_t)k : hb_p_last;
                if (16u * p >= kLo && 16u * p < kHi) *(u32x4*)(stage + (16u * p - kLo) + 16u * ((16u * p) / G::kHbAdv - kLo / G::kHbAdv)) = raw[k];
            }
        } else if constexpr (SEC == 0) {
#pragma unroll
            for (int k = 0; k < (int)G::kRoundsA; k++) *(u32x4*)(stage + in_lds + (int)G::kRound * k) = raw[k];
        } else {
#pragma unroll
            for (int k = (int)G::kRoundsA - 1; k < (int)G::kFullRounds; k++) *(u32x4*)(stage + in_lds + (int)G::kRound * (k - ((int)G::kRoundsA - 1))) = raw[k];
            *(u32x4*)(stage + in_lds + (in_last - (G::kRoundsA - 1u) * G::kRound)) = raw[G::kFullRounds];
        }
    };
    // ... and its split: task t of a section = (pair-row t % ROWS, input chunk and parity t / ROWS): the chunk's eight EVEN frames (or its
    // eight odd ones) of the lane's channel pair -> half `chunk & 1` of plane chunk `chunk / 2` (+ 10 for the odd frames)
    auto split_hb_task = [&](uint32_t t, uint32_t first_chunk, uint32_t image_byte0, bool first) __attribute__((always_inline)) {
        const uint32_t t2 = t / (uint32_t)ROWS, ic = first_chunk + (t2 >> 1), par = t2 & 1u;
        const bool zero = first && sp_srow == 0 && ic < 4u;         // the stream's block 0: the 64 frames before it read as zeros
        const uint32_t byte0 = sp_srow * G::kRowInPitch + (ic * (16u * G::kFb) - image_byte0) + par * G::kFb + 6u * sp_pair;
        const uint32_t at = byte0 & ~3u, sh = (byte0 & 2u) * 8u;  // (frames 2 kFb apart: all of them as far off a dword as the first)
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const u32x2 e = *(const u32x2_a4*)(stage + at + 2 * (int)G::kFb * m);
            const uint64_t v = (((uint64_t)e.y << 32) | e.x) >> sh;
            lo[m] = zero ? 0u : (uint32_t)v; hi[m] = zero ? 0u : (uint32_t)(v >> 32);
        }
        uint32_t pl[6][2];
        mf_split_frames(lo, hi, pl);
        const uint32_t c = par * 10u + (ic >> 1);
        uint8_t* const to = pl_lds + c * G::kChunk + (((ic & 1u) * G::kHalf + sp_row * 16u) ^ ((c & 1u) * 128u));     // + digit * kDigit
#pragma unroll
        for (int bpos = 0; bpos < 3; bpos++) {
            const int digit = SRC_LE ? bpos : 2 - bpos;
            const uint32_t flip = digit < 2 ? 0x80808080u : 0u;
            *(u32x4*)(to + digit * G::kDigit) = u32x4{pl[bpos][0] ^ flip, pl[bpos][1] ^ flip, pl[3 + bpos][0] ^ flip, pl[3 + bpos][1] ^ flip};
        }
    };
    // kHbRun: task t of a section = (pair-row t % its pair-rows, input chunk and parity t / them) over ALL twenty chunks of the rows the
    // section holds; the row's image starts at (row) x 256 frames of the run, less where the section starts
    auto split_hb_run_task = [&](auto sec_c, uint32_t t, bool first) __attribute__((always_inline)) {
        constexpr int SEC = decltype(sec_c)::value;
        constexpr uint32_t kRows = SEC == 0 ? G::kHbRows0 : G::kSR - G::kHbRows0, kPr = kRows * (uint32_t)PAIRS;
        constexpr uint32_t kPr0 = SEC == 0 ? 0u : G::kHbRows0 * (uint32_t)PAIRS, kLo = SEC == 0 ? 0u : G::kHbSec1Lo;
        const uint32_t t2 = t / kPr, pr = kPr0 + (t - t2 * kPr), ic = t2 >> 1, par = t2 & 1u;
        const uint32_t srow = pr / (uint32_t)PAIRS, pair = pr - srow * (uint32_t)PAIRS;
        const bool zero = first && srow == 0 && ic < 4u;            // the stream's block 0: the 64 frames before it read as zeros
        // (the row's own 16 spare bytes per 256 frames in front of it, and one more set behind its sixteenth chunk)
        const uint32_t byte0 = srow * G::kHbAdv - kLo + 16u * (srow - kLo / G::kHbAdv) + (ic >= 16u ? 16u : 0u) + ic * (16u * G::kFb) + par * G::kFb + 6u * pair;
        const uint32_t at = byte0 & ~3u, sh = (byte0 & 2u) * 8u;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const u32x2 e = *(const u32x2_a4*)(stage + at + 2 * (int)G::kFb * m);
            const uint64_t v = (((uint64_t)e.y << 32) | e.x) >> sh;
            lo[m] = zero ? 0u : (uint32_t)v; hi[m] = zero ? 0u : (uint32_t)(v >> 32);
        }
        uint32_t pl[6][2];
        mf_split_frames(lo, hi, pl);
        const uint32_t c = par * 10u + (ic >> 1);
        uint8_t* const to = pl_lds + c * G::kChunk + (((ic & 1u) * G::kHalf + pr * 16u) ^ ((c & 1u) * 128u));     // + digit * kDigit
#pragma unroll
        for (int bpos = 0; bpos < 3; bpos++) {
            const int digit = SRC_LE ? bpos : 2 - bpos;
            const uint32_t flip = digit < 2 ? 0x80808080u : 0u;
            *(u32x4*)(to + digit * G::kDigit) = u32x4{pl[bpos][0] ^ flip, pl[bpos][1] ^ flip, pl[3 + bpos][0] ^ flip, pl[3 + bpos][1] ^ flip};
        }
    };
    auto split_section = [&](auto sec_c, bool first) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_SPLIT
        return;
#endif
        constexpr int SEC = decltype(sec_c)::value;
        if constexpr (G::kHbRun) {
            constexpr uint32_t kTasksRun = (SEC == 0 ? G::kHbRows0 : G::kSR - G::kHbRows0) * (uint32_t)PAIRS * 40u;
#pragma unroll
            for (uint32_t k = 0; k * G::kThreads < kTasksRun; k++)
                if (k * G::kThreads + tid < kTasksRun) split_hb_run_task(sec_c, k * G::kThreads + tid, first);
            return;
        }
        constexpr uint32_t kFirstChunk = SEC == 0 ? 0u : G::kChunksA, kChunks = SEC == 0 ? G::kChunksA : G::kImgChunks - G::kChunksA;
        constexpr uint32_t kTasks = kChunks * 2u * (uint32_t)ROWS, kByte0 = SEC == 0 ? 0u : (G::kRoundsA - 1u) * G::kRound;
#pragma unroll
        for (uint32_t k = 0; k * G::kThreads < kTasks; k++)
            if (k * G::kThreads + tid < kTasks) split_hb_task(k * G::kThreads + tid, kFirstChunk, kByte0, first);
    };
    auto split_task = [&](uint32_t hc, bool first) __attribute__((always_inline)) {
        // the stream's block 0: the frames before it (chunks 0 and 1 of row 0) read as zeros
        const bool zero = first && sp_srow == 0 && hc < 4u;
        uint32_t pl[6][2];                                  // [3 * channel + byte of the 24-bit sample, least significant first... in memory order for packed][frames 0-3, 4-7]
        if constexpr (G::kS16) {
            // 16-bit stereo: a frame is one dword {L lo, L hi, R lo, R hi} (or hi first); a 4 x 4 byte transpose per four frames gives the
            // four byte planes, and the 24-bit sample is the 16 bits over a zero byte -- digit 0 is the offset digit of zero everywhere
            const uint8_t* const from = in_img + sp_span0 + 32u * hc;
            const u32x4 lo4 = *(const u32x4_a4*)from, hi4 = *(const u32x4_a4*)(from + 16);
            const uint32_t f[8] = {zero ? 0u : lo4.x, zero ? 0u : lo4.y, zero ? 0u : lo4.z, zero ? 0u : lo4.w,
                                   zero ? 0u : hi4.x, zero ? 0u : hi4.y, zero ? 0u : hi4.z, zero ? 0u : hi4.w};
            uint32_t by[4][2];                               // [byte of the frame][frames 0-3, 4-7]
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t p01 = mf_perm(f[4 * q + 1], f[4 * q], 0x05010400u), q01 = mf_perm(f[4 * q + 1], f[4 * q], 0x07030602u);
                const uint32_t p23 = mf_perm(f[4 * q + 3], f[4 * q + 2], 0x05010400u), q23 = mf_perm(f[4 * q + 3], f[4 * q + 2], 0x07030602u);
                by[0][q] = mf_perm(p23, p01, 0x05040100u); by[1][q] = mf_perm(p23, p01, 0x07060302u);
                by[2][q] = mf_perm(q23, q01, 0x05040100u); by[3][q] = mf_perm(q23, q01, 0x07060302u);
            }
            // pl[3 * channel + byte of the 24-bit sample, least significant first] (stored below as if the source were little endian)
            constexpr int kLo = SRC_LE ? 0 : 1, kHi = SRC_LE ? 1 : 0;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                pl[0][q] = 0u; pl[1][q] = by[kLo][q]; pl[2][q] = by[kHi][q];
                pl[3][q] = 0u; pl[4][q] = by[2 + kLo][q]; pl[5][q] = by[2 + kHi][q];
            }
        } else if constexpr (G::kPlanes) {
            // eight frames of each channel, 4 bytes apiece: a 4 x 4 byte transpose per four frames (two permute levels) of which the
            // sample's bytes are kept -- byte b of the plane value is byte b + k of the 24-bit sample
            constexpr int kShift = G::kPlanes ? PLANAR - 1 : 0;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const uint8_t* const from = stage + sp_row * G::kRowInPitch + c * kWgPlaneIn + 32u * hc;
                const u32x4 lo4 = *(const u32x4*)from, hi4 = *(const u32x4*)(from + 16);
                const uint32_t f[8] = {zero ? 0u : lo4.x, zero ? 0u : lo4.y, zero ? 0u : lo4.z, zero ? 0u : lo4.w,
                                       zero ? 0u : hi4.x, zero ? 0u : hi4.y, zero ? 0u : hi4.z, zero ? 0u : hi4.w};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const uint32_t p01 = mf_perm(f[4 * q + 1], f[4 * q], 0x05010400u), q01 = mf_perm(f[4 * q + 1], f[4 * q], 0x07030602u);   // {A0 B0 A1 B1}, {A2 B2 A3 B3}
                    const uint32_t p23 = mf_perm(f[4 * q + 3], f[4 * q + 2], 0x05010400u), q23 = mf_perm(f[4 * q + 3], f[4 * q + 2], 0x07030602u);
                    const uint32_t b0 = mf_perm(p23, p01, 0x05040100u), b1 = mf_perm(p23, p01, 0x07060302u), b2 = mf_perm(q23, q01, 0x05040100u);
                    // sample byte j (least significant first) = plane byte j - kShift, zero below
                    const uint32_t sb[3] = {kShift == 0 ? b0 : 0u, kShift == 0 ? b1 : (kShift == 1 ? b0 : 0u), kShift == 0 ? b2 : (kShift == 1 ? b1 : b0)};
#pragma unroll
                    for (int j = 0; j < 3; j++) pl[3 * c + j][q] = sb[j];
                }
            }
        } else if constexpr (PAIRS == 1 && !kDiagSplitGeneric) {
            // stereo: the eight frames are 48 contiguous bytes from an even address -- the thirteen dwords around them, moved down by the
            // two bytes an odd start is off (one v_alignbyte_b32 per dword), then the same network as for twelve aligned dwords
            const uint32_t byte0 = sp_span0 + hc * 48u;
            const uint32_t at = byte0 & ~3u, sh = byte0 & 2u;
            uint32_t d[13];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const u32x4 q = *(const u32x4_a4*)(in_img + at + 16 * k);
                d[4 * k] = q.x; d[4 * k + 1] = q.y; d[4 * k + 2] = q.z; d[4 * k + 3] = q.w;
            }
            d[12] = *(const uint32_t*)(in_img + at + 48);
            uint32_t w[12];
#pragma unroll
            for (int k = 0; k < 12; k++) w[k] = zero ? 0u : __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
            mf_split48(w, pl);
        } else {
            // eight frames of the lane's channel pair, kFb bytes apart: six bytes each, wherever they start (a row starts anywhere even in
            // the pass's run) -- the 8 aligned bytes around them (two dwords in one LDS read), shifted down by the two bytes an odd start
            // is off.  Frames at even and at odd distances differ in that for two and six channels (6 n and 18 n are a multiple of 4 or 2
            // off by turns), not for eight.
            const uint32_t byte0 = sp_span0 + hc * (8u * G::kFb);
            const uint32_t at_e = byte0 & ~3u, sh_e = (byte0 & 2u) * 8u;
            const uint32_t at_o = (byte0 + G::kFb) & ~3u, sh_o = ((byte0 + G::kFb) & 2u) * 8u;
            uint32_t lo[8], hi[8];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const u32x2 e = *(const u32x2_a4*)(in_img + at_e + 2 * (int)G::kFb * m), o = *(const u32x2_a4*)(in_img + at_o + 2 * (int)G::kFb * m);
                const uint64_t ve = (((uint64_t)e.y << 32) | e.x) >> sh_e, vo = (((uint64_t)o.y << 32) | o.x) >> sh_o;
                lo[2 * m] = zero ? 0u : (uint32_t)ve; hi[2 * m] = zero ? 0u : (uint32_t)(ve >> 32);
                lo[2 * m + 1] = zero ? 0u : (uint32_t)vo; hi[2 * m + 1] = zero ? 0u : (uint32_t)(vo >> 32);
            }
            mf_split_frames(lo, hi, pl);
        }
        // A plane's chunk is [half][row][channel 2][8 frames]: a lane's two channels are 16 contiguous bytes, the rows of a task a
        // contiguous run, a wave's tasks whole halves -- one conflict-free 16-byte store per digit (the 8-byte stores of a
        // [row][channel][16 frames] layout met the banks four deep).  Odd chunks are stored with bit 7 of the offset flipped, so that the
        // 8-byte reads of chunks kc + g and kc + g + 1, which share a 32-lane half of the A operand's load, fall into different banks.
        const uint32_t c = hc >> 1;
        uint8_t* const to = pl_lds + c * G::kChunk + (((hc & 1u) * G::kHalf + sp_row * 16u) ^ ((c & 1u) * 128u));     // + digit * kDigit
#pragma unroll
        for (int bpos = 0; bpos < 3; bpos++) {
            const int digit = (SRC_LE || PLANAR) ? bpos : 2 - bpos;     // (planes and 16-bit frames arrive least significant byte first)
            const uint32_t flip = digit < 2 ? 0x80808080u : 0u;
            *(u32x4*)(to + digit * G::kDigit) = u32x4{pl[bpos][0] ^ flip, pl[bpos][1] ^ flip, pl[3 + bpos][0] ^ flip, pl[3 + bpos][1] ^ flip};
        }
    };
    auto split_all = [&](bool first) __attribute__((always_inline)) {
#ifdef MF_DIAG_NO_SPLIT
        return;
#endif
        split_task(sp_hc0, first);
        if (sp_hc0 < 8u) split_task(sp_hc0 + 16u, first);
    };
    // a pass's input, from the registers it arrived in to the planes
    auto stage_and_split = [&](const u32x4 (&raw)[G::kInRounds], bool first) __attribute__((always_inline)) {
        if constexpr (HB) {
            stage_section(std::integral_constant<int, 0>{}, raw);
            wg_barrier<2>();
            split_section(std::integral_constant<int, 0>{}, first);
            wg_barrier<2>();                                 // section 0 has been read
            stage_section(std::integral_constant<int, 1>{}, raw);
            wg_barrier<2>();
            split_section(std::integral_constant<int, 1>{}, first);
        } else {
            stage_input(raw);
            wg_barrier<2>();
            split_all(first);
        }
    };

    // Units are dealt round robin: they cost the same (a ramped one a few instructions per tile more), so a workgroup's share
    // is even to within one unit in a hundred, and a unit index that is a launch constant plus a counter stays in scalar registers --
    // a claimed one would come back in a vector register, and with it every address of the unit.
    const uint32_t n_groups = gridDim.x;
    uint32_t u_cur = blockIdx.x;                           // the unit in the planes
    if (u_cur >= n_work) return;                            // (uniform; the launch keeps the grid within the units)
    uint32_t u_nxt = u_cur + n_groups;                     // the unit whose input is in flight
    Unit wk = fetch_unit(u_cur);
    u32x4 raw[G::kInRounds];
    Unit wk_nxt = wk;
    if constexpr (G::kDma) {
        issue_dma(wk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // (the bias table; the run)
        split_all(wk.first);
        wk_nxt = fetch_unit(u_nxt < n_work ? u_nxt : u_cur);
        wg_barrier();
        if (u_nxt < n_work) issue_dma(wk_nxt);              // (uniform)
    } else {
    issue_input(wk, raw);
    __syncthreads();                                        // (the bias table)
    stage_and_split(raw, wk.first);
    wk_nxt = fetch_unit(u_nxt < n_work ? u_nxt : u_cur);
    issue_input(wk_nxt, raw);                               // (past the last unit: the current one again, never used)
    wg_barrier();
    }

    while (true) {
        // ---- (C) this wave's five tiles of the unit ----
        const uint32_t n_blocks = wk.n_blocks;
        const bool ramped = wk.ramped;
        const uint8_t* const mbase = (const uint8_t*)planes + (uint64_t)wk.plane * plane_stride;
        // The five tiles as straight-line code, one copy per (pattern of the wave's tiles, ramped or not): which of the wave's A sets a
        // tile takes and which column tile it is are constants of the wave's position among the kCt waves that share five steps, so
        // nothing branches between two tiles and a tile's operands are fetched under its predecessor's matrix instructions.
        // A tile's sample operands: six 8-byte reads, written out (the compiler pairs the halves into ds_read2st64_b64, which the LDS
        // serves at half the rate).  They are ISSUED under the previous tile's matrix instructions and WAITED FOR in front of this
        // tile's, a whole epilogue later: in between the registers belong to the reads in flight -- they are outputs of the first
        // statement and in/outs of the second and of nothing else, so the compiler keeps them allocated and has no reason to name
        // them (tests/test_mfma_kernel_asm.py walks the LDS queue of every instantiation to see that it did not).
        auto issue_planes = [&](uint32_t kcs, uint32_t ct, u32x2 (&h)[6]) __attribute__((always_inline)) {
            const uint32_t in_chunk = (ct * 128u) ^ (((kcs + g_chunk) & 1u) * 128u);
            const uint32_t at = (uint32_t)(uintptr_t)(lds_u8_t)(b_lds + kcs * G::kChunk + in_chunk);
            asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:%7\n\t"
                         "ds_read_b64 %2, %6 offset:%8\n\tds_read_b64 %3, %6 offset:%9\n\t"
                         "ds_read_b64 %4, %6 offset:%10\n\tds_read_b64 %5, %6 offset:%11"
#ifdef MF_WG_EARLY_WAIT
                         "\n\ts_waitcnt lgkmcnt(0)"
#endif
                         : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5])
                         : "v"(at), "n"(G::kHalf), "n"(G::kDigit), "n"(G::kDigit + G::kHalf), "n"(2 * G::kDigit), "n"(2 * G::kDigit + G::kHalf)
                         : "memory");
        };
        // kDma: an output's two initial values from the table of two copies -- the same eight bytes twice fill the four registers of a
        // C operand (ds_read2_b64 with both offsets alike).  Written out, so the wait is the caller's: take_planes' comes next.
        auto read_bias2 = [&](const uint8_t* bi, v4i& s0, v4i& s2) __attribute__((always_inline)) {
            const uint32_t at = (uint32_t)(uintptr_t)(lds_u8_t)bi;
            asm volatile("ds_read2_b64 %0, %2\n\tds_read2_b64 %1, %2 offset0:%3 offset1:%3"
                         : "=&v"(s0), "=&v"(s2) : "v"(at), "n"(G::kBiasStep / 16) : "memory");
        };
        auto take_planes = [&](u32x2 (&h)[6], v4i (&bd)[3]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]) : : "memory");
#pragma unroll
            for (int d = 0; d < 3; d++) bd[d] = v4i{(int)h[2 * d].x, (int)h[2 * d].y, (int)h[2 * d + 1].x, (int)h[2 * d + 1].y};
        };
        auto run_tiles = [&](auto pattern_c, auto ramped_c) __attribute__((always_inline)) {
            constexpr uint32_t P = decltype(pattern_c)::value;
            constexpr bool RAMPED = decltype(ramped_c)::value;
            constexpr uint32_t kFirst = G::kTilesPerWave * P;           // the pattern's first tile, counted from a multiple of kTilesPerWave steps
            // a tile's results from its four outputs y: the ramp, the pack, the stores into the output image
            auto finish_tile = [&](auto step_set_c, auto ct_c, int (&y)[4]) __attribute__((always_inline)) {
                constexpr uint32_t ct = decltype(ct_c)::value;
                const uint32_t step = step0 + (uint32_t)decltype(step_set_c)::value;
                if constexpr (RAMPED) {
                    // RampApplicator::GetNextSample on the 24-bit value (Msg.cpp:840-895): top 16 bits * Q15 >> 15, low byte zero; the
                    // multiplier of the lane's frame in each of its two rows comes from the unit's plane (0xffff: the frame's message has no ramp)
                    const uint32_t row0 = pair_row_srow(ct, 0), row1 = pair_row_srow(ct, 1);
                    const uint32_t e0 = (row0 < n_blocks ? row0 * G::kOutFrames : 0u) + 16u * step + n;
                    const uint32_t e1 = (row1 < n_blocks ? row1 * G::kOutFrames : 0u) + 16u * step + n;
                    uint32_t mu[2];
                    mu[0] = *(const uint16_t*)(mbase + mf_here(2u * e0));
                    if constexpr (PAIRS == 4) mu[1] = mu[0];         // (pair-rows 2 g and 2 g + 1 are pairs of one row)
                    else mu[1] = *(const uint16_t*)(mbase + mf_here(2u * e1));
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int top = (int)((uint32_t)y[v] << 8) >> 16;  // bits 8..23, signed
                        const int r = (int)((uint32_t)((top * (int)mu[v >> 1]) >> 15) << 8);
                        y[v] = mu[v >> 1] != 0xffffu ? r : y[v];
                    }
                }
                uint8_t* const os = out_lds + ct * ((8u / (uint32_t)(PAIRS == 3 ? 1 : PAIRS)) * G::kRowOutPitch) + 16u * G::kFb * step;
                if constexpr (kWideOut && PAIRS == 1) {
                    // Stereo: TWELVE contiguous bytes per lane instead of two frames of six in two rows.  Lanes n and n + 1 (n even) hold
                    // frames n and n + 1 of pair-rows 2 g and 2 g + 1; the even lane takes both frames of row 2 g, the odd lane both of row
                    // 2 g + 1 -- four selects with the neighbour's value as their DPP operand (mf_pair_gather) -- and 6 n is a multiple of four for even n: three
                    // permutes, one ds_write2_b32 and one ds_write_b32 (10 cycles of the path to the LDS, MI355X_MICROARCH.md) where the six
                    // 2-byte stores took 24.
                    uint32_t f0, f1, f2, f3;
                    mf_pair_gather(y[0], y[1], y[2], y[3], f0, f1, f2, f3);
                    const uint32_t w0 = mf_perm(f1, f0, sel_lo), w1 = mf_perm(f2, f1, sel_mid), w2 = mf_perm(f3, f2, sel_top);
                    const uint32_t at = (uint32_t)(uintptr_t)(lds_u8_t)(os + wide_off);
                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write_b32 %0, %3 offset:8" : : "v"(at), "v"(w0), "v"(w1), "v"(w2) : "memory");
                } else if constexpr (kWideOut && PAIRS == 4) {
                    // Eight channels: pair-rows 2 g and 2 g + 1 are neighbouring pairs of ONE frame -- the lane's four values are twelve
                    // contiguous bytes at a multiple of four as they stand
                    const uint32_t w0 = mf_perm((uint32_t)y[1], (uint32_t)y[0], sel_lo), w1 = mf_perm((uint32_t)y[2], (uint32_t)y[1], sel_mid), w2 = mf_perm((uint32_t)y[3], (uint32_t)y[2], sel_top);
                    const uint32_t at = (uint32_t)(uintptr_t)(lds_u8_t)os;
                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write_b32 %0, %3 offset:8" : : "v"(at), "v"(w0), "v"(w1), "v"(w2) : "memory");
                } else {
                // pack: two permutes and three 16-bit stores per frame into the output image (a frame starts on an even byte)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const uint32_t lo = mf_perm((uint32_t)y[2 * q + 1], (uint32_t)y[2 * q], sel_lo);
                    const uint32_t hi = mf_perm((uint32_t)y[2 * q + 1], (uint32_t)y[2 * q], sel_hi);
                    // (written out: left to the compiler the first two become one 4-byte store, misaligned for odd frames.  LDS operations
                    // complete in order and every barrier here waits for lgkmcnt(0), so the compiler's own counts stay safe)
                    uint32_t at;
                    if constexpr (PAIRS == 3) {                      // (a tile's eight pair-rows start anywhere in a row of three)
                        const uint32_t pr = ct * 8u + 2u * g + (uint32_t)q, sr = pr / 3u;
                        at = (uint32_t)(uintptr_t)(lds_u8_t)(stage + sr * G::kRowOutPitch + 6u * (pr - 3u * sr) + G::kFb * n + 16u * G::kFb * step);
                    } else {
                        at = (uint32_t)(uintptr_t)(lds_u8_t)(os + q * (PAIRS == 1 ? G::kRowOutPitch : 6u));
                    }
                    asm volatile("ds_write_b16 %0, %1\n\tds_write_b16_d16_hi %0, %1 offset:2\n\tds_write_b16 %0, %2 offset:4"
                                 : : "v"(at), "v"(lo), "v"(hi) : "memory");
                }
                }
            };
            u32x2 h[6];
            issue_planes(kc[0], kFirst % G::kCt, h);
            if constexpr (kPipe) {
                // The tiles as a software pipeline of ONE wave: a tile's twelve matrix instructions in two halves -- (a) the six that
                // make s0, s1, s2, (b) the six that make s3, s4, s5 -- with the PREVIOUS tile's last twenty-eight vector instructions
                // (T2, W, y, clamp, pack, store: they need only U, s4, s5 of it) written between them and this tile's first twelve
                // (T0, T1, U) behind (b): in program order every vector instruction has matrix instructions of another output's
                // around it that it does not depend on, and the accumulators a half writes are dead by then -- no register more.
                // (Round 5: a wave alone alternated 192 cycles of matrix pipe and 160 of vector pipe, and with three waves a SIMD the
                // others filled a third of that: 303 cycles a tile with every global access compiled out.)
                int u_prev[4];
                v4i s4p, s5p;
                static_for([&](auto ic) __attribute__((always_inline)) {
                    constexpr uint32_t i = decltype(ic)::value;
                    constexpr bool kHas = i < G::kTilesPerWave, kPrev = i > 0;
                    constexpr uint32_t ii = kHas ? i : G::kTilesPerWave - 1u;
                    constexpr uint32_t set = (kFirst + ii) / G::kCt - kFirst / G::kCt;
                    constexpr uint32_t ip = kPrev ? i - 1u : 0u;
                    constexpr uint32_t set_p = (kFirst + ip) / G::kCt - kFirst / G::kCt, ct_p = (kFirst + ip) % G::kCt;
                    static_assert(set < G::kKcSets, "a wave's tiles touch kKcSets steps");
                    v4i bd[3];
                    v4i s0, s1 = v4i{0, 0, 0, 0}, s2, s3 = v4i{0, 0, 0, 0}, s4 = v4i{0, 0, 0, 0}, s5 = v4i{0, 0, 0, 0};
                    const v4i (&c)[4] = a[HB ? 0 : set];
                    if constexpr (kHas) {
                        const uint8_t* const bi = my_bias + (step0 + set) * G::kBiasStep;
                        if constexpr (G::kDma) read_bias2(bi, s0, s2);         // (in front of the wait in take_planes)
                        take_planes(h, bd);
                        if constexpr (HB) { s0 = hb_s0; s2 = hb_s2; }
                        else if constexpr (!G::kDma) { s0 = *(const v4i*)bi; s2 = *(const v4i*)(bi + G::kBiasStep / 2); }
                        s0 = MF_MFMA(bd[0], c[0], s0);
                        s1 = MF_MFMA(bd[0], c[1], s1);
                        s2 = MF_MFMA(bd[0], c[2], s2);
                        s1 = MF_MFMA(bd[1], c[0], s1);
                        s2 = MF_MFMA(bd[1], c[1], s2);
                        s2 = MF_MFMA(bd[2], c[0], s2);
                    }
                    if constexpr (kPrev) {
                        int y[4];
#pragma unroll
                        for (int v = 0; v < 4; v++) y[v] = mf_recombine_tail(u_prev[v], s4p[v], s5p[v]);
                        finish_tile(std::integral_constant<uint32_t, set_p>{}, std::integral_constant<uint32_t, ct_p>{}, y);
                    }
                    if constexpr (kHas) {
                        s3 = MF_MFMA(bd[0], c[3], s3);
                        s3 = MF_MFMA(bd[1], c[2], s3);
                        s3 = MF_MFMA(bd[2], c[1], s3);
                        s4 = MF_MFMA(bd[1], c[3], s4);
                        s4 = MF_MFMA(bd[2], c[2], s4);
                        s5 = MF_MFMA(bd[2], c[3], s5);
                        if constexpr (i + 1 < G::kTilesPerWave) {
                            constexpr uint32_t set_n = (kFirst + i + 1) / G::kCt - kFirst / G::kCt, ct_n = (kFirst + i + 1) % G::kCt;
                            issue_planes(kc[set_n], ct_n, h);
                        }
#pragma unroll
                        for (int v = 0; v < 4; v++) u_prev[v] = mf_recombine_head(s0[v], s1[v], s2[v], s3[v]);
                        s4p = s4; s5p = s5;
                    }
                }, std::make_integer_sequence<int, (int)G::kTilesPerWave + 1>{});
            } else {
            static_for([&](auto ic) __attribute__((always_inline)) {
                constexpr uint32_t i = decltype(ic)::value;
                v4i bd[3];
                constexpr uint32_t set = (kFirst + i) / G::kCt - kFirst / G::kCt, ct = (kFirst + i) % G::kCt;
                static_assert(set < G::kKcSets, "a wave's tiles touch kKcSets steps");
                const uint32_t step = step0 + set;
                const uint8_t* const bi = my_bias + step * G::kBiasStep;
                v4i s0, s1 = v4i{0, 0, 0, 0}, s2, s3 = v4i{0, 0, 0, 0}, s4 = v4i{0, 0, 0, 0}, s5 = v4i{0, 0, 0, 0};
                if constexpr (G::kDma) read_bias2(bi, s0, s2);                 // (in front of the wait in take_planes)
                take_planes(h, bd);
                if constexpr (HB) { s0 = hb_s0; s2 = hb_s2; }
                else if constexpr (!G::kDma) { s0 = *(const v4i*)bi; s2 = *(const v4i*)(bi + G::kBiasStep / 2); }
                const v4i (&c)[4] = a[HB ? 0 : set];
                s0 = MF_MFMA(bd[0], c[0], s0);
                s1 = MF_MFMA(bd[0], c[1], s1);
                s2 = MF_MFMA(bd[0], c[2], s2);
                s3 = MF_MFMA(bd[0], c[3], s3);
                s1 = MF_MFMA(bd[1], c[0], s1);
                s2 = MF_MFMA(bd[1], c[1], s2);
                s3 = MF_MFMA(bd[1], c[2], s3);
                s4 = MF_MFMA(bd[1], c[3], s4);
                s2 = MF_MFMA(bd[2], c[0], s2);
                s3 = MF_MFMA(bd[2], c[1], s3);
                s4 = MF_MFMA(bd[2], c[2], s4);
                s5 = MF_MFMA(bd[2], c[3], s5);
                if constexpr (i + 1 < G::kTilesPerWave) {
                    // the next tile's planes, while this one's matrix instructions run
                    constexpr uint32_t set_n = (kFirst + i + 1) / G::kCt - kFirst / G::kCt, ct_n = (kFirst + i + 1) % G::kCt;
                    issue_planes(kc[set_n], ct_n, h);
                }
                int y[4];
#pragma unroll
                for (int v = 0; v < 4; v++) y[v] = mf_recombine(s0[v], s1[v], s2[v], s3[v], s4[v], s5[v]);
                finish_tile(std::integral_constant<uint32_t, set>{}, std::integral_constant<uint32_t, ct>{}, y);
            }, std::make_integer_sequence<int, (int)G::kTilesPerWave>{});
            }
        };
#ifndef MF_DIAG_IO_ONLY
        static_for([&](auto pc) __attribute__((always_inline)) {
            if (wave % G::kCt == (uint32_t)decltype(pc)::value) {         // (wave-uniform)
                if (ramped) run_tiles(pc, std::true_type{});
                else run_tiles(pc, std::false_type{});
            }
        }, std::make_integer_sequence<int, (int)G::kCt>{});
#endif
#ifndef MF_WG_NO_PRIO
        __builtin_amdgcn_s_setprio(kPrioD);                  // (the tiles run at priority 3: -2 % on the headline in round 4, -4 % in round 5, same box, alternating)
#endif
        if constexpr (G::kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's pieces of the next run have landed; the last pass's stores with them, long since)
        wg_barrier<0>();                                    // the output image is whole; the planes are free (kDma: and the next run is in its buffer)

        // ---- (D) the unit leaves as lane-contiguous pieces.  vmcnt counts loads and stores together, in issue order: the next
        // unit's input -- requested a whole phase (C) ago -- is waited for HERE, in front of the stores, not behind them ----
        // (the unit after the next: its descriptor is asked for here, a phase before its loads are issued)
        const uint32_t u_n2 = u_nxt + n_groups;
        Unit wk_n2 = wk;
        if constexpr (G::kSpan && !kLateLoads) wk_n2 = fetch_unit(u_n2 < n_work ? u_n2 : u_cur);
        if constexpr (!G::kDma) asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]));
        if constexpr (G::kInRounds > 3 && !G::kDma) asm volatile("" : "+v"(raw[3]));
        if constexpr (G::kInRounds > 4) asm volatile("" : "+v"(raw[4]));
        if constexpr (G::kInRounds > 5) asm volatile("" : "+v"(raw[5]));
        if constexpr (G::kInRounds > 6) asm volatile("" : "+v"(raw[6]));
        if constexpr (G::kInRounds > 7) asm volatile("" : "+v"(raw[7]));
        static_assert(G::kInRounds <= 8, "the rounds of loads kept alive across (D)");
        {
            uint8_t* const unit_dst = dst + wk.dst0;
            const uint32_t out_bytes = n_blocks * G::kRowOut;
            u32x4 op[G::kStoreRounds];
#pragma unroll
            for (int k = 0; k < (int)G::kStoreRounds; k++) {
                uint32_t f = G::kThreads * k + tid;
                if (f >= G::kOutPieces) f = G::kOutPieces - 1;
                if constexpr (G::kOutLinear) {
                    op[k] = *(const u32x4*)(stage + 16u * f);
                } else {
                    const uint32_t row = f / (G::kRowOut / 16u);
                    op[k] = *(const u32x4*)(stage + row * G::kRowOutPitch + 16u * (f - row * (G::kRowOut / 16u)));
                }
            }
#pragma unroll
            for (int k = 0; k < (int)G::kStoreRounds; k++) {
                const uint32_t o = mf_here(16u * (G::kThreads * k + tid));
#if defined(MF_DIAG_NO_STORE)
                if (o < out_bytes && n_blocks > 1000000u) *(u32x4_u*)(unit_dst + o) = op[k];
#else
                if (o < out_bytes) __builtin_nontemporal_store(op[k], (u32x4_u*)(unit_dst + o));
#endif
            }
        }
        // (the output image has been read.  Packed sources' input image is the pass's run piece for piece, and the output image is too:
        // a lane stages into the very slots it has just emptied -- LDS operations of a wave complete in order -- so nothing has to
        // wait for another wave here; the row-by-row images of the planar and half-band forms do)
        if constexpr (G::kDma) {
            // (S) straight away: the split reads the run's buffer and writes the planes, (D) read the output image -- nothing of one is
            // the other's, so no barrier stands between them: two a pass, not three
            if (u_nxt >= n_work) break;                     // (uniform)
            split_all(wk_nxt.first);
            wk = wk_nxt;
            wk_nxt = wk_n2;
            u_cur = u_nxt;
            u_nxt += n_groups;
#ifndef MF_WG_NO_PRIO
            __builtin_amdgcn_s_setprio(kPrioC);
#endif
            wg_barrier<3>();                                // the planes are whole; the run's buffer and the output image are free
            if (u_nxt < n_work) issue_dma(wk_nxt);          // (uniform)
            continue;
        }
        if constexpr (!G::kSpan || !G::kOutLinear) wg_barrier<1>();
        if (u_nxt >= n_work) break;                         // (uniform)

        // ---- (A) + (S) the next unit: registers -> input image -> planes; then its successor's input is requested ----
        if constexpr (G::kSpan && !kLateLoads) {
            // ... requested as soon as the registers are free -- behind the stage's writes, in front of the split -- so that it is in
            // flight for (S) and (C), not for (C) alone (round 5: 0.3011 -> 0.3001 ms on the headline, same box, three alternating pairs;
            // that this is all it gains says the launch does not wait for its loads: tools/micro/run_copy.hip)
            stage_input(raw);
            issue_input(wk_n2, raw);
#ifndef MF_WG_NO_PRIO
            if constexpr (kPrioS != kPrioD) __builtin_amdgcn_s_setprio(kPrioS);
#endif
            wg_barrier<2>();
            split_all(wk_nxt.first);
            wk = wk_nxt;
            wk_nxt = wk_n2;
            u_cur = u_nxt;
            u_nxt += n_groups;
        } else {
        stage_and_split(raw, wk_nxt.first);
        wk = wk_nxt;
        u_cur = u_nxt;
        u_nxt += n_groups;
        wk_nxt = fetch_unit(u_nxt < n_work ? u_nxt : u_cur);
        issue_input(wk_nxt, raw);
        }
#ifndef MF_WG_NO_PRIO
        __builtin_amdgcn_s_setprio(kPrioC);
#endif
        wg_barrier<3>();
    }
}

bool src_mfma_wg_supported(uint32_t L_blk, uint32_t M_blk, uint32_t ch, uint32_t sb, uint32_t db, bool planar, bool halfband)
{
    // (planar: sb is the stream's sample size, 1..3 bytes, whatever the planes hold above it; packed: stereo, six or eight channels of
    // S24, and 16-bit stereo through the polyphase filters)
    const bool layout = planar ? (ch == 2 && sb >= 1 && sb <= 3 && !halfband)
                               : (((ch == 2 || ch == 6 || ch == 8) && sb == 3) || (ch == 2 && sb == 2 && !halfband));
    if (halfband) return layout && db == 3 && L_blk == 128u && M_blk == 256u;          // (2:1: WgGeom<.., HB>)
    return layout && db == 3 && L_blk == 160u && (M_blk + 31u) / 16u + 1u == 12u;     // (160 outputs from at most 160 inputs + 32 of history: twelve chunks)
}

// does a unit's input image -- `unit_rows` rows of 192 (half-band: 320) frames, whatever the number of blocks the unit holds -- lie inside the arena?
bool src_mfma_wg_unit_inside(int64_t src_row0, uint32_t row_src_bytes, uint64_t src_arena_bytes, uint32_t ch, uint32_t sb, uint32_t unit_rows, bool planar, uint64_t plane_stride, bool halfband)
{
    const uint64_t row_in = planar ? kWgPlaneIn : (halfband ? 320u : 192u) * sb * ch, last_plane = planar ? plane_stride : 0;
    // (packed: a pass's rows are fetched as one run of 16-byte pieces from the first row's start: the last piece may reach 15 bytes further)
    return src_row0 >= 0 && unit_rows >= 1 && (uint64_t)src_row0 + last_plane + (uint64_t)(unit_rows - 1u) * row_src_bytes + row_in + (planar ? 0u : 15u) <= src_arena_bytes;
}

template <int PLANAR, int PAIRS, bool HB, bool SRC_LE, bool DST_LE>
static hipError_t launch_wg_one(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& p, hipStream_t s, WgOccupancy* query)
{
    using G = WgGeom<OHGPU_WG_ROWS, PLANAR, PAIRS, HB>;
    auto kernel = src_mfma_wg_kernel<OHGPU_WG_ROWS, PLANAR, PAIRS, HB, SRC_LE, DST_LE>;
    const SrcFastPlan& f = b->fast;
    if (f.n_lean == 0) return hipSuccess;
    if (!src_mfma_wg_supported(p.L_blk, p.M_blk, p.channels, p.sb, p.db, G::kPlanes, HB) || p.channels != 2u * PAIRS || f.wg_unit_rows != G::kUnitRows) return hipErrorInvalidValue;
    const uint32_t cus = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    const uint32_t n_units = f.n_lean * G::kSubUnits;                // (edge units included: their loads are checked)
    uint32_t gsz = G::kGroupsPerCu * cus;                     // as many workgroups as the LDS holds
    if (gsz > n_units) gsz = n_units;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::kLdsBytes);
    if (e != hipSuccess) return e;
    if (query && query->query) {                             // (ohgpu_src_batch_occupancy: what the device grants this instantiation, nothing launched)
        int groups = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&groups, (const void*)kernel, (int)G::kThreads, (size_t)G::kLdsBytes);
        query->groups_per_cu = groups;
        query->designed_for = (int)G::kGroupsPerCu;
        query->lds_bytes = G::kLdsBytes;
        return e;
    }
    if (query && query->stop) {
        // (ohgpu_src_batch_run_timed: the two events ride on the dispatch -- its own start and end timestamps, nothing else in the queue)
        hipExtLaunchKernelGGL(kernel, dim3(gsz), dim3(G::kThreads), G::kLdsBytes, s, query->start, query->stop, 0,
                              (const LeanUnit*)f.d_lean_units, n_units, (const uint8_t*)f.d_mf_amat, (const MfStep*)f.d_mf_steps,
                              (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst, p.M_blk * (G::kPlanes ? 4u : G::kFbIn), p.src_arena_bytes);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kernel, dim3(gsz), dim3(G::kThreads), G::kLdsBytes, s,
                       (const LeanUnit*)f.d_lean_units, n_units, (const uint8_t*)f.d_mf_amat, (const MfStep*)f.d_mf_steps,
                       (const uint16_t*)f.d_planes, f.plane_stride, p.src, p.dst, p.M_blk * (G::kPlanes ? 4u : G::kFbIn), p.src_arena_bytes);
    return hipGetLastError();
}

template <int PAIRS, bool HB>
static hipError_t launch_wg_packed(const ohgpu_ctx* ctx, const ohgpu_batch* b, const SrcFastParams& prm, hipStream_t s, WgOccupancy* q)
{
    if (prm.src_le) return prm.dst_le ? launch_wg_one<0, PAIRS, HB, true, true>(ctx, b, prm, s, q) : launch_wg_one<0, PAIRS, HB, true, false>(ctx, b, prm, s, q);
    return prm.dst_le ? launch_wg_one<0, PAIRS, HB, false, true>(ctx, b, prm, s, q) : launch_wg_one<0, PAIRS, HB, false, false>(ctx, b, prm, s, q);
}

hipError_t launch_src_mfma_wg(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s, WgOccupancy* q)
{
    SrcFastParams prm = b->fast.params;
    prm.src = src;
    prm.dst = dst;
    if (b->src_planar) {
        switch (prm.sb) {                                    // (the stream's bytes per sample)
        case 3: return prm.dst_le ? launch_wg_one<1, 1, false, true, true>(ctx, b, prm, s, q) : launch_wg_one<1, 1, false, true, false>(ctx, b, prm, s, q);
        case 2: return prm.dst_le ? launch_wg_one<2, 1, false, true, true>(ctx, b, prm, s, q) : launch_wg_one<2, 1, false, true, false>(ctx, b, prm, s, q);
        case 1: return prm.dst_le ? launch_wg_one<3, 1, false, true, true>(ctx, b, prm, s, q) : launch_wg_one<3, 1, false, true, false>(ctx, b, prm, s, q);
        default: return hipErrorInvalidValue;
        }
    }
    const bool hb = b->fast.mfma_wg_halfband;
    if (prm.sb == 2) {                                       // packed 16-bit stereo (WgGeom: source format 4)
        if (prm.channels != 2 || hb) return hipErrorInvalidValue;
        if (prm.src_le) return prm.dst_le ? launch_wg_one<4, 1, false, true, true>(ctx, b, prm, s, q) : launch_wg_one<4, 1, false, true, false>(ctx, b, prm, s, q);
        return prm.dst_le ? launch_wg_one<4, 1, false, false, true>(ctx, b, prm, s, q) : launch_wg_one<4, 1, false, false, false>(ctx, b, prm, s, q);
    }
    switch (prm.channels) {
    case 2: return hb ? launch_wg_packed<1, true>(ctx, b, prm, s, q) : launch_wg_packed<1, false>(ctx, b, prm, s, q);
    case 6: return hb ? launch_wg_packed<3, true>(ctx, b, prm, s, q) : launch_wg_packed<3, false>(ctx, b, prm, s, q);
    case 8: return hb ? launch_wg_packed<4, true>(ctx, b, prm, s, q) : launch_wg_packed<4, false>(ctx, b, prm, s, q);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace ohgpu
