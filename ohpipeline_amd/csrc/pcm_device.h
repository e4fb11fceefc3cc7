// pcm_device.h -- per-subsample device primitives shared by every kernel of the PCM hot path.
//
// Everything here works on ONE subsample held as a "BE word": the subsample's bytes in pipeline
// (big-endian) order, left-justified in a 32-bit register -- exactly the value
// FlywheelInput::AppendSubsample8/16/24/32 writes (OpenHome/Media/Pipeline/StarvationRamper.cpp:117-147).
// All arithmetic is integer and follows the reference expression by expression; citations are
// file:line relative to the reference tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ohgpu.h"

namespace ohgpu {

constexpr uint32_t kRampMax = 16384u;        // Ramp::kMax, Msg.h:258
constexpr uint32_t kRampTableCount = 512u;   // kRampArrayCount, RampArray.h:76
constexpr uint32_t kMaxBytes = 9216u;        // AudioData::kMaxBytes, Msg.h:117

// DecodedAudio::CopyToBigEndian16/24/32 (Msg.cpp:380-408) fused with AppendSubsampleN: assemble the
// left-justified BE word from sb packed bytes in either byte order.
__device__ __forceinline__ uint32_t load_be_word(const uint8_t* __restrict__ p, uint32_t sb, bool little)
{
    uint32_t w = 0;
    if (little) {
        for (uint32_t b = 0; b < sb; b++) w |= (uint32_t)p[sb - 1 - b] << (24 - 8 * b);
    } else {
        for (uint32_t b = 0; b < sb; b++) w |= (uint32_t)p[b] << (24 - 8 * b);
    }
    return w;
}

// MsgPlayablePcm::ApplyAttenuation (Msg.cpp:2736-2751): TInt16 * TUint promotes to unsigned, the
// divide is unsigned, the result is truncated to 16 bits.  16-bit audio only (host validates).
__device__ __forceinline__ uint32_t attenuate_word(uint32_t w, uint32_t attenuation)
{
    const int32_t s16 = (int32_t)(int16_t)(w >> 16);
    const uint32_t prod = (uint32_t)s16 * attenuation;
    return ((prod / 256u) & 0xffffu) << 16;
}

// RampApplicator::GetNextSample, the per-frame part (Msg.cpp:835-837): ramp value -> table index.
// i*total uses TInt arithmetic and C division (truncation toward zero, total may be negative).
__device__ __forceinline__ uint32_t ramp_index(uint32_t ramp_start, int32_t total, int32_t i, int32_t n_frames)
{
    uint32_t ramp = (n_frames == 1) ? ramp_start : ramp_start - (uint32_t)((i * total) / (n_frames - 1));
    ramp &= 0xffffu;                                                  // (TUint16) cast
    const uint32_t idx = (kRampMax - ramp + (1u << 4)) >> 5;         // unsigned arithmetic, as in the reference
    return idx < kRampTableCount - 1 ? idx : kRampTableCount - 1;    // std::min(kRampArrayCount-1, ...)
}

// x / d for x < 2^31 with the host's multiplier (m == 0: d == 1)
__device__ __forceinline__ uint32_t udiv_magic(uint32_t x, uint32_t m, uint32_t s)
{
    return m ? (__umulhi(x, m) >> s) : x;
}

// ramp_index (pcm_device.h) with the division by (n_frames - 1) replaced by the exact multiplier
__device__ __forceinline__ uint32_t ramp_index_magic(uint32_t ramp_start, int32_t total, uint32_t i, uint32_t n_frames,
                                                     uint32_t m, uint32_t s)
{
    uint32_t ramp = ramp_start;
    if (n_frames != 1) {
        // TInt arithmetic, Msg.cpp:835.  Both factors fit 24 bits (i < 2^17: validation; |total| <= 2^16), so the full-rate
        // 24-bit multiply gives the same 32-bit product as the quarter-rate 32-bit one.
        const int32_t prod = __mul24((int)i, (int)total);
        const uint32_t mag = udiv_magic((uint32_t)(prod < 0 ? -prod : prod), m, s);
        ramp = ramp_start - (uint32_t)(prod < 0 ? -(int32_t)mag : (int32_t)mag);   // C division truncates toward zero
    }
    ramp &= 0xffffu;
    const uint32_t idx = (kRampMax - ramp + (1u << 4)) >> 5;
    return idx < kRampTableCount - 1 ? idx : kRampTableCount - 1;
}

// RampApplicator::GetNextSample, the per-subsample part (Msg.cpp:840-895): top 16 bits * Q15 >> 15
// (arithmetic shift), low byte(s) zeroed; 8-bit keeps one byte; 32-bit 6-channel gets channel<<4.
__device__ __forceinline__ uint32_t ramp_word(uint32_t w, uint32_t mult, uint32_t sb, uint32_t channels, uint32_t c)
{
    const int32_t s16 = (int32_t)(int16_t)(w >> 16);
    const int32_t r = (s16 * (int32_t)mult) >> 15;
    uint32_t o = ((uint32_t)r & 0xffffu) << 16;
    if (sb == 1) o &= 0xff000000u;
    if (sb == 4 && channels == 6) o |= (c << 4) & 0xffu;
    return o;
}

// MsgPlayableSilence::ReadBlock (Msg.cpp:2874-2893): zeros, except that 6-channel streams read from a
// buffer whose first 32 bytes carry 0x00,0x10..0x70 in every 4th byte; each <= kMaxBytes chunk restarts
// at the head of that buffer.  pos = byte position of this subsample inside the playable.
__device__ __forceinline__ uint32_t silence_word(uint64_t pos, uint32_t sb, uint32_t channels)
{
    if (channels != 6) return 0;
    const uint32_t frame_bytes = channels * sb;
    const uint32_t max_bytes = kMaxBytes - (kMaxBytes % frame_bytes);
    const uint32_t r0 = (uint32_t)(pos % max_bytes);
    uint32_t w = 0;
    for (uint32_t b = 0; b < sb; b++) {
        const uint32_t r = r0 + b;
        const uint32_t v = (r < 32u && (r & 3u) == 3u) ? ((r >> 2) << 4) : 0u;
        w |= v << (24 - 8 * b);
    }
    return w;
}

// Depth conversion = keep the db most significant bytes of the left-justified word
// (RampGenerator::ProcessFragment, StarvationRamper.cpp:281-327; "case 32" zero LSB on request),
// then store in the requested byte order.
__device__ __forceinline__ void store_word(uint8_t* __restrict__ p, uint32_t w, uint32_t db, bool little, bool zero_lsb32)
{
    if (zero_lsb32 && db == 4) w &= 0xffffff00u;
    if (little) {
        for (uint32_t b = 0; b < db; b++) p[db - 1 - b] = (uint8_t)(w >> (24 - 8 * b));
    } else {
        for (uint32_t b = 0; b < db; b++) p[b] = (uint8_t)(w >> (24 - 8 * b));
    }
}

// Resampler output stage: exact integer-valued fp64 accumulator -> S24 with round-half-up and saturation
// (DESIGN.md "Resampler": y = clamp((acc + 2^27) >> 28)).  |acc| < 2^53 so every step below is exact.
__device__ __forceinline__ int32_t src_round_s24(double acc)
{
    // |acc * 2^-28| < 2^25: the scaled sum, the +0.5 and the floor are all exact, the conversion is in range
    const int32_t y = (int32_t)floor(fma(acc, 1.0 / 268435456.0, 0.5));
    return y > 8388607 ? 8388607 : (y < -8388608 ? -8388608 : y);
}

}  // namespace ohgpu
