// Ramp.cpp -- control-plane ramp algebra (see Ramp.h for the reference lines each piece follows).
#include "Ramp.h"

#include <algorithm>
#include <mutex>

#include "../../include/ohgpu.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- Jiffies
TBool Jiffies::IsValidSampleRate(TUint aSampleRate)
{
    switch (aSampleRate) {
    case 7350: case 8000: case 11025: case 12000: case 14700: case 16000: case 22050: case 24000:
    case 29400: case 32000: case 44100: case 48000: case 88200: case 96000: case 176400: case 192000:
    case 352800: case 384000:
    case 2822400: case 5644800: case 11289600:     // DSD rates
        return true;
    default:
        return false;
    }
}

TUint Jiffies::PerSample(TUint aSampleRate)
{
    if (!IsValidSampleRate(aSampleRate)) {
        THROW(SampleRateInvalid);                   // Msg.cpp:470-472
    }
    return kPerSecond / aSampleRate;                // every legal rate divides kPerSecond exactly (Msg.h:213-233)
}

TUint Jiffies::ToBytesSampleBlock(TUint& aJiffies, TUint aJiffiesPerSample, TUint aNumChannels, TUint aBitsPerSubsample, TUint aSamplesPerBlock)
{
    ASSERT(aSamplesPerBlock != 0);
    aJiffies -= aJiffies % (aJiffiesPerSample * aSamplesPerBlock);     // round down to a whole block (Msg.cpp:484)
    const TUint subsamples = (aJiffies / aJiffiesPerSample) * aNumChannels;
    return (subsamples * aBitsPerSubsample + 7) / 8;
}

TUint Jiffies::ToBytes(TUint& aJiffies, TUint aJiffiesPerSample, TUint aNumChannels, TUint aBitsPerSubsample)
{
    return ToBytesSampleBlock(aJiffies, aJiffiesPerSample, aNumChannels, aBitsPerSubsample, 1);
}

void Jiffies::RoundDown(TUint& aJiffies, TUint aSampleRate)
{
    aJiffies -= aJiffies % PerSample(aSampleRate);
}

void Jiffies::RoundUp(TUint& aJiffies, TUint aSampleRate)
{
    const TUint jps = PerSample(aSampleRate);
    aJiffies += jps - 1;
    aJiffies -= aJiffies % jps;
}

void Jiffies::RoundDownNonZeroSampleBlock(TUint& aJiffies, TUint aSampleBlockJiffies)
{
    TUint down = aJiffies - (aJiffies % aSampleBlockJiffies);
    if (down == 0) {                                 // never round a non-empty request down to nothing (Msg.cpp:508-512)
        down = aJiffies + aSampleBlockJiffies - 1;
        down -= down % aSampleBlockJiffies;
    }
    aJiffies = down;
}

TUint Jiffies::SongcastTicksPerSecond(TUint aSampleRate)
{
    switch (aSampleRate) {
    case 7350: case 11025: case 14700: case 22050: case 29400: case 44100: case 88200: case 176400: case 352800:
        return 44100 * 256;
    case 8000: case 12000: case 16000: case 24000: case 32000: case 48000: case 96000: case 192000: case 384000:
        return 48000 * 256;
    default:
        THROW(SampleRateInvalid);
    }
}

TUint Jiffies::ToSongcastTime(TUint aJiffies, TUint aSampleRate)
{
    return static_cast<TUint>((static_cast<TUint64>(aJiffies) * SongcastTicksPerSecond(aSampleRate)) / kPerSecond);
}

TUint64 Jiffies::FromSongcastTime(TUint64 aSongcastTime, TUint aSampleRate)
{
    return (aSongcastTime * kPerSecond) / SongcastTicksPerSecond(aSampleRate);
}

// ---------------------------------------------------------------- RampArray
const TUint16* RampArray()
{
    static TUint16 table[kRampArrayCount];
    static std::once_flag once;
    std::call_once(once, [] { ohgpu_ramp_table(table); });   // same generator the device table comes from
    return table;
}

// ---------------------------------------------------------------- Ramp
Ramp::Ramp()
{
    Reset();
}

void Ramp::Reset()
{
    iStart = iEnd = kMax;
    iDirection = ENone;
    iEnabled = false;
}

void Ramp::SetMuted()
{
    iStart = iEnd = kMin;
    iDirection = EMute;
    iEnabled = true;
}

void Ramp::SelectLowerRampPoints(TUint aRequestedStart, TUint aRequestedEnd)
{
    iStart = std::min(iStart, aRequestedStart);
    iEnd = std::min(iEnd, aRequestedEnd);
    iDirection = (iStart == iEnd) ? ENone : (iStart > iEnd ? EDown : EUp);
}

TBool Ramp::DoValidate() const
{
    if (iStart > kMax || iEnd > kMax) {
        return false;
    }
    switch (iDirection) {
    case ENone: return iStart == iEnd;
    case EUp:   return iStart < iEnd;
    case EDown: return iStart > iEnd;
    case EMute: return iStart == iEnd && iStart == kMin;
    }
    ASSERTS();
}

void Ramp::Validate() const
{
    ASSERT(DoValidate());
}

TBool Ramp::Set(TUint aStart, TUint aFragmentSize, TUint aRemainingDuration, EDirection aDirection, Ramp& aSplit, TUint& aSplitPos)
{
    ASSERT(aRemainingDuration >= aFragmentSize);
    ASSERT(aDirection != ENone);
    iEnabled = true;
    aSplit.Reset();
    aSplitPos = 0xffffffff;

    // Where this fragment leaves the ramp: it moves by its share of what is still to go -- toGo * fragment / remaining, rounded UP, so
    // that a ramp always completes within its duration (Msg.cpp:603-605) -- as ONE signed step from aStart, held to the scale.  What
    // the hold takes away is the rounding's overshoot and never more than a fragment's worth of it: anything larger is a caller's
    // inconsistent duration, and asserts as the reference's two branches do (Msg.cpp:606-628).
    const TInt64 sign = (aDirection == EDown) ? -1 : 1;
    const TUint64 toGo = (aDirection == EDown) ? aStart - kMin : kMax - aStart;
    const TInt64 landed = (TInt64)aStart + sign * (TInt64)((toGo * aFragmentSize + aRemainingDuration - 1) / aRemainingDuration);
    const TInt64 held = std::min<TInt64>(std::max<TInt64>(landed, kMin), kMax);
    ASSERT((TUint64)(sign * (landed - held)) <= (TUint64)(TUint)(aFragmentSize - 1));
    const TUint newEnd = (TUint)held;

    if (iDirection == ENone) {
        iDirection = aDirection;
        iStart = aStart;
        iEnd = newEnd;
    }
    else if (iDirection == aDirection) {
        SelectLowerRampPoints(aStart, newEnd);
    }
    else {
        // Existing and requested ramps run in opposite directions.  Treat both as straight lines over
        // [0, aFragmentSize]; (lo0 -> lo1) is the one that starts lower, (hi0 -> hi1) the other.  If they cross
        // strictly inside the fragment, this ramp keeps the rising part up to the crossing and aSplit takes
        // the falling part after it; otherwise the lower start / lower end win (Msg.cpp:637-701).
        TInt64 lo0, lo1, hi0, hi1;
        if (iStart < aStart) {
            lo0 = iStart; lo1 = iEnd; hi0 = aStart; hi1 = newEnd;
        }
        else {
            lo0 = aStart; lo1 = newEnd; hi0 = iStart; hi1 = iEnd;
        }
        const TInt64 slopeDiff = (lo1 - lo0) - (hi1 - hi0);
        TBool crossed = false;
        if (slopeDiff != 0) {
            const TInt64 x = (static_cast<TInt64>(aFragmentSize) * (hi0 - lo0)) / slopeDiff;
            const TInt64 y = ((lo1 - lo0) * (hi0 - lo0)) / slopeDiff + lo0;
            if (x > 0 && static_cast<TUint>(x) < aFragmentSize) {
                crossed = true;
                aSplitPos = static_cast<TUint>(x);
                aSplit.iStart = static_cast<TUint>(y);
                aSplit.iEnd = std::min(iEnd, newEnd);
                aSplit.iDirection = (aSplit.iStart == aSplit.iEnd) ? ENone : EDown;
                aSplit.iEnabled = true;
                const TUint first = std::min(iStart, aStart);
                iDirection = (first == static_cast<TUint>(y)) ? ENone : EUp;
                iStart = first;
                iEnd = static_cast<TUint>(y);
            }
        }
        if (!crossed) {
            SelectLowerRampPoints(aStart, newEnd);
        }
    }
    ASSERT(DoValidate());                           // Msg.cpp:703-710
    return aSplit.IsEnabled();
}

Ramp Ramp::Split(TUint aNewSize, TUint aCurrentSize)
{
    Ramp remaining;
    remaining.iEnd = iEnd;
    remaining.iDirection = iDirection;
    remaining.iEnabled = true;
    // the first part covers aNewSize/aCurrentSize of the span, rounded toward the start value
    const TUint span = (iDirection == EUp) ? iEnd - iStart : iStart - iEnd;
    const TUint part = static_cast<TUint>((static_cast<TUint64>(span) * aNewSize) / aCurrentSize);
    iEnd = (iDirection == EUp) ? iStart + part : iStart - part;
    if (iStart == iEnd) {
        iDirection = ENone;
    }
    remaining.iStart = iEnd;
    Validate();
    remaining.Validate();
    return remaining;
}

TUint Ramp::MedianMultiplier(const Ramp& aRamp)
{
    TUint mid;
    switch (aRamp.Direction()) {
    case EUp:   mid = aRamp.Start() + (aRamp.End() - aRamp.Start()) / 2; break;
    case EDown: mid = aRamp.Start() - (aRamp.Start() - aRamp.End()) / 2; break;
    case EMute: return 0;
    default:    mid = aRamp.Start(); break;
    }
    const TUint index = (kMax - kMin - mid + (1 << 4)) >> 5;
    return index < kRampArrayCount ? RampArray()[index] : 0;
}

}  // namespace Media
}  // namespace OpenHome
