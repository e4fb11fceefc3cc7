// Ramp.h -- host-side (control plane) ramp algebra and the pipeline's time unit.
//
// Mirrors, with the same names and argument meaning, the reference's
//   Jiffies          OpenHome/Media/Pipeline/Msg.h:190-238, Msg.cpp:412-558
//   Ramp             OpenHome/Media/Pipeline/Msg.h:253-286, Msg.cpp:569-807
//   RampApplicator::MedianMultiplier   Msg.cpp:901-920
// These stay on the CPU: they decide per-message ramp endpoints and byte windows; the GPU applies them.
#pragma once

#include "OhTypes.h"

namespace OpenHome {
namespace Media {

OH_EXCEPTION(SampleRateInvalid);

class Jiffies {
public:
    static const TUint kPerSecond = 56448000;   // lcm(384000, 352800)
    static const TUint kPerMs = kPerSecond / 1000;
public:
    static TBool IsValidSampleRate(TUint aSampleRate);
    static TUint PerSample(TUint aSampleRate);   // throws SampleRateInvalid
    static TUint ToBytes(TUint& aJiffies, TUint aJiffiesPerSample, TUint aNumChannels, TUint aBitsPerSubsample);
    static TUint ToBytesSampleBlock(TUint& aJiffies, TUint aJiffiesPerSample, TUint aNumChannels, TUint aBitsPerSubsample, TUint aSamplesPerBlock);
    static void RoundDown(TUint& aJiffies, TUint aSampleRate);
    static void RoundUp(TUint& aJiffies, TUint aSampleRate);
    static void RoundDownNonZeroSampleBlock(TUint& aJiffies, TUint aSampleBlockJiffies);
    static TUint ToSongcastTime(TUint aJiffies, TUint aSampleRate);
    static TUint64 FromSongcastTime(TUint64 aSongcastTime, TUint aSampleRate);
    static TUint SongcastTicksPerSecond(TUint aSampleRate);
    static TUint ToMs(TUint aJiffies) { return aJiffies / kPerMs; }
    static TUint ToSamples(TUint aJiffies, TUint aSampleRate) { return aJiffies / PerSample(aSampleRate); }
    static const TUint kMaxJiffiesPerSample = kPerSecond / 7350;
};

class Ramp {
public:
    static const TUint kMax = 1 << 14;
    static const TUint kMin = 0;
    enum EDirection { ENone, EUp, EDown, EMute };
public:
    Ramp();
    void Reset();
    // returns true iff aSplit is set
    TBool Set(TUint aStart, TUint aFragmentSize, TUint aRemainingDuration, EDirection aDirection, Ramp& aSplit, TUint& aSplitPos);
    void SetMuted();
    Ramp Split(TUint aNewSize, TUint aCurrentSize);
    TUint Start() const { return iStart; }
    TUint End() const { return iEnd; }
    EDirection Direction() const { return iDirection; }
    TBool IsEnabled() const { return iEnabled; }
    // Q15 multiplier at the ramp's midpoint (what VolumeRamper hands to a hardware volume control)
    static TUint MedianMultiplier(const Ramp& aRamp);
private:
    void SelectLowerRampPoints(TUint aRequestedStart, TUint aRequestedEnd);
    TBool DoValidate() const;
    void Validate() const;
private:
    TUint iStart;
    TUint iEnd;
    EDirection iDirection;
    TBool iEnabled;
};

// RampArray.h:7-74 regenerated from its closed form (see csrc/host_design.cpp)
const TUint16* RampArray();
static const TUint kRampArrayCount = 512;

}  // namespace Media
}  // namespace OpenHome
