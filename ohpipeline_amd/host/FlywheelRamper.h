// FlywheelRamper.h -- host-side mirror of the reference's FlywheelRamperManager (OpenHome/Media/FlywheelRamper.h:69-88)
// over the GPU kernel.  Same constructor arguments, same Ramp() call, same output through
// IPcmProcessor in blocks of at most 1 ms (FlywheelRamper.cpp:52-63, 124-130); the computation happens in
// ohgpu_flywheel_*.  FlywheelRamperBatch is the shape the GPU wants: the requests of all streams that ran dry in the same
// period, one launch.
#pragma once

#include <vector>

#include "Msg.h"

namespace OpenHome {
namespace Media {

class FlywheelRamper {                                   // the statics callers of the reference use (FlywheelRamper.h:49-56)
public:
    static const TUint kBytesPerSample = 4;              // 32 bit audio
    static TUint SampleCount(TUint aSampleRate, TUint aJiffies) { return Jiffies::ToSamples(aJiffies, aSampleRate); }
    static TUint DecimationFactor(TUint aSampleRate);    // FlywheelRamper.cpp:316-331
};

class FlywheelRamperBatch {
public:
    explicit FlywheelRamperBatch(MsgFactory& aFactory);
    /** aSamples: planar big-endian 32-bit audio of aChannelCount channels (FlywheelInput's output), copied. */
    void Add(IPcmProcessor& aOutput, const Brx& aSamples, TUint aSampleRate, TUint aChannelCount, TUint aInputJiffies, TUint aOutputJiffies);
    void Run();                                          // launches, waits, delivers in the order the requests were added
    TUint Count() const { return (TUint)iItems.size(); }
private:
    struct Item { IPcmProcessor* output; TUint64 srcOffset; TUint channelBytes, sampleRate, channels, inSamples, outFrames, blockFrames; };
    MsgFactory& iFactory;
    std::vector<Item> iItems;
    std::vector<TByte> iSrc;
};

class FlywheelRamperManager {                            // FlywheelRamper.h:69-88
public:
    static const TUint kMaxOutputJiffiesBlockSize;       // 1 ms
public:
    FlywheelRamperManager(MsgFactory& aFactory, IPcmProcessor& aOutput, TUint aInputJiffies, TUint aOutputJiffies);
    void Ramp(const Brx& aSamples, TUint aSampleRate, TUint aChannelCount);
private:
    MsgFactory& iFactory;
    IPcmProcessor& iOutput;
    TUint iInputJiffies, iOutputJiffies;
};

} // namespace Media
} // namespace OpenHome
