// RampGenerator.h -- host-side mirrors of the two helpers StarvationRamper uses when a stream runs dry
// (OpenHome/Media/Pipeline/StarvationRamper.cpp):
//   FlywheelInput (:74-193)   collects the most recent audio as planar big-endian 32-bit samples
//   RampGenerator (:196-364)  extrapolates it with FlywheelRamperManager (row N1), cuts the 32-bit result back to the stream's
//                             depth (row a12), and hands it out as <= 1 ms MsgAudioPcm messages carrying a down-ramp
// Same constructor arguments, same calls; no audio arithmetic on the host: a11 = ohgpu_fmt (UNPACK_PLANAR), the ramp audio =
// ohgpu_flywheel, a12 = ohgpu_pcm (32 -> N bits, OHGPU_FLAG_ZERO_LSB32 for N = 32), chained on the device.  The reference runs
// RampGenerator's work on its own thread behind a semaphore; here Start() returns when the messages are queued.
#pragma once

#include <deque>
#include <vector>

#include "Msg.h"

namespace OpenHome {
namespace Media {

class FlywheelInput : public IPcmProcessor {             // StarvationRamper.cpp:74-193
    static const TUint kSubsampleBytes = 4;
public:
    FlywheelInput(MsgFactory& aFactory, TUint aMaxJiffies);
    /** Reads the playables of aAudio (consuming them) and returns channel-contiguous BE 32-bit audio of the last aJiffies. */
    const Brx& Prepare(std::deque<MsgAudio*>& aAudio, TUint aJiffies, TUint aSampleRate, TUint aBitDepth, TUint aNumChannels);
private: // from IPcmProcessor
    void BeginBlock() override {}
    void ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) override;
    void ProcessSilence(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) override;
    void EndBlock() override {}
    void Flush() override {}
private:
    MsgFactory& iFactory;
    std::vector<TByte> iPacked;      // interleaved, as read
    std::vector<TByte> iPlanar;
    Brn iResult;
    TUint iSubsampleBytes = 0, iChannels = 0;
};

class RampGenerator {                                    // StarvationRamper.cpp:196-364
public:
    RampGenerator(MsgFactory& aFactory, TUint aInputJiffies, TUint aRampJiffies);
    ~RampGenerator();
    void Start(const Brx& aRecentAudio, TUint aSampleRate, TUint aNumChannels, TUint aBitDepth, TUint aCurrentRampValue);
    TBool TryGetAudio(Msg*& aMsg);                        // false once every message has been handed out
private:
    MsgFactory& iFactory;
    TUint iInputJiffies, iRampJiffies;
    std::deque<Msg*> iQueue;
};

} // namespace Media
} // namespace OpenHome
