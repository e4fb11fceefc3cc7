// DecodedAudioAggregator.cpp -- see DecodedAudioAggregator.h.  File:line comments are relative to the reference tree.
#include "DecodedAudioAggregator.h"

#include <algorithm>

using namespace OpenHome;
using namespace OpenHome::Media;

// ---------------------------------------------------------------------------------------------- DecodedAudioAggregator
const TUint DecodedAudioAggregator::kSupportedMsgTypes =   eMode | eTrack | eDrain | eDelay | eEncodedStream | eMetatext
                                                         | eStreamInterrupted | eHalt | eFlush | eWait | eDecodedStream
                                                         | eAudioPcm | eAudioDsd | eQuit;     // DecodedAudioAggregator.cpp:13-26

DecodedAudioAggregator::DecodedAudioAggregator(IPipelineElementDownstream& aDownstreamElement)
    : PipelineElement(kSupportedMsgTypes)
    , iDownstreamElement(aDownstreamElement)
    , iDecodedAudio(nullptr)
    , iChannels(0)
    , iSampleRate(0)
    , iBitDepth(0)
    , iSupportsLatency(false)
    , iAggregationDisabled(false)
    , iAggregatedJiffies(0)
{
}

DecodedAudioAggregator::~DecodedAudioAggregator()
{
    if (iDecodedAudio != nullptr) {
        iDecodedAudio->RemoveRef();
    }
}

void DecodedAudioAggregator::Push(Msg* aMsg)
{
    ASSERT(aMsg != nullptr);
    Msg* msg = aMsg->Process(*this);
    if (msg != nullptr) {
        iDownstreamElement.Push(msg);
    }
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgMode* aMsg)
{
    OutputAggregatedAudio();
    iSupportsLatency = (aMsg->Info().LatencyMode() != Latency::NotSupported);
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgTrack* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgDrain* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgEncodedStream* aMsg)
{
    OutputAggregatedAudio();
    // raw PCM in a mode that manages latency is passed on as it arrives (:91-101)
    iAggregationDisabled = (iSupportsLatency && aMsg->StreamFormat() != MsgEncodedStream::Format::Encoded);
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgStreamInterrupted* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgHalt* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgFlush* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgWait* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgDecodedStream* aMsg)
{
    OutputAggregatedAudio();
    ASSERT(iDecodedAudio == nullptr);
    const DecodedStreamInfo& info = aMsg->StreamInfo();
    iChannels = info.NumChannels();
    iSampleRate = info.SampleRate();
    iBitDepth = info.BitDepth();
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgAudioPcm* aMsg)
{
    return TryAggregate(aMsg);
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgQuit* aMsg)
{
    OutputAggregatedAudio();
    return aMsg;
}

TBool DecodedAudioAggregator::AggregatorFull(TUint aBytes, TUint aJiffies)
{
    return (aBytes == DecodedAudio::kMaxBytes || aJiffies >= kMaxJiffies);
}

MsgAudioPcm* DecodedAudioAggregator::TryAggregate(MsgAudioPcm* aMsg)
{                                                        // DecodedAudioAggregator.cpp:137-186
    if (iAggregationDisabled) {
        return aMsg;
    }

    TUint msgJiffies = aMsg->Jiffies();
    const TUint jiffiesPerSample = Jiffies::PerSample(iSampleRate);
    const TUint msgBytes = Jiffies::ToBytes(msgJiffies, jiffiesPerSample, iChannels, iBitDepth); // jiffies might be modified here
    ASSERT(msgJiffies == aMsg->Jiffies());               // refuse to handle msgs not terminating on sample boundaries

    if (iDecodedAudio == nullptr) {
        if (AggregatorFull(msgBytes, msgJiffies)) {
            return aMsg;
        }
        iDecodedAudio = aMsg;
        iAggregatedJiffies = msgJiffies;
        return nullptr;
    }

    TUint aggregatedBytes = Jiffies::ToBytes(iAggregatedJiffies, jiffiesPerSample, iChannels, iBitDepth);
    if (aggregatedBytes + msgBytes <= kMaxBytes) {
        // Have byte capacity to add new data.
        iDecodedAudio->Aggregate(aMsg);
        iAggregatedJiffies += msgJiffies;
        aggregatedBytes = Jiffies::ToBytes(iAggregatedJiffies, jiffiesPerSample, iChannels, iBitDepth);
        if (AggregatorFull(aggregatedBytes, iAggregatedJiffies)) {
            MsgAudioPcm* msg = iDecodedAudio;
            iDecodedAudio = nullptr;
            iAggregatedJiffies = 0;
            return msg;
        }
        return nullptr;
    }
    // Lazy approach here - if new aMsg can't be appended, just return iDecodedAudio and set iDecodedAudio = aMsg.
    MsgAudioPcm* msg = iDecodedAudio;
    iDecodedAudio = aMsg;
    iAggregatedJiffies = msgJiffies;
    return msg;
}

void DecodedAudioAggregator::OutputAggregatedAudio()
{
    if (iDecodedAudio != nullptr) {
        iDownstreamElement.Push(iDecodedAudio);
        iDecodedAudio = nullptr;
    }
}

// ---------------------------------------------------------------------------------------------- CodecController (output side)
CodecController::CodecController(MsgFactory& aMsgFactory, IPipelineElementDownstream& aDownstreamElement, TUint aMaxOutputJiffies)
    : iMsgFactory(aMsgFactory)
    , iDownstreamElement(aDownstreamElement)
    , iMaxOutputJiffies(aMaxOutputJiffies)
    , iStreamId(0)
    , iChannels(0)
    , iSampleRate(0)
    , iBitDepth(0)
    , iMaxOutputSamples(0)
    , iMaxOutputBytes(0)
{
}

void CodecController::OutputDecodedStream(TUint aBitRate, TUint aBitDepth, TUint aSampleRate, TUint aNumChannels,
                                          const Brx& aCodecName, TUint64 aTrackLength, TUint64 aSampleStart, TBool aLossless)
{
    if (!Jiffies::IsValidSampleRate(aSampleRate)) {
        THROW(CodecStreamFeatureUnsupported);
    }
    DecodedStreamInfo info;
    info.iStreamId = ++iStreamId;
    info.iBitRate = aBitRate;
    info.iBitDepth = aBitDepth;
    info.iSampleRate = aSampleRate;
    info.iNumChannels = aNumChannels;
    info.iCodecName.assign((const char*)aCodecName.Ptr(), aCodecName.Bytes());
    info.iTrackLength = aTrackLength;
    info.iSampleStart = aSampleStart;
    info.iLossless = aLossless;
    info.iMultiroom = (aSampleRate > 192000) ? Multiroom::Forbidden : Multiroom::Allowed;       // :729-732
    // DoOutputDecodedStream, :760-797
    iChannels = aNumChannels;
    iSampleRate = aSampleRate;
    iBitDepth = aBitDepth;
    iMaxOutputSamples = Jiffies::ToSamples(iMaxOutputJiffies, iSampleRate);
    iMaxOutputBytes = (iMaxOutputSamples * iBitDepth * iChannels) / 8;
    iDownstreamElement.Push(iMsgFactory.CreateMsgDecodedStream(info));
}

TUint64 CodecController::OutputAudioPcm(const Brx& aData, TUint aChannels, TUint aSampleRate, TUint aBitDepth,
                                        AudioDataEndian aEndian, TUint64 aTrackOffset)
{
    ASSERT(aChannels == iChannels);
    ASSERT(aSampleRate == iSampleRate);
    ASSERT(aBitDepth == iBitDepth);
    if (aData.Bytes() == 0) {
        // allow for codecs which had a tiny bit of data which was later rounded down to 0 samples
        return 0;
    }
    const TUint64 offsetBefore = aTrackOffset;
    const TByte* p = aData.Ptr();
    TUint remaining = aData.Bytes();
    do {
        const TUint bytes = std::min(iMaxOutputBytes, remaining);
        MsgAudioPcm* audio = iMsgFactory.CreateMsgAudioPcm(Brn(p, bytes), aChannels, aSampleRate, aBitDepth, aEndian, aTrackOffset);
        const TUint64 jiffies = audio->Jiffies();                    // DoOutputAudio, :839-860 (no flush / seek pending here)
        iDownstreamElement.Push(audio);
        aTrackOffset += jiffies;
        p += bytes;
        remaining -= bytes;
    } while (remaining > 0);
    return aTrackOffset - offsetBefore;
}
