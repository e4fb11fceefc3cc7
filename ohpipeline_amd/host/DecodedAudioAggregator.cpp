// DecodedAudioAggregator.cpp -- see DecodedAudioAggregator.h.  File:line comments are relative to the reference tree.
#include "DecodedAudioAggregator.h"

#include <algorithm>

using namespace OpenHome;
using namespace OpenHome::Media;

// ---------------------------------------------------------------------------------------------- DecodedAudioAggregator
DecodedAudioAggregator::DecodedAudioAggregator(IPipelineElementDownstream& aDownstreamElement)
    : PipelineElement(eMode | eTrack | eDrain | eDelay | eEncodedStream | eMetatext | eStreamInterrupted | eHalt | eFlush | eWait
                      | eDecodedStream | eAudioPcm | eAudioDsd | eQuit)                  // DecodedAudioAggregator.cpp:13-26
    , iDownstreamElement(aDownstreamElement)
    , iHeld(nullptr), iHeldJiffies(0), iHeldBytes(0)
    , iChannels(0), iSampleRate(0), iBitDepth(0)
    , iLatencyManaged(false), iPassThrough(false)
{
}

DecodedAudioAggregator::~DecodedAudioAggregator()
{
    if (iHeld != nullptr) {
        iHeld->RemoveRef();
    }
}

void DecodedAudioAggregator::Push(Msg* aMsg)
{
    ASSERT(aMsg != nullptr);
    if (Msg* out = aMsg->Process(*this)) {
        iDownstreamElement.Push(out);
    }
}

void DecodedAudioAggregator::Release()
{
    if (iHeld != nullptr) {
        iDownstreamElement.Push(Take());
    }
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgMode* aMsg)
{
    Release();
    iLatencyManaged = aMsg->Info().LatencyMode() != Latency::NotSupported;               // :56
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgEncodedStream* aMsg)
{
    Release();
    iPassThrough = iLatencyManaged && aMsg->StreamFormat() != MsgEncodedStream::Format::Encoded;   // :91-101
    return aMsg;
}

Msg* DecodedAudioAggregator::ProcessMsg(MsgDecodedStream* aMsg)
{
    Release();
    const DecodedStreamInfo& stream = aMsg->StreamInfo();
    iChannels = stream.NumChannels();
    iSampleRate = stream.SampleRate();
    iBitDepth = stream.BitDepth();
    return aMsg;
}

// DecodedAudioAggregator::TryAggregate, :137-186.  Returns what should go downstream now (possibly aMsg itself), or nullptr.
MsgAudioPcm* DecodedAudioAggregator::Absorb(MsgAudioPcm* aMsg)
{
    if (iPassThrough) {
        return aMsg;
    }
    TUint jiffies = aMsg->Jiffies();
    const TUint bytes = Jiffies::ToBytes(jiffies, Jiffies::PerSample(iSampleRate), iChannels, iBitDepth);   // rounds jiffies down to a sample
    ASSERT(jiffies == aMsg->Jiffies());                  // refuse messages that do not end on a sample boundary (:147)

    if (iHeld == nullptr) {
        if (Complete(bytes, jiffies)) {
            return aMsg;                                 // already as large as a message gets
        }
        Hold(aMsg, jiffies, bytes);
        return nullptr;
    }
    if (iHeldBytes + bytes > kMaxBytes) {
        // no room: what is held goes on, the newcomer starts the next one (the reference does not chop messages either)
        MsgAudioPcm* full = iHeld;
        Hold(aMsg, jiffies, bytes);
        return full;
    }
    iHeld->Aggregate(aMsg);
    iHeldJiffies += jiffies;
    iHeldBytes += bytes;
    return Complete(iHeldBytes, iHeldJiffies) ? Take() : nullptr;
}

// ---------------------------------------------------------------------------------------------- CodecController (output side)
CodecController::CodecController(MsgFactory& aMsgFactory, IPipelineElementDownstream& aDownstreamElement, TUint aMaxOutputJiffies)
    : iMsgFactory(aMsgFactory)
    , iDownstreamElement(aDownstreamElement)
    , iMaxOutputJiffies(aMaxOutputJiffies)
    , iStreamId(0), iChannels(0), iSampleRate(0), iBitDepth(0), iMaxOutputBytes(0)
{
}

void CodecController::OutputDecodedStream(TUint aBitRate, TUint aBitDepth, TUint aSampleRate, TUint aNumChannels,
                                          const Brx& aCodecName, TUint64 aTrackLength, TUint64 aSampleStart, TBool aLossless)
{
    if (!Jiffies::IsValidSampleRate(aSampleRate)) {
        THROW(CodecStreamFeatureUnsupported);            // :721-723
    }
    iChannels = aNumChannels;
    iSampleRate = aSampleRate;
    iBitDepth = aBitDepth;
    // the largest piece OutputAudioPcm hands on: whole samples within iMaxOutputJiffies (:792-793)
    iMaxOutputBytes = Jiffies::ToSamples(iMaxOutputJiffies, aSampleRate) * aBitDepth * aNumChannels / 8;

    DecodedStreamInfo info;
    info.iStreamId = ++iStreamId;
    info.iBitRate = aBitRate;
    info.iBitDepth = aBitDepth;
    info.iSampleRate = aSampleRate;
    info.iNumChannels = aNumChannels;
    info.iCodecName.assign(reinterpret_cast<const char*>(aCodecName.Ptr()), aCodecName.Bytes());
    info.iTrackLength = aTrackLength;
    info.iSampleStart = aSampleStart;
    info.iLossless = aLossless;
    info.iMultiroom = aSampleRate > 192000 ? Multiroom::Forbidden : Multiroom::Allowed;   // :729-732
    iDownstreamElement.Push(iMsgFactory.CreateMsgDecodedStream(info));
}

TUint64 CodecController::OutputAudioPcm(const Brx& aData, TUint aChannels, TUint aSampleRate, TUint aBitDepth,
                                        AudioDataEndian aEndian, TUint64 aTrackOffset)
{
    ASSERT(aChannels == iChannels && aSampleRate == iSampleRate && aBitDepth == iBitDepth);   // :801-803
    TUint64 jiffiesOut = 0;
    // (an empty buffer is allowed: a codec's last crumbs may have rounded down to no samples, :805-808)
    for (TUint done = 0; done < aData.Bytes(); ) {
        const TUint piece = std::min(iMaxOutputBytes, aData.Bytes() - done);
        MsgAudioPcm* audio = iMsgFactory.CreateMsgAudioPcm(Brn(aData.Ptr() + done, piece), aChannels, aSampleRate, aBitDepth,
                                                           aEndian, aTrackOffset + jiffiesOut);
        jiffiesOut += audio->Jiffies();                  // DoOutputAudio, :839-860 (no flush or seek pending here)
        iDownstreamElement.Push(audio);
        done += piece;
    }
    return jiffiesOut;
}
