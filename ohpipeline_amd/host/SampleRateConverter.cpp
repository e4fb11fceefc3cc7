// SampleRateConverter.cpp -- see SampleRateConverter.h.
#include "SampleRateConverter.h"

#include <algorithm>
#include <cstring>

#include "../../include/ohgpu.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- SampleRateConverterStream
SampleRateConverterStream::SampleRateConverterStream(MsgFactory& aFactory, TUint aRateIn, TUint aRateOut, TUint aChannels,
                                                     TUint aBitDepth, AudioDataEndian aEndian, TUint aTapsPerPhase,
                                                     double aBeta, double aPassHz, TUint aHistoryMs)
    : SampleRateConverterStream(aFactory.SharedFilter(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz), aRateIn, aChannels, aBitDepth,
                                aEndian, aHistoryMs)
{
}

SampleRateConverterStream::SampleRateConverterStream(const SrcFilter& aFilter, TUint aRateIn, TUint aChannels, TUint aBitDepth,
                                                     AudioDataEndian aEndian, TUint aHistoryMs)
    : iFilter(aFilter)
    , iChannels(aChannels), iBitDepth(aBitDepth), iFrameBytes(aChannels * (aBitDepth / 8))
    , iEndian(aEndian)
    , iCapacity(0), iFrames(0)
{
    // what a window may reach back over: the history asked for, and never less than a filter length and two maximal messages
    iCapacity = (TUint64)aRateIn * aHistoryMs / 1000;
    const TUint64 least = (TUint64)iFilter.T + 2 * (DecodedAudio::kMaxBytes / iFrameBytes + 1);
    if (iCapacity < least) iCapacity = least;
    iRing.resize((size_t)(iCapacity * iFrameBytes));
}

TUint64 SampleRateConverterStream::Append(const TByte* aData, TUint aBytes)
{
    ASSERT(aBytes % iFrameBytes == 0);
    TUint64 frames = aBytes / iFrameBytes;
    ASSERT(frames <= iCapacity);
    std::lock_guard<std::mutex> hold(iLock);
    const TUint64 at = iFrames % iCapacity;
    const TUint64 head = std::min(frames, iCapacity - at);            // up to the ring's end, the rest from its start
    memcpy(&iRing[(size_t)(at * iFrameBytes)], aData, (size_t)(head * iFrameBytes));
    if (frames > head) memcpy(&iRing[0], aData + head * iFrameBytes, (size_t)((frames - head) * iFrameBytes));
    iFrames += frames;
    return ohgpu_src_out_frames(iFilter.L, iFilter.M, iFrames);
}

TUint64 SampleRateConverterStream::InputFrames() const
{
    std::lock_guard<std::mutex> hold(iLock);
    return iFrames;
}

void SampleRateConverterStream::Window(TUint64 aOut0, TUint aCount, TUint64& aFirst, TUint& aFrames) const
{
    ASSERT(aCount != 0);
    const TUint64 newestOfFirst = aOut0 * iFilter.M / iFilter.L;                      // n0 of the run's first output
    const TUint64 newestOfLast = (aOut0 + aCount - 1) * iFilter.M / iFilter.L;        // ... of its last
    aFirst = newestOfFirst >= iFilter.T - 1 ? newestOfFirst - (iFilter.T - 1) : 0;
    aFrames = (TUint)(newestOfLast - aFirst + 1);
}

void SampleRateConverterStream::CopyFrames(TUint64 aFirst, TUint aFrames, TByte* aDst) const
{
    std::lock_guard<std::mutex> hold(iLock);
    ASSERT(aFirst + aFrames <= iFrames);                               // the output exists only once its input has arrived
    ASSERT(aFirst + iCapacity >= iFrames);                             // ... and the ring must not have gone round over it
    const TUint64 at = aFirst % iCapacity;
    const TUint64 head = std::min<TUint64>(aFrames, iCapacity - at);
    memcpy(aDst, &iRing[(size_t)(at * iFrameBytes)], (size_t)(head * iFrameBytes));
    if (aFrames > head) memcpy(aDst + head * iFrameBytes, &iRing[0], (size_t)((aFrames - head) * iFrameBytes));
}

// ---------------------------------------------------------------- SampleRateConverter
const TUint SampleRateConverter::kSupportedMsgTypes =
    eMode | eTrack | eDrain | eDelay | eEncodedStream | eMetatext | eStreamInterrupted | eHalt | eFlush | eWait |
    eDecodedStream | eAudioPcm | eSilence | eQuit;

SampleRateConverter::SampleRateConverter(MsgFactory& aFactory, IPipelineElementUpstream& aUpstreamElement, TUint aOutputRate,
                                         TUint aTapsPerPhase, double aBeta, double aPassHz)
    : PipelineElement(kSupportedMsgTypes)
    , iFactory(aFactory)
    , iUpstreamElement(aUpstreamElement)
    , iOutputRate(aOutputRate), iTapsPerPhase(aTapsPerPhase)
    , iBeta(aBeta), iPassHz(aPassHz)
    , iOutFrames(0)
    , iTrackOffset(0)
{
    (void)Jiffies::PerSample(aOutputRate);       // throws SampleRateInvalid for a rate the pipeline cannot express
}

Msg* SampleRateConverter::Pull()
{
    Msg* msg;
    do {                                          // input that does not complete an output frame yields nothing yet
        msg = iUpstreamElement.Pull();
        msg = msg->Process(*this);
    } while (msg == nullptr);
    return msg;
}

Msg* SampleRateConverter::ProcessMsg(MsgDecodedStream* aMsg)
{
    iInfo = aMsg->StreamInfo();
    iStream.reset();
    iOutFrames = 0;
    iTrackOffset = 0;
    if (iInfo.Format() != AudioFormat::Pcm || iInfo.SampleRate() == iOutputRate) {
        return aMsg;                              // nothing to convert
    }
    DecodedStreamInfo out = iInfo;
    out.iSampleRate = iOutputRate;
    out.iBitDepth = 24;                           // the converter works, and delivers, in the S24 domain
    out.iBitRate = iOutputRate * 24 * iInfo.NumChannels();
    aMsg->RemoveRef();
    return iFactory.CreateMsgDecodedStream(out);
}

Msg* SampleRateConverter::ProcessMsg(MsgAudioPcm* aMsg)
{
    if (iInfo.SampleRate() == iOutputRate || iInfo.SampleRate() == 0) {
        return aMsg;
    }
    ASSERT(aMsg->iAudioData != nullptr);                              // input must be real audio
    ASSERT(!aMsg->Ramp().IsEnabled());                                // ramps are set downstream of the converter
    const TUint jps = Jiffies::PerSample(aMsg->SampleRate());
    const TUint frameBytes = aMsg->NumChannels() * (aMsg->BitDepth() / 8);
    if (iStream == nullptr) {
        iStream = std::make_shared<SampleRateConverterStream>(iFactory, aMsg->SampleRate(), iOutputRate, aMsg->NumChannels(),
                                                              aMsg->BitDepth(), aMsg->iAudioData->Endian(), iTapsPerPhase,
                                                              iBeta, iPassHz);
    }
    const TUint firstFrame = aMsg->iOffset / jps;
    const TUint frames = aMsg->iSize / jps;
    const TUint64 outAvailable = iStream->Append(aMsg->iAudioData->Ptr(firstFrame * frameBytes), frames * frameBytes);
    aMsg->RemoveRef();
    if (outAvailable == iOutFrames) {
        return nullptr;
    }
    const TUint n = (TUint)(outAvailable - iOutFrames);
    const TUint jpsOut = Jiffies::PerSample(iOutputRate);
    MsgAudioPcm* out = new MsgAudioPcm(iFactory, nullptr, iOutputRate, 24, iInfo.NumChannels(), iTrackOffset);
    out->iResampled = iStream;
    out->iResampledFrame0 = iOutFrames;
    out->iSize = n * jpsOut;
    iOutFrames = outAvailable;
    iTrackOffset += (TUint64)n * jpsOut;
    return out;
}

Msg* SampleRateConverter::ProcessMsg(MsgSilence* aMsg)
{
    if (iInfo.SampleRate() == iOutputRate || iInfo.SampleRate() == 0) {
        return aMsg;
    }
    TUint jiffies = aMsg->Jiffies();
    aMsg->RemoveRef();
    return iFactory.CreateMsgSilence(jiffies, iOutputRate, 24, iInfo.NumChannels());
}

Msg* SampleRateConverter::ProcessMsg(MsgHalt* aMsg)
{
    return aMsg;
}

} // namespace Media
} // namespace OpenHome
