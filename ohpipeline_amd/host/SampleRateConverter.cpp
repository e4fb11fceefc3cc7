// SampleRateConverter.cpp -- see SampleRateConverter.h.
#include "SampleRateConverter.h"

#include <cstring>

#include "../../include/ohgpu.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- SampleRateConverterStream
SampleRateConverterStream::SampleRateConverterStream(MsgFactory& aFactory, TUint aRateIn, TUint aRateOut, TUint aChannels,
                                                     TUint aBitDepth, AudioDataEndian aEndian, TUint aTapsPerPhase,
                                                     double aBeta, double aPassHz)
    : iFactory(aFactory)
    , iFilter(nullptr)
    , iL(0), iM(0), iT(aTapsPerPhase)
    , iChannels(aChannels), iBitDepth(aBitDepth), iEndian(aEndian)
    , iFrame0(0), iFrames(0)
{
    uint32_t L = 0, M = 0;
    int err = ohgpu_src_design(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz, nullptr, 0, &L, &M);
    ASSERT(err == OHGPU_OK);
    std::vector<int32_t> coef((size_t)L * aTapsPerPhase);
    err = ohgpu_src_design(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz, coef.data(), coef.size(), &L, &M);
    ASSERT(err == OHGPU_OK);
    err = ohgpu_src_create(iFactory.Gpu(), L, M, aTapsPerPhase, coef.data(), &iFilter);
    ASSERT(err == OHGPU_OK);
    iL = L;
    iM = M;
}

SampleRateConverterStream::~SampleRateConverterStream()
{
    ohgpu_src_destroy(iFactory.Gpu(), iFilter);
}

TUint64 SampleRateConverterStream::Append(const TByte* aData, TUint aBytes)
{
    const TUint frameBytes = iChannels * (iBitDepth / 8);
    ASSERT(aBytes % frameBytes == 0);
    // keep what later messages can still need: two seconds' worth is far beyond any pipeline's buffering
    const TUint64 keepFrames = 2 * 192000;
    if (iFrames > 2 * keepFrames) {
        const TUint64 drop = iFrames - keepFrames;
        iHistory.erase(iHistory.begin(), iHistory.begin() + (size_t)(drop * frameBytes));
        iFrame0 += drop;
        iFrames -= drop;
    }
    iHistory.insert(iHistory.end(), aData, aData + aBytes);
    iFrames += aBytes / frameBytes;
    return ohgpu_src_out_frames(iL, iM, iFrame0 + iFrames);
}

void SampleRateConverterStream::DescribeWindow(TUint64 aOut0, TUint aCount, ohgpu_src_msg_desc& aDesc) const
{
    (void)aOut0; (void)aCount;
    aDesc.src_offset = 0;
    aDesc.src_frame0 = iFrame0;
    aDesc.src_frames = iFrames;
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
