// Elements.cpp -- see Elements.h.
#include "Elements.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- Ramper (Ramper.cpp:10-152)
const TUint Ramper::kSupportedMsgTypes =
    eMode | eTrack | eDrain | eDelay | eEncodedStream | eMetatext | eStreamInterrupted | eHalt | eFlush | eWait |
    eDecodedStream | eAudioPcm | eAudioDsd | eSilence | eQuit;

Ramper::Ramper(IPipelineElementUpstream& aUpstreamElement, TUint aRampJiffiesLong, TUint aRampJiffiesShort)
    : PipelineElement(kSupportedMsgTypes)
    , iUpstreamElement(aUpstreamElement)
    , iStreamId(0xffffffff)
    , iRamping(false)
    , iRampJiffiesLong(aRampJiffiesLong), iRampJiffiesShort(aRampJiffiesShort)
    , iRampJiffies(aRampJiffiesLong), iRemainingRampSize(0), iCurrentRampValue(Ramp::kMin)
{
}

Msg* Ramper::Pull()
{
    Msg* msg;
    if (!iQueue.empty()) {
        msg = iQueue.front();
        iQueue.pop_front();
    }
    else {
        msg = iUpstreamElement.Pull();
    }
    msg = msg->Process(*this);
    ASSERT(msg != nullptr);
    return msg;
}

Msg* Ramper::ProcessMsg(MsgMode* aMsg)
{
    iRampJiffies = aMsg->Info().RampPauseResumeLong() ? iRampJiffiesLong : iRampJiffiesShort;
    return aMsg;
}

Msg* Ramper::ProcessMsg(MsgHalt* aMsg)
{
    iRamping = false;
    return aMsg;
}

Msg* Ramper::ProcessMsg(MsgDecodedStream* aMsg)
{
    const DecodedStreamInfo& info = aMsg->StreamInfo();
    if (IsRampApplicable(info)) {
        iRamping = true;
        iCurrentRampValue = Ramp::kMin;
        iRemainingRampSize = iRampJiffies;
    }
    else {
        iRamping = false;
        iCurrentRampValue = Ramp::kMax;
        iRemainingRampSize = 0;
    }
    iStreamId = info.StreamId();
    return aMsg;
}

Msg* Ramper::ProcessMsg(MsgAudioPcm* aMsg)
{
    if (iRamping) {
        if (aMsg->Jiffies() > iRemainingRampSize) {            // only the ramped part carries a ramp (Ramper.cpp:117-122)
            iQueue.push_back(aMsg->Split(iRemainingRampSize));
        }
        MsgAudio* split = nullptr;
        iCurrentRampValue = aMsg->SetRamp(iCurrentRampValue, iRemainingRampSize, Ramp::EUp, split);
        if (split != nullptr) {
            iQueue.push_front(split);
        }
        if (iRemainingRampSize == 0 || iCurrentRampValue == Ramp::kMax) {
            iRamping = false;
        }
    }
    return aMsg;
}

Msg* Ramper::ProcessMsg(MsgSilence* aMsg)
{
    iRamping = false;
    iCurrentRampValue = Ramp::kMax;
    iRemainingRampSize = 0;
    return aMsg;
}

TBool Ramper::IsRampApplicable(const DecodedStreamInfo& aInfo)
{
    if (aInfo.Live()) {
        return true;
    }
    const TBool newStream = (aInfo.StreamId() != iStreamId);
    return newStream && aInfo.SampleStart() > 0;                // a stream picked up mid-track starts with a ramp
}

// ---------------------------------------------------------------- PreDriver (PreDriver.cpp:17-133)
const TUint PreDriver::kSupportedMsgTypes = eMode | eDrain | eStreamInterrupted | eHalt | eDecodedStream | eAudioPcm | eSilence | eQuit;

PreDriver::PreDriver(IPipelineElementUpstream& aUpstreamElement)
    : PipelineElement(kSupportedMsgTypes)
    , iUpstreamElement(aUpstreamElement)
    , iSampleRate(0), iBitDepth(0), iNumChannels(0)
{
}

Msg* PreDriver::Pull()
{
    Msg* msg;
    do {
        msg = iUpstreamElement.Pull();
        msg = msg->Process(*this);
    } while (msg == nullptr);
    return msg;
}

Msg* PreDriver::ProcessMsg(MsgDecodedStream* aMsg)
{
    const DecodedStreamInfo& info = aMsg->StreamInfo();
    if (info.SampleRate() == iSampleRate && info.BitDepth() == iBitDepth && info.NumChannels() == iNumChannels) {
        aMsg->RemoveRef();                                       // no change in format: the driver need not know
        return nullptr;
    }
    iSampleRate = info.SampleRate();
    iBitDepth = info.BitDepth();
    iNumChannels = info.NumChannels();
    return aMsg;
}

Msg* PreDriver::ProcessMsg(MsgAudioPcm* aMsg)
{
    return aMsg->CreatePlayable();
}

Msg* PreDriver::ProcessMsg(MsgSilence* aMsg)
{
    return aMsg->CreatePlayable();
}

} // namespace Media
} // namespace OpenHome
