// Elements.h -- the two reference elements that bracket the data plane, mirrored on the host adapter:
//   Ramper     OpenHome/Media/Pipeline/Ramper.{h,cpp}     ramps a stream up when it starts live / mid-track
//   PreDriver  OpenHome/Media/Pipeline/PreDriver.{h,cpp}  last element: MsgAudioPcm/MsgSilence -> MsgPlayable
// Both only manipulate metadata (split, SetRamp, CreatePlayable); the bytes move when the driver reads the playables.
#pragma once

#include <deque>

#include "Msg.h"

namespace OpenHome {
namespace Media {

class Ramper : public PipelineElement, public IPipelineElementUpstream {
    static const TUint kSupportedMsgTypes;
public:
    Ramper(IPipelineElementUpstream& aUpstreamElement, TUint aRampJiffiesLong, TUint aRampJiffiesShort);
public: // from IPipelineElementUpstream
    Msg* Pull() override;
private: // IMsgProcessor
    Msg* ProcessMsg(MsgMode* aMsg) override;
    Msg* ProcessMsg(MsgHalt* aMsg) override;
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgSilence* aMsg) override;
private:
    TBool IsRampApplicable(const DecodedStreamInfo& aInfo);
private:
    IPipelineElementUpstream& iUpstreamElement;
    TUint iStreamId;
    TBool iRamping;
    const TUint iRampJiffiesLong, iRampJiffiesShort;
    TUint iRampJiffies, iRemainingRampSize, iCurrentRampValue;
    std::deque<Msg*> iQueue;
};

class PreDriver : public PipelineElement, public IPipelineElementUpstream {
    static const TUint kSupportedMsgTypes;
public:
    explicit PreDriver(IPipelineElementUpstream& aUpstreamElement);
public: // from IPipelineElementUpstream
    Msg* Pull() override;
private: // IMsgProcessor
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgSilence* aMsg) override;
private:
    IPipelineElementUpstream& iUpstreamElement;
    TUint iSampleRate, iBitDepth, iNumChannels;
};

} // namespace Media
} // namespace OpenHome
