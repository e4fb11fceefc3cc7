// SampleRateConverter.h -- the pipeline element BASELINE.json's north star names.  The reference has NO
// sample-rate converter: it rejects unsupported rates (Codec/CodecController.cpp:724-726) and pulls
// the DAC clock instead (ClockPuller.h:17-34).  This element is therefore new; it follows the reference's element
// conventions (PipelineElement + IPipelineElementUpstream, Pull() = upstream.Pull()->Process(*this),
// Ramper.cpp:46-58) and the reference's lazy data plane: it never touches a PCM byte.  It keeps each stream's input
// history and hands downstream MsgAudioPcm messages at the output rate whose audio is *virtual* -- "output frames
// [m0, m0+n) of this stream" -- so that downstream elements can split them and set ramps on them as usual, and the
// fused resample -> ramp -> pack kernel produces the bytes when the driver finally reads the playable.
// Specification of the filter: DESIGN.md section 4.
#pragma once

#include <memory>
#include <vector>

#include "Msg.h"

struct ohgpu_src_msg_desc;

namespace OpenHome {
namespace Media {

/** Input history + filter of one rate-converted stream (shared by the output messages that refer to it). */
class SampleRateConverterStream {
public:
    SampleRateConverterStream(MsgFactory& aFactory, TUint aRateIn, TUint aRateOut, TUint aChannels, TUint aBitDepth,
                              AudioDataEndian aEndian, TUint aTapsPerPhase, double aBeta, double aPassHz);
    ~SampleRateConverterStream();
    /** Appends input frames; returns how many output frames exist now (ceil(in*L/M)). */
    TUint64 Append(const TByte* aData, TUint aBytes);
    TUint64 InputFrames() const { return iFrame0 + iFrames; }
    /** Fills src_offset / src_frame0 / src_frames for the window output frames [aOut0, aOut0+aCount) need. */
    void DescribeWindow(TUint64 aOut0, TUint aCount, ohgpu_src_msg_desc& aDesc) const;
    const ohgpu_src* Filter() const { return iFilter; }
    const TByte* HistoryPtr() const { return iHistory.data(); }
    TUint64 HistoryBytes() const { return iHistory.size(); }
    TUint SourceBitDepth() const { return iBitDepth; }
    AudioDataEndian SourceEndian() const { return iEndian; }
    TUint L() const { return iL; }
    TUint M() const { return iM; }
private:
    MsgFactory& iFactory;
    ohgpu_src* iFilter;
    TUint iL, iM, iT;
    TUint iChannels, iBitDepth;
    AudioDataEndian iEndian;
    std::vector<TByte> iHistory;     // frames [iFrame0, iFrame0 + iFrames)
    TUint64 iFrame0, iFrames;
};

class SampleRateConverter : public PipelineElement, public IPipelineElementUpstream {
    static const TUint kSupportedMsgTypes;
public:
    SampleRateConverter(MsgFactory& aFactory, IPipelineElementUpstream& aUpstreamElement, TUint aOutputRate,
                        TUint aTapsPerPhase = 32, double aBeta = 9.0, double aPassHz = 20000.0);
public: // from IPipelineElementUpstream
    Msg* Pull() override;
private: // IMsgProcessor
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgSilence* aMsg) override;
    Msg* ProcessMsg(MsgHalt* aMsg) override;
private:
    MsgFactory& iFactory;
    IPipelineElementUpstream& iUpstreamElement;
    const TUint iOutputRate, iTapsPerPhase;
    const double iBeta, iPassHz;
    DecodedStreamInfo iInfo;
    std::shared_ptr<SampleRateConverterStream> iStream;   // null while the stream already runs at the output rate
    TUint64 iOutFrames;                                   // output frames handed downstream so far
    TUint64 iTrackOffset;                                 // jiffies, at the output rate
};

} // namespace Media
} // namespace OpenHome
