// DecodedAudioAggregator.h -- the batch builder's first half: how big the messages are that reach
// the GPU path.  Host-side mirrors of
//   CodecController::OutputDecodedStream / OutputAudioPcm   OpenHome/Media/Codec/CodecController.cpp:716-836
//       a codec's output is cut into MsgAudioPcm of at most iMaxOutputJiffies (whole samples), track offset running
//   DecodedAudioAggregator                                   OpenHome/Media/Pipeline/DecodedAudioAggregator.{h,cpp}
//       small MsgAudioPcm are aggregated until 5 ms or DecodedAudio::kMaxBytes are reached
// Both only create / aggregate messages; no PCM byte is interpreted (the codec's endian travels in the DecodedAudio and
// is resolved by the device when the audio is read).  DSD is out of scope: MsgAudioDsd passes through untouched.
#pragma once

#include "Msg.h"

namespace OpenHome {
namespace Media {

OH_EXCEPTION(CodecStreamFeatureUnsupported);

class DecodedAudioAggregator : public PipelineElement, public IPipelineElementDownstream {
public:
    static const TUint kMaxBytes = DecodedAudio::kMaxBytes;
    static const TUint kMaxMs = 5;  // buffer MsgAudioPcm until we have this many ms (unless we hit DecodedAudio::kMaxBytes
                                    // first); may be violated if a MsgAudioPcm can be added without chopping it
    static const TUint kMaxJiffies = (Jiffies::kPerMs * kMaxMs) - Jiffies::kMaxJiffiesPerSample;
    static const TUint kSupportedMsgTypes;
public:
    explicit DecodedAudioAggregator(IPipelineElementDownstream& aDownstreamElement);
    ~DecodedAudioAggregator();
public: // from IPipelineElementDownstream
    void Push(Msg* aMsg) override;
private: // IMsgProcessor
    Msg* ProcessMsg(MsgMode* aMsg) override;
    Msg* ProcessMsg(MsgTrack* aMsg) override;
    Msg* ProcessMsg(MsgDrain* aMsg) override;
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override;
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override;
    Msg* ProcessMsg(MsgHalt* aMsg) override;
    Msg* ProcessMsg(MsgFlush* aMsg) override;
    Msg* ProcessMsg(MsgWait* aMsg) override;
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgQuit* aMsg) override;
private:
    static TBool AggregatorFull(TUint aBytes, TUint aJiffies);
    MsgAudioPcm* TryAggregate(MsgAudioPcm* aMsg);
    void OutputAggregatedAudio();
private:
    IPipelineElementDownstream& iDownstreamElement;
    MsgAudioPcm* iDecodedAudio;
    TUint iChannels, iSampleRate, iBitDepth;
    TBool iSupportsLatency, iAggregationDisabled;
    TUint iAggregatedJiffies;
};

/** The output side of CodecController (the part a codec calls through ICodecController). */
class CodecController {
public:
    CodecController(MsgFactory& aMsgFactory, IPipelineElementDownstream& aDownstreamElement, TUint aMaxOutputJiffies);
    /** CodecController.cpp:716-730, 760-797: announces the stream and sizes the output chunks for it. */
    void OutputDecodedStream(TUint aBitRate, TUint aBitDepth, TUint aSampleRate, TUint aNumChannels, const Brx& aCodecName,
                             TUint64 aTrackLength, TUint64 aSampleStart, TBool aLossless);
    /** CodecController.cpp:799-826: returns the jiffies output; asserts the format is the announced one. */
    TUint64 OutputAudioPcm(const Brx& aData, TUint aChannels, TUint aSampleRate, TUint aBitDepth, AudioDataEndian aEndian,
                           TUint64 aTrackOffset);
    TUint MaxOutputBytes() const { return iMaxOutputBytes; }
private:
    MsgFactory& iMsgFactory;
    IPipelineElementDownstream& iDownstreamElement;
    const TUint iMaxOutputJiffies;
    TUint iStreamId, iChannels, iSampleRate, iBitDepth, iMaxOutputSamples, iMaxOutputBytes;
};

} // namespace Media
} // namespace OpenHome
