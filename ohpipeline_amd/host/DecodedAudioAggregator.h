// DecodedAudioAggregator.h -- the batch builder's first half: how big the messages are that reach
// the GPU path.  Host-side counterparts, with the reference's names and behaviour, of
//   CodecController::OutputDecodedStream / OutputAudioPcm   OpenHome/Media/Codec/CodecController.cpp:716-836
//       a codec's output leaves as MsgAudioPcm of at most iMaxOutputJiffies (whole samples), the track offset running on
//   DecodedAudioAggregator                                   OpenHome/Media/Pipeline/DecodedAudioAggregator.{h,cpp}
//       small MsgAudioPcm are joined until 5 ms or DecodedAudio::kMaxBytes are reached
// Neither interprets a PCM byte (the codec's endian travels in the DecodedAudio and is resolved by the device when the
// audio is read).  DSD is out of scope: MsgAudioDsd passes through untouched.
//
// Shape of this implementation: the aggregator holds at most one message; every message that is not PCM audio first
// releases it (one helper serves all of them), PCM audio is absorbed by Absorb(), which keeps a running byte count next to
// the jiffies (messages end on sample boundaries, so the two never drift apart).
#pragma once

#include "Msg.h"

namespace OpenHome {
namespace Media {

OH_EXCEPTION(CodecStreamFeatureUnsupported);

class DecodedAudioAggregator : public PipelineElement, public IPipelineElementDownstream {
public:
    static const TUint kMaxBytes = DecodedAudio::kMaxBytes;
    static const TUint kMaxMs = 5;                       // join until there is this much audio, or kMaxBytes; a whole message
                                                         // that still fits may take it over 5 ms (never over kMaxBytes)
    static const TUint kMaxJiffies = (Jiffies::kPerMs * kMaxMs) - Jiffies::kMaxJiffiesPerSample;
public:
    explicit DecodedAudioAggregator(IPipelineElementDownstream& aDownstreamElement);
    ~DecodedAudioAggregator();
public: // from IPipelineElementDownstream
    void Push(Msg* aMsg) override;
private: // IMsgProcessor: whatever is not PCM audio sends the held audio on ahead of itself
    Msg* ProcessMsg(MsgMode* aMsg) override;
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override;
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override { return Absorb(aMsg); }
    Msg* ProcessMsg(MsgTrack* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgDrain* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgHalt* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgFlush* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgWait* aMsg) override { return After(aMsg); }
    Msg* ProcessMsg(MsgQuit* aMsg) override { return After(aMsg); }
private:
    Msg* After(Msg* aMsg) { Release(); return aMsg; }    // the held audio goes downstream first
    MsgAudioPcm* Absorb(MsgAudioPcm* aMsg);
    void Hold(MsgAudioPcm* aMsg, TUint aJiffies, TUint aBytes) { iHeld = aMsg; iHeldJiffies = aJiffies; iHeldBytes = aBytes; }
    MsgAudioPcm* Take() { MsgAudioPcm* m = iHeld; iHeld = nullptr; iHeldJiffies = iHeldBytes = 0; return m; }
    void Release();
    static TBool Complete(TUint aBytes, TUint aJiffies) { return aBytes == kMaxBytes || aJiffies >= kMaxJiffies; }
private:
    IPipelineElementDownstream& iDownstreamElement;
    MsgAudioPcm* iHeld;                                  // the message being filled, or nullptr
    TUint iHeldJiffies, iHeldBytes;
    TUint iChannels, iSampleRate, iBitDepth;             // of the current stream
    TBool iLatencyManaged;                               // the mode controls its own latency ...
    TBool iPassThrough;                                  // ... and the stream is raw PCM: hand audio on as it comes
};

/** The output side of CodecController (the part a codec calls through ICodecController). */
class CodecController {
public:
    CodecController(MsgFactory& aMsgFactory, IPipelineElementDownstream& aDownstreamElement, TUint aMaxOutputJiffies);
    /** CodecController.cpp:716-730, 760-797: announces the stream and sizes the output pieces for it. */
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
    TUint iStreamId, iChannels, iSampleRate, iBitDepth, iMaxOutputBytes;
};

} // namespace Media
} // namespace OpenHome
