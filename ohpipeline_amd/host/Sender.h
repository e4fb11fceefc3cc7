// Sender.h -- host-side mirror of the Songcast sender's data path:
//   Sender           OpenHome/Av/Songcast/Sender.{h,cpp}      pipeline element: 5 ms packetisation of the audio pushed at it
//   OhmSenderDriver  OpenHome/Av/Songcast/OhmSender.{h,cpp}   per-stream frame counters and the audio frame itself
// Same names, argument meaning and error behaviour as the reference for the methods that shape the datagrams.  What is
// left out is everything around them that is not on the data path: the UPnP service, zone handling, configuration values,
// the resend history, timestamping and the socket -- a datagram is handed to an IOhmDatagramSink where the reference calls
// iSocket.Send (OhmSender.cpp:472).
// What differs, on purpose (MI355X-first): no PCM byte is touched on the CPU.  SendPendingAudio records WHAT a packet
// consists of (playables + header fields); OhmFrameBatch turns any number of recorded packets, of any number of senders,
// into datagrams in one pass on the device (include/ohgpu.h, ohgpu_ohm_*) and delivers them in the order they were made.
// By default a Sender runs its batch after every packet, like the reference; SetBatching() lets a multi-stream host
// collect more before it pays for a launch.
#pragma once

#include <string>
#include <vector>

#include "Msg.h"

namespace OpenHome {
namespace Av {

class IOhmDatagramSink {                                 // where OhmSenderDriver::SendAudio calls iSocket.Send (OhmSender.cpp:472)
public:
    virtual ~IOhmDatagramSink() {}
    virtual void Send(const Brx& aDatagram) = 0;
};

/** One OhmSenderDriver::SendAudio recorded for the device. */
struct OhmFrameWork {
    std::vector<Media::MsgPlayable*> playables;          // what Sender::SendPendingAudio read, in order; owned until the batch ran
    TBool halt = false, lossless = false;
    TUint frame = 0, mediaLatency = 0;
    TUint64 sampleStart = 0, samplesTotal = 0;
    TUint sampleRate = 0, bitRate = 0, numChannels = 0, bitDepth = 0;   // the pipeline's format (the wire's is derived from it)
    std::string codecName;
    IOhmDatagramSink* sink = nullptr;
};

/** Any number of recorded frames -> datagrams, one pass on the device. */
class OhmFrameBatch {
public:
    explicit OhmFrameBatch(Media::MsgFactory& aFactory);
    ~OhmFrameBatch();
    void Add(OhmFrameWork&& aWork);                      // takes over the playables' references
    void Run();                                          // launches, waits, delivers in order, releases
    TUint Count() const { return (TUint)iFrames.size(); }
private:
    Media::MsgFactory& iFactory;
    std::vector<OhmFrameWork> iFrames;
};

class OhmSenderDriver {                                  // OhmSender.h:70-128
public:
    OhmSenderDriver(OhmFrameBatch& aBatch, IOhmDatagramSink& aSink);
    void SetAudioFormat(TUint aSampleRate, TUint aBitRate, TUint aChannels, TUint aBitDepth, TBool aLossless,
                        const Brx& aCodecName, TUint64 aSampleStart);                   // OhmSender.cpp:325-344
    /** OhmSender.cpp:418-480; aPlayables are the reads that filled the OhmMsgAudio, aSamples what they amount to.
     *  aNumChannels / aBitDepth describe the playables (SetAudioFormat was told the wire's). */
    void SendAudio(std::vector<Media::MsgPlayable*>& aPlayables, TUint aSamples, TUint aNumChannels, TUint aBitDepth, TBool aHalt);
    void StreamInterrupted();                                                             // :482-488
    void SetEnabled(TBool aValue) { Gate(aValue, iActive); }                              // :490-505
    void SetActive(TBool aValue) { Gate(iEnabled, aValue); }                              // :507-525
    void SetLatency(TUint aValue);                                                        // :544-549
    void SetTrackPosition(TUint64 aSamplesTotal, TUint64 aSampleStart);                   // :551-556
    TUint Frame() const { return iFrame; }
    TUint64 SampleStart() const { return iSampleStart; }
private:
    void UpdateLatencyOhm() { iLatencyOhm = iLatencyMs * iTimestampMultiplier / 1000; } // :320-323
    void Gate(TBool aEnabled, TBool aActive);
private:
    OhmFrameBatch& iBatch;
    IOhmDatagramSink& iSink;
    TBool iEnabled, iActive, iSend;
    TUint iFrame, iSampleRate, iBitRate, iTimestampMultiplier, iBytesPerSample;
    TBool iLossless;
    TUint64 iSamplesTotal, iSampleStart;
    TUint iLatencyMs, iLatencyOhm;
    std::string iCodecName;
};

class Sender : public Media::IPipelineElementDownstream {
public:
    static const TUint kSongcastPacketMs = 5;                                             // Sender.h:35-37
    static const TUint kSongcastPacketJiffies = Media::Jiffies::kPerMs * kSongcastPacketMs;
public:
    /** aSharedBatch: let several senders (streams) share one device pass; the owner of the batch calls Run(). */
    Sender(Media::MsgFactory& aFactory, IOhmDatagramSink& aSink, TUint aMinLatencyMs, OhmFrameBatch* aSharedBatch = nullptr);
    ~Sender();
    /** Frames to collect before the device runs (1 = after every packet, like the reference; 0 = only when Transmit()
     *  is called).  Ignored with a shared batch. */
    void SetBatching(TUint aFrames) { iBatchFrames = aFrames; }
    void Transmit();                                     // run what has been collected now
    OhmSenderDriver& Driver() { return iDriver; }
public: // from Media::IPipelineElementDownstream
    void Push(Media::Msg* aMsg) override;                // consumes the message (Sender.cpp:117-123)
private:
    void NewStream(const Media::DecodedStreamInfo& aInfo);
    void Queue(Media::MsgAudio* aAudio);                 // cuts a packet every kSongcastPacketJiffies
    void SendQueued(TBool aHalt);
private:
    OhmFrameBatch* iOwnBatch;
    OhmFrameBatch& iBatch;
    OhmSenderDriver iDriver;
    std::vector<Media::MsgAudio*> iQueued;
    TUint iQueuedJiffies;
    TUint iSampleRate, iNumChannels, iBitDepth;
    const TUint iMinLatencyMs;
    TBool iStreamForbidden;
    TUint iBatchFrames;
};

} // namespace Av
} // namespace OpenHome
