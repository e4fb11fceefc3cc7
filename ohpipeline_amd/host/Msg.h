// Msg.h -- host-side mirror of the reference's message model for the PCM hot path.
//
// Same names, argument meaning, ownership and error behaviour as the reference (file:line below are relative to the
// reference tree), so that an element or driver written against OpenHome/Media/Pipeline/Msg.h keeps working:
//   Msg / AddRef / RemoveRef        Msg.h:81-105, 242-251   (intrusive refcount; whoever holds a Msg* owns one ref)
//   DecodedAudio                    Msg.h:167-183           (<= 9216 bytes of packed PCM)
//   MsgAudio / MsgAudioPcm / MsgSilence   Msg.cpp:1949-2276, 2466-2560
//   MsgPlayable (Pcm / Silence)     Msg.cpp:2591-2893
//   IMsgProcessor, IPcmProcessor    Msg.h:1177-1240
//   PipelineElement, IPipelineElementUpstream/Downstream   Msg.h:1475-1525, 1844-1856
//   MsgFactory                      Msg.h:1987-2075
// What differs, on purpose (MI355X-first):
//   * DecodedAudio keeps the bytes in the order the codec delivered them; the LE->BE copy of
//     DecodedAudio::ConstructPcm (Msg.cpp:347-408) is a descriptor bit applied by the GPU load.
//   * No PCM byte is touched on the CPU.  MsgPlayable::Read() hands the playable to the GPU through the C ABI
//     (include/ohgpu.h) and then replays the reference's callback sequence (BeginBlock, ProcessFragment*, EndBlock,
//     256-byte fragments when a ramp is enabled, Msg.cpp:2753-2786).  PlayableBatch reads many playables (many
//     streams) in one launch -- that is the path a multi-stream driver should use.
//   * ApplyAttenuation is a pure function of the descriptor: the reference attenuates the shared DecodedAudio in
//     place (Msg.cpp:2742-2750), so reading a clone attenuates twice; here every read sees the original bytes.
//   * Messages are heap objects, not pool cells; the pool-exhaustion asserts (Msg.cpp:123-134) have no equivalent.
#pragma once

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "OhTypes.h"
#include "Ramp.h"

struct ohgpu_ctx;
struct ohgpu_src;

namespace OpenHome {
namespace Media {

enum class AudioDataEndian { Invalid, Little, Big };     // Msg.h:107-112
enum class AudioFormat { Pcm, Dsd, Undefined };
enum class Multiroom { Allowed, Forbidden };              // Msg.h:584-588

class MsgMode; class MsgTrack; class MsgDrain; class MsgDelay; class MsgEncodedStream; class MsgStreamSegment;
class MsgAudioEncoded; class MsgMetaText; class MsgStreamInterrupted; class MsgHalt; class MsgFlush; class MsgWait;
class MsgDecodedStream; class MsgAudioPcm; class MsgAudioDsd; class MsgSilence; class MsgPlayable; class MsgQuit;
class MsgFactory;
class SampleRateConverterStream;

/** A designed polyphase filter on the device (include/ohgpu.h: ohgpu_src), kept by the factory and shared by every stream of the
 *  same conversion: streams that share a filter can share a launch. */
struct SrcFilter {
    ohgpu_src* handle = nullptr;
    TUint L = 0, M = 0, T = 0;
};

class IMsgProcessor;

class Msg {                                              // Msg.h:81-105, 242-251
public:
    void AddRef();
    void RemoveRef();                                    // deletes the message when the count reaches zero
    virtual Msg* Process(IMsgProcessor& aProcessor) = 0;
protected:
    Msg();
    virtual ~Msg();
private:
    std::atomic<TUint> iRefCount;
};

class IMsgProcessor {                                    // Msg.h:1177-1199
public:
    virtual ~IMsgProcessor() {}
    virtual Msg* ProcessMsg(MsgMode* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgTrack* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgDrain* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgDelay* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgEncodedStream* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgStreamSegment* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgAudioEncoded* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgMetaText* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgStreamInterrupted* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgHalt* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgFlush* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgWait* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgDecodedStream* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgAudioPcm* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgAudioDsd* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgSilence* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgPlayable* aMsg) = 0;
    virtual Msg* ProcessMsg(MsgQuit* aMsg) = 0;
};

/** Used to retrieve PCM audio data from a MsgPlayable (Msg.h:1204-1240). */
class IPcmProcessor {
public:
    virtual ~IPcmProcessor() {}
    virtual void BeginBlock() = 0;
    /** aData: packed big endian pcm, always a complete number of samples; aSubsampleBytes in 1..4 */
    virtual void ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) = 0;
    virtual void ProcessSilence(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) = 0;
    virtual void EndBlock() = 0;
    virtual void Flush() = 0;
};

/** Reads packed data into a growing buffer (Media/Utils/ProcessorAudioUtils.cpp:31-56). */
class ProcessorPcmBufTest : public IPcmProcessor {
public:
    Brn Buf() const { return Brn(iBuf.data(), (TUint)iBuf.size()); }
    const TByte* Ptr() const { return iBuf.data(); }
    const std::vector<TUint>& Fragments() const { return iFragments; }
public: // from IPcmProcessor
    void BeginBlock() override { iBuf.clear(); iFragments.clear(); }
    void ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) override;
    void ProcessSilence(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) override;
    void EndBlock() override {}
    void Flush() override {}
private:
    std::vector<TByte> iBuf;
    std::vector<TUint> iFragments;
};

/** Packed PCM as delivered by a codec; shared between the messages split from / cloned from one another. */
class DecodedAudio {
public:
    static const TUint kMaxBytes = 9216;                 // AudioData::kMaxBytes, Msg.h:117
    static const TUint kMaxNumChannels = 8;
public:
    DecodedAudio(const Brx& aData, TUint aBitDepth, AudioDataEndian aEndian);   // DecodedAudio::ConstructPcm asserts
    void Aggregate(const DecodedAudio& aOther);                                  // Msg.cpp:317-323
    const TByte* Ptr(TUint aOffsetBytes) const;
    TUint Bytes() const { return (TUint)iData.size(); }
    AudioDataEndian Endian() const { return iEndian; }
    TUint BitDepth() const { return iBitDepth; }
private:
    std::vector<TByte> iData;
    TUint iBitDepth;
    AudioDataEndian iEndian;
};

// ---- the message types a PCM element has to recognise; only the audio ones carry behaviour here ----
enum class Latency { NotSupported, Internal, External }; // Msg.h:366-371

class ModeInfo {
public:
    TBool iRampPauseResumeLong = true;                   // ModeInfo::RampPauseResumeLong()
    Latency iLatencyMode = Latency::NotSupported;
    TBool RampPauseResumeLong() const { return iRampPauseResumeLong; }
    Latency LatencyMode() const { return iLatencyMode; }
    void SetLatencyMode(Latency aLatencyMode) { iLatencyMode = aLatencyMode; }
};

class MsgMode : public Msg {
public:
    explicit MsgMode(const ModeInfo& aInfo, const std::string& aMode = std::string()) : iInfo(aInfo), iMode(aMode) {}
    const ModeInfo& Info() const { return iInfo; }
    const std::string& Mode() const { return iMode; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    ModeInfo iInfo;
    std::string iMode;
};

#define OH_TRIVIAL_MSG(Name)                                                                              \
    class Name : public Msg {                                                                            \
    public:                                                                                               \
        Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }        \
    }
OH_TRIVIAL_MSG(MsgTrack);
OH_TRIVIAL_MSG(MsgDrain);
OH_TRIVIAL_MSG(MsgStreamSegment);
OH_TRIVIAL_MSG(MsgAudioEncoded);
OH_TRIVIAL_MSG(MsgMetaText);
OH_TRIVIAL_MSG(MsgStreamInterrupted);
OH_TRIVIAL_MSG(MsgHalt);
OH_TRIVIAL_MSG(MsgWait);
OH_TRIVIAL_MSG(MsgAudioDsd);
OH_TRIVIAL_MSG(MsgQuit);

class MsgDelay : public Msg {                            // Msg.h:470-487
public:
    explicit MsgDelay(TUint aRemainingJiffies = 0) : iRemainingJiffies(aRemainingJiffies) {}
    TUint RemainingJiffies() const { return iRemainingJiffies; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    TUint iRemainingJiffies;
};

class MsgEncodedStream : public Msg {                    // Msg.h:603-650 (the one field the PCM path reads)
public:
    enum class Format { Encoded, Pcm, Dsd };
    explicit MsgEncodedStream(Format aFormat = Format::Encoded) : iFormat(aFormat) {}
    Format StreamFormat() const { return iFormat; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    Format iFormat;
};

class MsgFlush : public Msg {                            // Msg.h:749-763
public:
    static const TUint kIdInvalid = 0;
    explicit MsgFlush(TUint aId = kIdInvalid) : iId(aId) {}
    TUint Id() const { return iId; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    TUint iId;
};

class IStreamHandler {                                   // Msg.h:556-575 (the one call the PCM path makes)
public:
    static const TUint kStreamIdInvalid = 0;             // IPipelineIdProvider::kStreamIdInvalid
    virtual ~IStreamHandler() {}
    virtual void NotifyStarving(const Brx& aMode, TUint aStreamId, TBool aStarving) = 0;
};

class DecodedStreamInfo {                              // Msg.h:1062-1110 (the fields the PCM path reads)
public:
    TUint iStreamId = 0, iBitRate = 0, iBitDepth = 0, iSampleRate = 0, iNumChannels = 0;
    TUint64 iTrackLength = 0, iSampleStart = 0;
    TBool iLossless = true, iSeekable = false, iLive = false;
    AudioFormat iFormat = AudioFormat::Pcm;
    Media::Multiroom iMultiroom = Media::Multiroom::Allowed;
    std::string iCodecName;
    IStreamHandler* iStreamHandler = nullptr;
    IStreamHandler* StreamHandler() const { return iStreamHandler; }
    TUint BitRate() const { return iBitRate; }
    TUint64 TrackLength() const { return iTrackLength; }             // jiffies
    TBool Lossless() const { return iLossless; }
    Media::Multiroom Multiroom() const { return iMultiroom; }
    Brn CodecName() const { return Brn((const TByte*)iCodecName.data(), (TUint)iCodecName.size()); }
    TUint StreamId() const { return iStreamId; }
    TUint BitDepth() const { return iBitDepth; }
    TUint SampleRate() const { return iSampleRate; }
    TUint NumChannels() const { return iNumChannels; }
    TUint64 SampleStart() const { return iSampleStart; }
    TBool Live() const { return iLive; }
    AudioFormat Format() const { return iFormat; }
};

class MsgDecodedStream : public Msg {
public:
    explicit MsgDecodedStream(const DecodedStreamInfo& aInfo) : iInfo(aInfo) {}
    const DecodedStreamInfo& StreamInfo() const { return iInfo; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    DecodedStreamInfo iInfo;
};

class MsgAudio : public Msg {                            // Msg.cpp:1940-2107
public:
    MsgAudio* Split(TUint aJiffies);                     // returns the remainder; asserts 0 < aJiffies < Jiffies()
    virtual MsgAudio* Clone();
    TUint Jiffies() const { return iSize; }
    // returns the ramp value reached at the end of this message; aSplit is set when ramps crossed inside it
    TUint SetRamp(TUint aStart, TUint& aRemainingDuration, Ramp::EDirection aDirection, MsgAudio*& aSplit);
    void ClearRamp() { iRamp.Reset(); }
    void SetMuted() { iRamp.SetMuted(); }
    const Media::Ramp& Ramp() const { return iRamp; }
    TUint MedianRampMultiplier();
    TUint SampleRate() const { return iSampleRate; }
    TUint BitDepth() const { return iBitDepth; }
    TUint NumChannels() const { return iNumChannels; }
protected:
    MsgAudio(TUint aSampleRate, TUint aBitDepth, TUint aChannels);
    virtual MsgAudio* Allocate() = 0;                    // a fresh message of the dynamic type, fields copied by Split/Clone
    virtual void SplitCompleted(MsgAudio& aRemaining) {}
protected:
    TUint iSize = 0;                                     // jiffies
    TUint iOffset = 0;                                   // jiffies into the DecodedAudio
    Media::Ramp iRamp;
    TUint iSampleRate, iBitDepth, iNumChannels;
};

class MsgAudioPcm : public MsgAudio {                    // Msg.cpp:2109-2305
    friend class MsgFactory;
    friend class SampleRateConverter;
public:
    static const TUint kUnityAttenuation = 256;
    static const TUint64 kTrackOffsetInvalid = UINT64_MAX;
public:
    MsgAudio* Clone() override;
    TUint64 TrackOffset() const { return iTrackOffset; }
    void Aggregate(MsgAudioPcm* aMsg);                   // consumes aMsg's reference
    MsgPlayable* CreatePlayable();                       // consumes this message's reference (Msg.cpp:2234-2262)
    void SetAttenuation(TUint aAttenuation) { iAttenuation = aAttenuation; }
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    MsgAudioPcm(MsgFactory& aFactory, std::shared_ptr<DecodedAudio> aAudio, TUint aSampleRate, TUint aBitDepth, TUint aChannels, TUint64 aTrackOffset);
    MsgAudio* Allocate() override;
    void SplitCompleted(MsgAudio& aRemaining) override;
private:
    MsgFactory& iFactory;
    std::shared_ptr<DecodedAudio> iAudioData;            // null for resampled audio
    std::shared_ptr<SampleRateConverterStream> iResampled;   // set for audio produced by SampleRateConverter
    TUint64 iResampledFrame0 = 0;                        // absolute output frame of jiffy offset 0
    TUint64 iTrackOffset;
    TUint iAttenuation = kUnityAttenuation;
};

class MsgSilence : public MsgAudio {                     // Msg.cpp:2458-2560
    friend class MsgFactory;
public:
    MsgPlayable* CreatePlayable();
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
private:
    MsgSilence(MsgFactory& aFactory, TUint& aJiffies, TUint aSampleRate, TUint aBitDepth, TUint aChannels);
    MsgAudio* Allocate() override;
    void SplitCompleted(MsgAudio& aRemaining) override;
private:
    MsgFactory& iFactory;
};

/** What one MsgPlayable asks the device to do: exactly the fields of ohgpu_msg_desc / ohgpu_src_msg_desc. */
struct PlayableWork {
    TBool silence = false;
    TBool resampled = false;
    std::shared_ptr<DecodedAudio> audio;
    std::shared_ptr<SampleRateConverterStream> stream;
    TUint64 outFrame0 = 0;       // resampled: absolute first output frame
    TUint offsetBytes = 0, sizeBytes = 0, frames = 0;
    TUint sampleRate = 0, bitDepth = 0, channels = 0, attenuation = 256;
    Media::Ramp ramp;
};

class MsgPlayable : public Msg {                         // Msg.cpp:2591-2653
    friend class MsgAudioPcm;
    friend class MsgSilence;
    friend class PlayableBatch;
public:
    MsgPlayable* Split(TUint aBytes);                    // returns the remainder, nullptr when aBytes == Bytes()
    TUint Bytes() const { return iWork.sizeBytes; }
    TUint Jiffies() const { return iJiffies; }
    const Media::Ramp& Ramp() const { return iWork.ramp; }
    /** Runs this one playable on the GPU and replays the reference's callback sequence.  Prefer PlayableBatch. */
    void Read(IPcmProcessor& aProcessor);
    Msg* Process(IMsgProcessor& aProcessor) override { return aProcessor.ProcessMsg(this); }
    const PlayableWork& Work() const { return iWork; }
private:
    MsgPlayable(MsgFactory& aFactory, const PlayableWork& aWork, TUint aJiffies);
private:
    MsgFactory& iFactory;
    PlayableWork iWork;
    TUint iJiffies;
};

/** Reads many playables -- typically one per stream per driver period -- then replays each one's BeginBlock /
 *  ProcessFragment* / EndBlock sequence in the order they were added (Msg.cpp:2646-2653, 2753-2786 per message, on the driver's
 *  thread: AnimatorBasic.cpp:77-142).  One device call for all the plain audio and ONE PER FILTER for the rate-converted audio, however
 *  many streams share it: what goes to the device is each message's window of input (SampleRateConverterStream::Window), packed
 *  back to back in the factory's pinned arena, and what comes back is the messages' output and nothing else.  Output depth/endian
 *  default to the playable's own (pass-through); SetOutputFormat asks the device for the conversion the processor would do.
 *  The fragments handed to the processors point into the factory's arena: valid until the factory's next Run (the reference
 *  lends a Brx for the call only, Msg.h:1204-1240).  A batch object may be reused period after period: it keeps its scratch. */
class PlayableBatch {
public:
    explicit PlayableBatch(MsgFactory& aFactory);
    ~PlayableBatch();
    void SetOutputFormat(TUint aBitDepth, AudioDataEndian aEndian);   // 0 = keep each playable's depth
    void Add(MsgPlayable* aPlayable, IPcmProcessor& aProcessor);      // takes over the caller's reference
    void Run();                                                        // launches, waits, delivers, releases
    TUint Count() const { return (TUint)iItems.size(); }
private:
    struct Item { MsgPlayable* playable; IPcmProcessor* processor; TUint64 outOffset; TUint outBits; };
    struct WindowRun;                                                  // consecutive items of one stream whose outputs follow on: one window
    struct Group;                                                      // the items one device call serves
    MsgFactory& iFactory;
    std::vector<Item> iItems;
    TUint iOutBits = 0;
    AudioDataEndian iOutEndian = AudioDataEndian::Big;
    struct Scratch;
    std::unique_ptr<Scratch> iScratch;
};

class MsgFactory {                                       // Msg.h:1987-2075 (the creators the PCM path uses)
public:
    /** aDevice: HIP device index; throws AssertionFailed if it cannot be opened (there is no CPU fallback).
     *  aDevice < 0 builds a control-plane-only factory: messages, splits and ramps work, reading audio asserts. */
    explicit MsgFactory(int aDevice = 0);
    ~MsgFactory();
    MsgMode* CreateMsgMode(const ModeInfo& aInfo);
    MsgMode* CreateMsgMode(const ModeInfo& aInfo, const std::string& aMode) { return new MsgMode(aInfo, aMode); }
    MsgDecodedStream* CreateMsgDecodedStream(const DecodedStreamInfo& aInfo);
    MsgAudioPcm* CreateMsgAudioPcm(const Brx& aData, TUint aChannels, TUint aSampleRate, TUint aBitDepth, AudioDataEndian aEndian, TUint64 aTrackOffset);
    MsgSilence* CreateMsgSilence(TUint& aSizeJiffies, TUint aSampleRate, TUint aBitDepth, TUint aChannels);
    MsgHalt* CreateMsgHalt();
    MsgQuit* CreateMsgQuit();
    MsgTrack* CreateMsgTrack() { return new MsgTrack(); }
    MsgDrain* CreateMsgDrain() { return new MsgDrain(); }
    MsgDelay* CreateMsgDelay(TUint aRemainingJiffies) { return new MsgDelay(aRemainingJiffies); }
    MsgEncodedStream* CreateMsgEncodedStream(MsgEncodedStream::Format aFormat = MsgEncodedStream::Format::Encoded) { return new MsgEncodedStream(aFormat); }
    MsgMetaText* CreateMsgMetaText() { return new MsgMetaText(); }
    MsgStreamInterrupted* CreateMsgStreamInterrupted() { return new MsgStreamInterrupted(); }
    MsgFlush* CreateMsgFlush(TUint aId) { return new MsgFlush(aId); }
    MsgWait* CreateMsgWait() { return new MsgWait(); }
    ohgpu_ctx* Gpu() const;
    /** The filter of a conversion, designed and uploaded on first use and kept for the factory's lifetime. */
    const SrcFilter& SharedFilter(TUint aRateIn, TUint aRateOut, TUint aTapsPerPhase, double aBeta, double aPassHz);
    TUint FilterCount() const;
    /** The driver thread's pinned staging for host-buffer reads (PlayableBatch::Run): at least the sizes asked for, kept and grown
     *  with headroom.  Like the ohgpu_ctx it belongs to one thread at a time. */
    void ReserveArena(size_t aSrcBytes, size_t aDstBytes, TByte*& aSrc, TByte*& aDst);
private:
    ohgpu_ctx* iCtx;
    mutable std::mutex iFilterLock;
    std::map<std::tuple<TUint, TUint, TUint, double, double>, SrcFilter> iFilters;
    TByte* iArenaSrc = nullptr;
    TByte* iArenaDst = nullptr;
    size_t iArenaSrcBytes = 0, iArenaDstBytes = 0;
};

// ---- element plumbing (Msg.h:1475-1525, 1844-1856; Msg.cpp:3585-3705) ----
class IPipelineElementUpstream {
public:
    virtual ~IPipelineElementUpstream() {}
    virtual Msg* Pull() = 0;
};

/** The dynamic type of a message, found by one visit (elements that treat most kinds alike switch on it instead of overriding eighteen hooks). */
enum class MsgKind { Mode, Track, Drain, Delay, EncodedStream, StreamSegment, AudioEncoded, MetaText, StreamInterrupted, Halt,
                     Flush, Wait, DecodedStream, AudioPcm, AudioDsd, Silence, Playable, Quit };
MsgKind KindOf(Msg* aMsg);

class IPipelineElementDownstream {
public:
    virtual ~IPipelineElementDownstream() {}
    virtual void Push(Msg* aMsg) = 0;
};

class PipelineElement : public IMsgProcessor {
protected:
    enum MsgType {
        eMode = 1, eTrack = 1 << 1, eDrain = 1 << 2, eDelay = 1 << 3, eEncodedStream = 1 << 4, eStreamSegment = 1 << 5,
        eAudioEncoded = 1 << 6, eMetatext = 1 << 7, eStreamInterrupted = 1 << 8, eHalt = 1 << 9, eFlush = 1 << 10,
        eWait = 1 << 11, eDecodedStream = 1 << 12, eAudioPcm = 1 << 13, eAudioDsd = 1 << 14, eSilence = 1 << 15,
        ePlayable = 1 << 16, eQuit = 1 << 17
    };
    explicit PipelineElement(TUint aSupportedTypes) : iSupportedTypes(aSupportedTypes) {}
protected: // IMsgProcessor: pass a supported message through, ASSERT on an unsupported one (Msg.cpp:3594-3597)
    Msg* ProcessMsg(MsgMode* aMsg) override;
    Msg* ProcessMsg(MsgTrack* aMsg) override;
    Msg* ProcessMsg(MsgDrain* aMsg) override;
    Msg* ProcessMsg(MsgDelay* aMsg) override;
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override;
    Msg* ProcessMsg(MsgStreamSegment* aMsg) override;
    Msg* ProcessMsg(MsgAudioEncoded* aMsg) override;
    Msg* ProcessMsg(MsgMetaText* aMsg) override;
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override;
    Msg* ProcessMsg(MsgHalt* aMsg) override;
    Msg* ProcessMsg(MsgFlush* aMsg) override;
    Msg* ProcessMsg(MsgWait* aMsg) override;
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgAudioDsd* aMsg) override;
    Msg* ProcessMsg(MsgSilence* aMsg) override;
    Msg* ProcessMsg(MsgPlayable* aMsg) override;
    Msg* ProcessMsg(MsgQuit* aMsg) override;
private:
    void CheckSupported(MsgType aType) const { ASSERT((iSupportedTypes & aType) == (TUint)aType); }
    TUint iSupportedTypes;
};

} // namespace Media
} // namespace OpenHome
