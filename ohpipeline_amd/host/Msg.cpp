// Msg.cpp -- host-side message model for the PCM hot path (see Msg.h for the reference lines each class follows).
#include "Msg.h"

#include <algorithm>
#include <cstring>
#include <map>

#include "../../include/ohgpu.h"
#include "SampleRateConverter.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- Msg
Msg::Msg() : iRefCount(1) {}
Msg::~Msg() {}

void Msg::AddRef()
{
    iRefCount++;
}

void Msg::RemoveRef()
{
    ASSERT(iRefCount != 0);
    if (--iRefCount == 0) {
        delete this;
    }
}

// ---------------------------------------------------------------- ProcessorPcmBufTest
void ProcessorPcmBufTest::ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes)
{
    ASSERT(aData.Bytes() % (aSubsampleBytes * aNumChannels) == 0);
    iBuf.insert(iBuf.end(), aData.Ptr(), aData.Ptr() + aData.Bytes());
    iFragments.push_back(aData.Bytes());
}

void ProcessorPcmBufTest::ProcessSilence(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes)
{
    ProcessFragment(aData, aNumChannels, aSubsampleBytes);
}

// ---------------------------------------------------------------- DecodedAudio
DecodedAudio::DecodedAudio(const Brx& aData, TUint aBitDepth, AudioDataEndian aEndian)
    : iBitDepth(aBitDepth)
    , iEndian(aEndian)
{
    ASSERT((aBitDepth & 7) == 0);                            // Msg.cpp:349-350
    ASSERT(aBitDepth == 8 || aBitDepth == 16 || aBitDepth == 24 || aBitDepth == 32);
    ASSERT(aData.Bytes() % (aBitDepth / 8) == 0);
    ASSERT(aData.Bytes() <= kMaxBytes);
    iData.assign(aData.Ptr(), aData.Ptr() + aData.Bytes());
}

void DecodedAudio::Aggregate(const DecodedAudio& aOther)
{
    ASSERT(aOther.iEndian == iEndian && aOther.iBitDepth == iBitDepth);
    ASSERT(iData.size() + aOther.iData.size() <= kMaxBytes);  // Bws<kMaxBytes>::Append asserts on overflow
    iData.insert(iData.end(), aOther.iData.begin(), aOther.iData.end());
}

const TByte* DecodedAudio::Ptr(TUint aOffsetBytes) const
{
    ASSERT(aOffsetBytes <= iData.size());
    return iData.data() + aOffsetBytes;
}

// ---------------------------------------------------------------- MsgAudio
MsgAudio::MsgAudio(TUint aSampleRate, TUint aBitDepth, TUint aChannels)
    : iSampleRate(aSampleRate)
    , iBitDepth(aBitDepth)
    , iNumChannels(aChannels)
{
}

MsgAudio* MsgAudio::Split(TUint aJiffies)
{
    ASSERT(aJiffies > 0);
    ASSERT(aJiffies < iSize);
    MsgAudio* remaining = Allocate();
    remaining->iOffset = iOffset + aJiffies;
    remaining->iSize = iSize - aJiffies;
    if (iRamp.IsEnabled()) {
        remaining->iRamp = iRamp.Split(aJiffies, iSize);
    }
    else {
        remaining->iRamp.Reset();
    }
    iSize = aJiffies;
    SplitCompleted(*remaining);
    return remaining;
}

MsgAudio* MsgAudio::Clone()
{
    MsgAudio* clone = Allocate();
    clone->iSize = iSize;
    clone->iOffset = iOffset;
    clone->iRamp = iRamp;
    return clone;
}

TUint MsgAudio::SetRamp(TUint aStart, TUint& aRemainingDuration, Ramp::EDirection aDirection, MsgAudio*& aSplit)
{
    const TUint remainingDuration = aRemainingDuration;
    aSplit = nullptr;
    ASSERT(aDirection == Ramp::EUp || aDirection == Ramp::EDown);
    if (iRamp.IsEnabled() && iRamp.Direction() == Ramp::EMute) {      // a muted message stays muted (Msg.cpp:1997-2002)
        if (aDirection == Ramp::EDown) {
            aRemainingDuration = 0;
        }
        return iRamp.End();
    }
    Media::Ramp split;
    TUint splitPos;
    if (iRamp.Set(aStart, iSize, remainingDuration, aDirection, split, splitPos)) {
        if (splitPos == 0) {
            iRamp = split;
        }
        else if (splitPos != iSize) {
            const Media::Ramp first = iRamp;       // Split() rescales ramps; put the intended pair back afterwards
            aSplit = Split(splitPos);
            iRamp = first;
            aSplit->iRamp = split;
        }
    }
    aRemainingDuration -= iSize;
    if (aSplit != nullptr && aSplit->iRamp.Direction() != aDirection && aDirection == Ramp::EUp) {
        aRemainingDuration += aSplit->iSize;       // the tail runs the other way: roughly compensate (Msg.cpp:2032-2034)
    }
    if (aDirection == Ramp::EDown && iRamp.End() == Ramp::kMin) {
        aRemainingDuration = 0;
    }
    else if (aDirection == Ramp::EUp && iRamp.End() == Ramp::kMax) {
        aRemainingDuration = 0;
    }
    return iRamp.End();
}

TUint MsgAudio::MedianRampMultiplier()
{
    if (!iRamp.IsEnabled()) {
        return 0x8000;
    }
    if (iRamp.Direction() == Ramp::EMute) {
        return 0;
    }
    const TUint mult = Ramp::MedianMultiplier(iRamp);
    iRamp.Reset();
    return mult;
}

// ---------------------------------------------------------------- MsgAudioPcm
MsgAudioPcm::MsgAudioPcm(MsgFactory& aFactory, std::shared_ptr<DecodedAudio> aAudio, TUint aSampleRate, TUint aBitDepth,
                         TUint aChannels, TUint64 aTrackOffset)
    : MsgAudio(aSampleRate, aBitDepth, aChannels)
    , iFactory(aFactory)
    , iAudioData(aAudio)
    , iTrackOffset(aTrackOffset)
{
}

MsgAudio* MsgAudioPcm::Allocate()
{
    MsgAudioPcm* msg = new MsgAudioPcm(iFactory, iAudioData, iSampleRate, iBitDepth, iNumChannels, iTrackOffset);
    msg->iResampled = iResampled;
    msg->iResampledFrame0 = iResampledFrame0;
    msg->iAttenuation = iAttenuation;
    return msg;
}

MsgAudio* MsgAudioPcm::Clone()
{
    return MsgAudio::Clone();
}

void MsgAudioPcm::SplitCompleted(MsgAudio& aRemaining)
{
    MsgAudioPcm& remaining = static_cast<MsgAudioPcm&>(aRemaining);
    remaining.iTrackOffset = (iTrackOffset == kTrackOffsetInvalid) ? iTrackOffset : iTrackOffset + iSize;
}

void MsgAudioPcm::Aggregate(MsgAudioPcm* aMsg)
{
    ASSERT(aMsg->iSampleRate == iSampleRate);
    ASSERT(aMsg->iBitDepth == iBitDepth);
    ASSERT(aMsg->iNumChannels == iNumChannels);
    ASSERT(aMsg->iTrackOffset == iTrackOffset + Jiffies());          // must logically follow this one
    ASSERT(!iRamp.IsEnabled() && !aMsg->iRamp.IsEnabled());          // no ramps allowed
    ASSERT(iAudioData != nullptr && aMsg->iAudioData != nullptr);
    iAudioData->Aggregate(*aMsg->iAudioData);
    iSize += aMsg->Jiffies();
    aMsg->RemoveRef();
}

MsgPlayable* MsgAudioPcm::CreatePlayable()
{
    const TUint jiffiesPerSample = Jiffies::PerSample(iSampleRate);
    TUint offsetJiffies = iOffset;
    const TUint offsetBytes = Jiffies::ToBytes(offsetJiffies, jiffiesPerSample, iNumChannels, iBitDepth);
    TUint sizeJiffies = iSize + (iOffset - offsetJiffies);           // offset and size round down to whole samples
    const TUint sizeBytes = Jiffies::ToBytes(sizeJiffies, jiffiesPerSample, iNumChannels, iBitDepth);
    PlayableWork work;
    work.sampleRate = iSampleRate;
    work.bitDepth = iBitDepth;
    work.channels = iNumChannels;
    work.sizeBytes = sizeBytes;
    work.frames = sizeBytes / ((iBitDepth / 8) * iNumChannels);
    if (iRamp.Direction() != Ramp::EMute) {
        work.offsetBytes = offsetBytes;
        work.attenuation = iAttenuation;
        work.ramp = iRamp;
        if (iResampled != nullptr) {
            work.resampled = true;
            work.stream = iResampled;
            work.outFrame0 = iResampledFrame0 + offsetJiffies / jiffiesPerSample;
        }
        else {
            work.audio = iAudioData;
        }
    }
    else {                                                           // muted: silence of the same length, no ramp
        work.silence = true;
    }
    MsgPlayable* playable = new MsgPlayable(iFactory, work, iSize);
    RemoveRef();
    return playable;
}

// ---------------------------------------------------------------- MsgSilence
MsgSilence::MsgSilence(MsgFactory& aFactory, TUint& aJiffies, TUint aSampleRate, TUint aBitDepth, TUint aChannels)
    : MsgAudio(aSampleRate, aBitDepth, aChannels)
    , iFactory(aFactory)
{
    Jiffies::RoundDownNonZeroSampleBlock(aJiffies, Jiffies::PerSample(aSampleRate));
    iSize = aJiffies;
}

MsgAudio* MsgSilence::Allocate()
{
    TUint jiffies = Jiffies::PerSample(iSampleRate);
    return new MsgSilence(iFactory, jiffies, iSampleRate, iBitDepth, iNumChannels);
}

void MsgSilence::SplitCompleted(MsgAudio& aRemaining)
{
    // both parts stay whole samples; what does not fit the first part moves to the second (Msg.cpp:2520-2545)
    MsgSilence& remaining = static_cast<MsgSilence&>(aRemaining);
    const TUint block = Jiffies::PerSample(iSampleRate);
    const TUint extra = iSize % block;
    iSize -= extra;
    remaining.iSize += extra;
}

MsgPlayable* MsgSilence::CreatePlayable()
{
    const TUint jiffiesPerSample = Jiffies::PerSample(iSampleRate);
    TUint jiffies = iSize;
    PlayableWork work;
    work.silence = true;
    work.sampleRate = iSampleRate;
    work.bitDepth = iBitDepth;
    work.channels = iNumChannels;
    work.sizeBytes = Jiffies::ToBytes(jiffies, jiffiesPerSample, iNumChannels, iBitDepth);
    work.frames = work.sizeBytes / ((iBitDepth / 8) * iNumChannels);
    work.ramp = iRamp;
    MsgPlayable* playable = new MsgPlayable(iFactory, work, iSize);
    RemoveRef();
    return playable;
}

// ---------------------------------------------------------------- MsgPlayable
MsgPlayable::MsgPlayable(MsgFactory& aFactory, const PlayableWork& aWork, TUint aJiffies)
    : iFactory(aFactory)
    , iWork(aWork)
    , iJiffies(aJiffies)
{
}

MsgPlayable* MsgPlayable::Split(TUint aBytes)
{
    ASSERT(aBytes <= iWork.sizeBytes);
    ASSERT(aBytes != 0);
    if (aBytes == iWork.sizeBytes) {
        return nullptr;
    }
    const TUint bytesPerSample = (iWork.bitDepth / 8) * iWork.channels;
    const TUint numSamples = aBytes / bytesPerSample;
    const TUint splitJiffies = numSamples * Jiffies::PerSample(iWork.sampleRate);
    PlayableWork rest = iWork;
    rest.offsetBytes = iWork.offsetBytes + aBytes;
    rest.sizeBytes = iWork.sizeBytes - aBytes;
    rest.frames = rest.sizeBytes / bytesPerSample;
    rest.outFrame0 = iWork.outFrame0 + numSamples;
    if (iWork.ramp.IsEnabled()) {
        rest.ramp = iWork.ramp.Split(aBytes, iWork.sizeBytes);       // bytes are the unit here (Msg.cpp:2611-2613)
    }
    else {
        rest.ramp.Reset();
    }
    MsgPlayable* remaining = new MsgPlayable(iFactory, rest, iJiffies - splitJiffies);
    iWork.sizeBytes = aBytes;
    iWork.frames = numSamples;
    iJiffies = splitJiffies;
    return remaining;
}

void MsgPlayable::Read(IPcmProcessor& aProcessor)
{
    PlayableBatch batch(iFactory);
    AddRef();                        // the batch releases one reference; Read() leaves ownership with the caller
    batch.Add(this, aProcessor);
    batch.Run();
}

// ---------------------------------------------------------------- PlayableBatch
PlayableBatch::PlayableBatch(MsgFactory& aFactory)
    : iFactory(aFactory)
{
}

PlayableBatch::~PlayableBatch()
{
    for (auto& item : iItems) {
        item.playable->RemoveRef();
    }
}

void PlayableBatch::SetOutputFormat(TUint aBitDepth, AudioDataEndian aEndian)
{
    ASSERT(aBitDepth == 0 || aBitDepth == 8 || aBitDepth == 16 || aBitDepth == 24 || aBitDepth == 32);
    ASSERT(aEndian == AudioDataEndian::Big || aEndian == AudioDataEndian::Little);
    iOutBits = aBitDepth;
    iOutEndian = aEndian;
}

void PlayableBatch::Add(MsgPlayable* aPlayable, IPcmProcessor& aProcessor)
{
    iItems.push_back({aPlayable, &aProcessor});
}

static uint8_t GpuEndian(AudioDataEndian aEndian)
{
    return aEndian == AudioDataEndian::Little ? OHGPU_ENDIAN_LITTLE : OHGPU_ENDIAN_BIG;
}

void PlayableBatch::Run()
{
    ohgpu_ctx* ctx = iFactory.Gpu();
    // ---- lay out the arenas: every distinct DecodedAudio once in the source arena, outputs back to back ----
    std::vector<TByte> src;
    std::map<const DecodedAudio*, TUint64> audioBase;
    std::vector<ohgpu_msg_desc> descs;
    std::vector<size_t> descItem;
    struct SrcGroup { std::vector<ohgpu_src_msg_desc> descs; std::vector<size_t> item; std::vector<TByte> in; };
    std::map<SampleRateConverterStream*, SrcGroup> groups;
    std::vector<TUint64> outOffset(iItems.size());
    std::vector<TUint> outBits(iItems.size());
    TUint64 dstBytes = 0;
    for (size_t i = 0; i < iItems.size(); i++) {
        const PlayableWork& w = iItems[i].playable->Work();
        const TUint srcBits = w.resampled ? 24 : w.bitDepth;
        outBits[i] = (iOutBits == 0) ? w.bitDepth : iOutBits;
        outOffset[i] = dstBytes;
        dstBytes += (TUint64)w.frames * w.channels * (outBits[i] / 8);
        dstBytes = (dstBytes + 63) & ~(TUint64)63;                    // keeps every message's output line-aligned
        if (w.frames == 0) {
            continue;
        }
        if (w.resampled) {
            SrcGroup& g = groups[w.stream.get()];
            ohgpu_src_msg_desc d;
            memset(&d, 0, sizeof(d));
            w.stream->DescribeWindow(w.outFrame0, w.frames, d);       // src_offset/src_frame0/src_frames within the stream's history
            d.out_frame0 = w.outFrame0;
            d.dst_offset = outOffset[i];
            d.n_frames = w.frames;
            d.ramp_start = (uint16_t)w.ramp.Start();
            d.ramp_end = (uint16_t)w.ramp.End();
            d.attenuation = OHGPU_UNITY_ATTENUATION;
            d.channels = (uint8_t)w.channels;
            d.src_bits = (uint8_t)w.stream->SourceBitDepth();
            d.src_endian = GpuEndian(w.stream->SourceEndian());
            d.dst_bits = (uint8_t)outBits[i];
            d.dst_endian = GpuEndian(iOutEndian);
            d.flags = w.ramp.IsEnabled() ? OHGPU_FLAG_RAMP : 0;
            g.descs.push_back(d);
            g.item.push_back(i);
            (void)srcBits;
            continue;
        }
        ohgpu_msg_desc d;
        memset(&d, 0, sizeof(d));
        d.dst_offset = outOffset[i];
        d.n_frames = w.frames;
        d.ramp_start = (uint16_t)w.ramp.Start();
        d.ramp_end = (uint16_t)w.ramp.End();
        d.attenuation = (uint16_t)w.attenuation;
        d.channels = (uint8_t)w.channels;
        d.src_bits = (uint8_t)w.bitDepth;
        d.dst_bits = (uint8_t)outBits[i];
        d.dst_endian = GpuEndian(iOutEndian);
        d.src_endian = OHGPU_ENDIAN_BIG;
        if (w.silence) {
            d.flags = OHGPU_FLAG_SILENCE;
        }
        else {
            auto it = audioBase.find(w.audio.get());
            if (it == audioBase.end()) {
                it = audioBase.emplace(w.audio.get(), (TUint64)src.size()).first;
                src.insert(src.end(), w.audio->Ptr(0), w.audio->Ptr(0) + w.audio->Bytes());
            }
            d.src_offset = it->second + w.offsetBytes;
            d.src_endian = GpuEndian(w.audio->Endian());
            d.flags = w.ramp.IsEnabled() ? OHGPU_FLAG_RAMP : 0;
        }
        descs.push_back(d);
        descItem.push_back(i);
    }
    std::vector<TByte> dst((size_t)dstBytes);
    if (!descs.empty()) {
        const int err = ohgpu_pcm_process_host(ctx, descs.data(), descs.size(), src.data(), src.size(), dst.data(), dst.size());
        ASSERT(err == OHGPU_OK);
    }
    for (auto& kv : groups) {
        SampleRateConverterStream* stream = kv.first;
        SrcGroup& g = kv.second;
        const int err = ohgpu_src_process_host(ctx, stream->Filter(), g.descs.data(), g.descs.size(),
                                               stream->HistoryPtr(), stream->HistoryBytes(), dst.data(), dst.size());
        ASSERT(err == OHGPU_OK);
    }
    // ---- deliver, message by message, with the reference's callback sequence (Msg.cpp:2646-2653, 2753-2786, 2874-2893) ----
    for (size_t i = 0; i < iItems.size(); i++) {
        MsgPlayable* playable = iItems[i].playable;
        IPcmProcessor& proc = *iItems[i].processor;
        const PlayableWork& w = playable->Work();
        const TUint subsampleBytes = outBits[i] / 8;
        const TUint outFrameBytes = subsampleBytes * w.channels;
        const TByte* out = dst.data() + outOffset[i];
        proc.BeginBlock();
        if (w.frames > 0) {
            const TUint srcFrameBytes = (w.bitDepth / 8) * w.channels;
            TUint framesPerFragment;
            if (w.silence) {
                const TUint maxBytes = DecodedAudio::kMaxBytes - (DecodedAudio::kMaxBytes % srcFrameBytes);
                framesPerFragment = maxBytes / srcFrameBytes;
            }
            else if (w.ramp.IsEnabled()) {
                framesPerFragment = 256 / srcFrameBytes;              // Bws<256> rampedBuf
            }
            else {
                framesPerFragment = w.frames;                          // one zero-copy fragment
            }
            for (TUint done = 0; done < w.frames; done += framesPerFragment) {
                const TUint n = std::min(framesPerFragment, w.frames - done);
                const Brn frag(out + (size_t)done * outFrameBytes, n * outFrameBytes);
                if (w.silence) {
                    proc.ProcessSilence(frag, w.channels, subsampleBytes);
                }
                else {
                    proc.ProcessFragment(frag, w.channels, subsampleBytes);
                }
            }
        }
        proc.EndBlock();
        playable->RemoveRef();
    }
    iItems.clear();
}

// ---------------------------------------------------------------- MsgFactory
MsgFactory::MsgFactory(int aDevice)
    : iCtx(nullptr)
{
    if (aDevice >= 0) {
        const int err = ohgpu_init(aDevice, &iCtx);
        ASSERT(err == OHGPU_OK);    // no GPU, no data plane: this path never falls back to the CPU
    }
}

MsgFactory::~MsgFactory()
{
    if (iCtx != nullptr) {
        ohgpu_shutdown(iCtx);
    }
}

ohgpu_ctx* MsgFactory::Gpu() const
{
    ASSERT(iCtx != nullptr);        // a control-plane-only factory (device < 0) cannot read audio
    return iCtx;
}

MsgMode* MsgFactory::CreateMsgMode(const ModeInfo& aInfo)
{
    return new MsgMode(aInfo);
}

MsgDecodedStream* MsgFactory::CreateMsgDecodedStream(const DecodedStreamInfo& aInfo)
{
    return new MsgDecodedStream(aInfo);
}

MsgAudioPcm* MsgFactory::CreateMsgAudioPcm(const Brx& aData, TUint aChannels, TUint aSampleRate, TUint aBitDepth,
                                           AudioDataEndian aEndian, TUint64 aTrackOffset)
{
    auto audio = std::make_shared<DecodedAudio>(aData, aBitDepth, aEndian);
    const TUint numSubsamples = aData.Bytes() / (aBitDepth / 8);
    ASSERT(aChannels != 0 && numSubsamples % aChannels == 0);        // Msg.cpp:2164
    MsgAudioPcm* msg = new MsgAudioPcm(*this, audio, aSampleRate, aBitDepth, aChannels, aTrackOffset);
    msg->iSize = (numSubsamples / aChannels) * Jiffies::PerSample(aSampleRate);
    if (msg->iSize == 0) {
        msg->RemoveRef();
        ASSERTS();                                                   // zero-length audio asserts (Msg.cpp:2166)
    }
    return msg;
}

MsgSilence* MsgFactory::CreateMsgSilence(TUint& aSizeJiffies, TUint aSampleRate, TUint aBitDepth, TUint aChannels)
{
    return new MsgSilence(*this, aSizeJiffies, aSampleRate, aBitDepth, aChannels);
}

MsgHalt* MsgFactory::CreateMsgHalt()
{
    return new MsgHalt();
}

MsgQuit* MsgFactory::CreateMsgQuit()
{
    return new MsgQuit();
}

// ---------------------------------------------------------------- PipelineElement
#define OH_PASS_THROUGH(Type, Flag)                       \
    Msg* PipelineElement::ProcessMsg(Type* aMsg)          \
    {                                                     \
        CheckSupported(Flag);                             \
        return aMsg;                                      \
    }
OH_PASS_THROUGH(MsgMode, eMode)
OH_PASS_THROUGH(MsgTrack, eTrack)
OH_PASS_THROUGH(MsgDrain, eDrain)
OH_PASS_THROUGH(MsgDelay, eDelay)
OH_PASS_THROUGH(MsgEncodedStream, eEncodedStream)
OH_PASS_THROUGH(MsgStreamSegment, eStreamSegment)
OH_PASS_THROUGH(MsgAudioEncoded, eAudioEncoded)
OH_PASS_THROUGH(MsgMetaText, eMetatext)
OH_PASS_THROUGH(MsgStreamInterrupted, eStreamInterrupted)
OH_PASS_THROUGH(MsgHalt, eHalt)
OH_PASS_THROUGH(MsgFlush, eFlush)
OH_PASS_THROUGH(MsgWait, eWait)
OH_PASS_THROUGH(MsgDecodedStream, eDecodedStream)
OH_PASS_THROUGH(MsgAudioPcm, eAudioPcm)
OH_PASS_THROUGH(MsgAudioDsd, eAudioDsd)
OH_PASS_THROUGH(MsgSilence, eSilence)
OH_PASS_THROUGH(MsgPlayable, ePlayable)
OH_PASS_THROUGH(MsgQuit, eQuit)

// ---- MsgKind
namespace {
class KindVisitor : public IMsgProcessor {
public:
    MsgKind iKind = MsgKind::Quit;
private:
#define OH_KIND(Type, Value) Msg* ProcessMsg(Type* aMsg) override { iKind = MsgKind::Value; return aMsg; }
    OH_KIND(MsgMode, Mode) OH_KIND(MsgTrack, Track) OH_KIND(MsgDrain, Drain) OH_KIND(MsgDelay, Delay)
    OH_KIND(MsgEncodedStream, EncodedStream) OH_KIND(MsgStreamSegment, StreamSegment) OH_KIND(MsgAudioEncoded, AudioEncoded)
    OH_KIND(MsgMetaText, MetaText) OH_KIND(MsgStreamInterrupted, StreamInterrupted) OH_KIND(MsgHalt, Halt) OH_KIND(MsgFlush, Flush)
    OH_KIND(MsgWait, Wait) OH_KIND(MsgDecodedStream, DecodedStream) OH_KIND(MsgAudioPcm, AudioPcm) OH_KIND(MsgAudioDsd, AudioDsd)
    OH_KIND(MsgSilence, Silence) OH_KIND(MsgPlayable, Playable) OH_KIND(MsgQuit, Quit)
#undef OH_KIND
};

} // namespace

MsgKind KindOf(Msg* aMsg)
{
    KindVisitor v;
    (void)aMsg->Process(v);
    return v.iKind;
}

} // namespace Media
} // namespace OpenHome
