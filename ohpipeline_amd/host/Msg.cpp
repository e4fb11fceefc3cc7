// Msg.cpp -- host-side message model for the PCM hot path (see Msg.h for the reference lines each class follows).
#include "Msg.h"

#include <algorithm>
#include <cstring>
#include <deque>
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

// The three functions below are the message algebra of Msg.cpp:1949-2046 restated: what they must DO is fixed by the reference
// (its suites run against them, tests/cpp/test_host.cpp), how they are written is this file's own.
MsgAudio* MsgAudio::Split(TUint aJiffies)
{
    ASSERT(aJiffies != 0 && aJiffies < iSize);                     // both ends keep audio
    MsgAudio* tail = Allocate();
    tail->iOffset = iOffset + aJiffies;
    tail->iSize = iSize - aJiffies;
    tail->iRamp = iRamp.IsEnabled() ? iRamp.Split(aJiffies, iSize) : Media::Ramp();   // (Ramp::Split shortens iRamp to the head's share)
    iSize = aJiffies;
    SplitCompleted(*tail);
    return tail;
}

MsgAudio* MsgAudio::Clone()
{
    MsgAudio* twin = Allocate();
    twin->iOffset = iOffset;
    twin->iSize = iSize;
    twin->iRamp = iRamp;
    return twin;
}

TUint MsgAudio::SetRamp(TUint aStart, TUint& aRemainingDuration, Ramp::EDirection aDirection, MsgAudio*& aSplit)
{
    ASSERT(aDirection == Ramp::EUp || aDirection == Ramp::EDown);
    aSplit = nullptr;
    const bool fadingOut = aDirection == Ramp::EDown;
    if (iRamp.IsEnabled() && iRamp.Direction() == Ramp::EMute) {
        // muted audio stays muted, and a fade-out that has reached silence has nothing left to do (Msg.cpp:1997-2002)
        if (fadingOut) aRemainingDuration = 0;
        return iRamp.End();
    }
    // Ramp::Set merges the requested ramp with the one already here.  Where the two lines cross it reports the position and
    // the ramp of the part behind it; that part becomes a message of its own.
    Media::Ramp behindCrossing;
    TUint crossAt = 0;
    const bool crossed = iRamp.Set(aStart, iSize, aRemainingDuration, aDirection, behindCrossing, crossAt);
    TUint handedBack = 0;
    if (crossed && crossAt == 0) {
        iRamp = behindCrossing;                                    // the other line is the lower one from the first jiffy on
    }
    else if (crossed && crossAt != iSize) {
        const Media::Ramp head = iRamp;                            // (Split rescales both halves; the pair Set computed is the one that counts)
        aSplit = Split(crossAt);
        iRamp = head;
        aSplit->iRamp = behindCrossing;
        if (!fadingOut && behindCrossing.Direction() != aDirection) {
            handedBack = aSplit->iSize;                            // the tail still runs the old way: its length is not progress (Msg.cpp:2032-2034)
        }
    }
    const TUint target = fadingOut ? Ramp::kMin : Ramp::kMax;
    aRemainingDuration = (iRamp.End() == target) ? 0 : aRemainingDuration - iSize + handedBack;
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
    // same format, the audio that directly follows this one, no ramp on either (Msg.cpp:2167-2172)
    ASSERT(aMsg->iSampleRate == iSampleRate && aMsg->iBitDepth == iBitDepth && aMsg->iNumChannels == iNumChannels);
    ASSERT(aMsg->iTrackOffset == iTrackOffset + Jiffies() && !iRamp.IsEnabled() && !aMsg->iRamp.IsEnabled());
    ASSERT(iAudioData != nullptr && aMsg->iAudioData != nullptr);
    iAudioData->Aggregate(*aMsg->iAudioData);
    iSize += aMsg->Jiffies();
    aMsg->RemoveRef();
}

MsgPlayable* MsgAudioPcm::CreatePlayable()
{
    // Msg.cpp:2234-2262 in frames: the window [iOffset, iOffset + iSize) of the audio, both ends rounded DOWN to whole samples
    // (Jiffies::ToBytes), is what a playable reads; a muted message plays silence of that length instead.
    const TUint perSample = Jiffies::PerSample(iSampleRate);
    const TUint frameBytes = (iBitDepth / 8) * iNumChannels;
    const TUint firstFrame = iOffset / perSample;
    const TUint endFrame = static_cast<TUint>((static_cast<TUint64>(iOffset) + iSize) / perSample);
    PlayableWork work;
    work.sampleRate = iSampleRate;
    work.bitDepth = iBitDepth;
    work.channels = iNumChannels;
    work.frames = endFrame - firstFrame;
    work.sizeBytes = work.frames * frameBytes;
    work.silence = iRamp.Direction() == Ramp::EMute;               // (and then no ramp, no attenuation, no source)
    if (!work.silence) {
        work.offsetBytes = firstFrame * frameBytes;
        work.attenuation = iAttenuation;
        work.ramp = iRamp;
        work.resampled = iResampled != nullptr;
        if (work.resampled) {
            work.stream = iResampled;
            work.outFrame0 = iResampledFrame0 + firstFrame;
        }
        else {
            work.audio = iAudioData;
        }
    }
    MsgPlayable* playable = new MsgPlayable(iFactory, work, iSize);
    RemoveRef();                                                   // the playable stands in for this message from here on
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
    // Msg.cpp:2591-2624: the driver cuts a playable to its period.  The head keeps aBytes; the ramp is divided in proportion to
    // BYTES here (jiffies in MsgAudio::Split).
    ASSERT(aBytes != 0 && aBytes <= iWork.sizeBytes);
    if (aBytes == iWork.sizeBytes) {
        return nullptr;                                            // nothing behind the cut
    }
    const TUint frameBytes = (iWork.bitDepth / 8) * iWork.channels;
    const TUint headFrames = aBytes / frameBytes;
    const TUint headJiffies = headFrames * Jiffies::PerSample(iWork.sampleRate);
    PlayableWork tail = iWork;
    tail.offsetBytes += aBytes;
    tail.sizeBytes -= aBytes;
    tail.frames = tail.sizeBytes / frameBytes;
    tail.outFrame0 += headFrames;
    tail.ramp = iWork.ramp.IsEnabled() ? iWork.ramp.Split(aBytes, iWork.sizeBytes) : Media::Ramp();
    MsgPlayable* rest = new MsgPlayable(iFactory, tail, iJiffies - headJiffies);
    iWork.sizeBytes = aBytes;
    iWork.frames = headFrames;
    iJiffies = headJiffies;
    return rest;
}

void MsgPlayable::Read(IPcmProcessor& aProcessor)
{
    PlayableBatch batch(iFactory);
    AddRef();                        // the batch releases one reference; Read() leaves ownership with the caller
    batch.Add(this, aProcessor);
    batch.Run();
}

// ---------------------------------------------------------------- PlayableBatch
struct PlayableBatch::WindowRun {
    SampleRateConverterStream* stream;
    TUint64 firstFrame;              // the union window of the run's items: input frames [firstFrame, firstFrame + frames)
    TUint frames;
    TUint64 nextOut;                 // the output frame an item must start at to join the run
    TUint64 srcOffset;               // of the window, in its group's part of the source arena
};

struct PlayableBatch::Group {
    const SrcFilter* filter = nullptr;                    // nullptr: the plain audio (one ohgpu_pcm_process_host)
    std::vector<size_t> items;
    std::vector<ohgpu_msg_desc> pcm;
    std::vector<ohgpu_src_msg_desc> src;
    std::vector<WindowRun> runs;
    std::vector<size_t> runOf;                            // per resampled item: its run
    TUint64 srcBase = 0, srcBytes = 0, dstBase = 0, dstBytes = 0;
    void Clear() { items.clear(); pcm.clear(); src.clear(); runs.clear(); runOf.clear(); srcBase = srcBytes = dstBase = dstBytes = 0; }
};

struct PlayableBatch::Scratch {
    std::deque<Group> groups;                             // [0] the plain audio, [1 + k] the k-th filter met (a deque: Take's references stay valid)
    size_t used = 0;
    std::map<const DecodedAudio*, TUint64> audioBase;
    Group& Take(const SrcFilter* aFilter)
    {
        for (size_t g = 0; g < used; g++) if (groups[g].filter == aFilter) return groups[g];
        if (used == groups.size()) groups.emplace_back();
        Group& g = groups[used++];
        g.Clear();
        g.filter = aFilter;
        return g;
    }
};

PlayableBatch::PlayableBatch(MsgFactory& aFactory)
    : iFactory(aFactory)
    , iScratch(new Scratch())
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
    iItems.push_back({aPlayable, &aProcessor, 0, 0});
}

static uint8_t GpuEndian(AudioDataEndian aEndian)
{
    return aEndian == AudioDataEndian::Little ? OHGPU_ENDIAN_LITTLE : OHGPU_ENDIAN_BIG;
}

void PlayableBatch::Run()
{
    ohgpu_ctx* ctx = iFactory.Gpu();
    Scratch& sc = *iScratch;
    sc.used = 0;
    sc.audioBase.clear();
    // ---- who goes with whom: the plain audio in one call, the rate-converted audio in one call per filter.  Within a group the
    // outputs lie back to back in the order the items came (they tile the group's span of the destination: one copy back), and
    // of the sources each distinct DecodedAudio lies there once, each run of consecutive outputs of a stream as ONE window ----
    Group& plain = sc.Take(nullptr);
    for (size_t i = 0; i < iItems.size(); i++) {
        const PlayableWork& w = iItems[i].playable->Work();
        iItems[i].outBits = (iOutBits == 0) ? w.bitDepth : iOutBits;
        if (w.frames == 0) {
            continue;
        }
        Group& g = w.resampled ? sc.Take(&w.stream->Filter()) : plain;
        g.items.push_back(i);
        if (w.resampled) {
            TUint64 first = 0;
            TUint frames = 0;
            w.stream->Window(w.outFrame0, w.frames, first, frames);
            if (!g.runs.empty() && g.runs.back().stream == w.stream.get() && g.runs.back().nextOut == w.outFrame0) {
                WindowRun& r = g.runs.back();                              // follows on: the window grows, the history is not repeated
                r.frames = (TUint)(first + frames - r.firstFrame);
                r.nextOut += w.frames;
            }
            else {
                g.runs.push_back({w.stream.get(), first, frames, w.outFrame0 + w.frames, 0});
            }
            g.runOf.push_back(g.runs.size() - 1);
        }
        else if (!w.silence && sc.audioBase.find(w.audio.get()) == sc.audioBase.end()) {
            sc.audioBase.emplace(w.audio.get(), g.srcBytes);
            g.srcBytes += (w.audio->Bytes() + 15u) & ~15u;
        }
    }
    TUint64 srcTotal = 0, dstTotal = 0;
    for (size_t k = 0; k < sc.used; k++) {
        Group& g = sc.groups[k];
        for (WindowRun& r : g.runs) {
            r.srcOffset = g.srcBytes;
            g.srcBytes += ((TUint64)r.frames * r.stream->FrameBytes() + 15u) & ~(TUint64)15u;
        }
        for (size_t i : g.items) {
            const PlayableWork& w = iItems[i].playable->Work();
            iItems[i].outOffset = g.dstBytes;                            // (within the group, for now)
            g.dstBytes += (TUint64)w.frames * w.channels * (iItems[i].outBits / 8);
        }
        g.srcBase = srcTotal;
        g.dstBase = dstTotal;
        srcTotal += (g.srcBytes + 63u) & ~(TUint64)63u;
        dstTotal += (g.dstBytes + 63u) & ~(TUint64)63u;
    }
    TByte* src = nullptr;
    TByte* dst = nullptr;
    iFactory.ReserveArena((size_t)srcTotal, (size_t)dstTotal, src, dst);
    // ---- the descriptors, and the sources into the arena ----
    for (const auto& kv : sc.audioBase) {
        memcpy(src + plain.srcBase + kv.second, kv.first->Ptr(0), kv.first->Bytes());
    }
    for (size_t k = 0; k < sc.used; k++) {
        Group& g = sc.groups[k];
        for (const WindowRun& r : g.runs) {
            r.stream->CopyFrames(r.firstFrame, r.frames, src + g.srcBase + r.srcOffset);
        }
        size_t nthResampled = 0;
        for (size_t i : g.items) {
            const PlayableWork& w = iItems[i].playable->Work();
            if (w.resampled) {
                const WindowRun& r = g.runs[g.runOf[nthResampled++]];
                ohgpu_src_msg_desc d;
                memset(&d, 0, sizeof(d));
                d.src_offset = r.srcOffset;
                d.src_frame0 = r.firstFrame;
                d.src_frames = r.frames;
                d.out_frame0 = w.outFrame0;
                d.dst_offset = iItems[i].outOffset;
                d.n_frames = w.frames;
                d.ramp_start = (uint16_t)w.ramp.Start();
                d.ramp_end = (uint16_t)w.ramp.End();
                d.attenuation = OHGPU_UNITY_ATTENUATION;
                d.channels = (uint8_t)w.channels;
                d.src_bits = (uint8_t)w.stream->SourceBitDepth();
                d.src_endian = GpuEndian(w.stream->SourceEndian());
                d.dst_bits = (uint8_t)iItems[i].outBits;
                d.dst_endian = GpuEndian(iOutEndian);
                d.flags = w.ramp.IsEnabled() ? OHGPU_FLAG_RAMP : 0;
                g.src.push_back(d);
            }
            else {
                ohgpu_msg_desc d;
                memset(&d, 0, sizeof(d));
                d.dst_offset = iItems[i].outOffset;
                d.n_frames = w.frames;
                d.ramp_start = (uint16_t)w.ramp.Start();
                d.ramp_end = (uint16_t)w.ramp.End();
                d.attenuation = (uint16_t)w.attenuation;
                d.channels = (uint8_t)w.channels;
                d.src_bits = (uint8_t)w.bitDepth;
                d.dst_bits = (uint8_t)iItems[i].outBits;
                d.dst_endian = GpuEndian(iOutEndian);
                d.src_endian = OHGPU_ENDIAN_BIG;
                if (w.silence) {
                    d.flags = OHGPU_FLAG_SILENCE;
                }
                else {
                    d.src_offset = sc.audioBase[w.audio.get()] + w.offsetBytes;
                    d.src_endian = GpuEndian(w.audio->Endian());
                    d.flags = w.ramp.IsEnabled() ? OHGPU_FLAG_RAMP : 0;
                }
                g.pcm.push_back(d);
            }
            iItems[i].outOffset += g.dstBase;                            // (from here on: in the arena)
        }
        // ---- the group's one call: its part of the source arena in, its span of the destination back ----
        if (!g.pcm.empty()) {
            const int err = ohgpu_pcm_process_host(ctx, g.pcm.data(), g.pcm.size(), src + g.srcBase, g.srcBytes, dst + g.dstBase, g.dstBytes);
            ASSERT(err == OHGPU_OK);
        }
        if (!g.src.empty()) {
            const int err = ohgpu_src_process_host(ctx, g.filter->handle, g.src.data(), g.src.size(), src + g.srcBase, g.srcBytes,
                                                   dst + g.dstBase, g.dstBytes);
            ASSERT(err == OHGPU_OK);
        }
    }
    // ---- deliver, message by message, with the reference's callback sequence (Msg.cpp:2646-2653, 2753-2786, 2874-2893) ----
    for (size_t i = 0; i < iItems.size(); i++) {
        MsgPlayable* playable = iItems[i].playable;
        IPcmProcessor& proc = *iItems[i].processor;
        const PlayableWork& w = playable->Work();
        const TUint subsampleBytes = iItems[i].outBits / 8;
        const TUint outFrameBytes = subsampleBytes * w.channels;
        proc.BeginBlock();
        if (w.frames > 0) {
            const TByte* out = dst + iItems[i].outOffset;
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
        for (auto& kv : iFilters) {
            ohgpu_src_destroy(iCtx, kv.second.handle);
        }
        if (iArenaSrc != nullptr) ohgpu_free_host(iCtx, iArenaSrc);
        if (iArenaDst != nullptr) ohgpu_free_host(iCtx, iArenaDst);
        ohgpu_shutdown(iCtx);
    }
}

ohgpu_ctx* MsgFactory::Gpu() const
{
    ASSERT(iCtx != nullptr);        // a control-plane-only factory (device < 0) cannot read audio
    return iCtx;
}

const SrcFilter& MsgFactory::SharedFilter(TUint aRateIn, TUint aRateOut, TUint aTapsPerPhase, double aBeta, double aPassHz)
{
    std::lock_guard<std::mutex> hold(iFilterLock);
    const auto key = std::make_tuple(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz);
    auto it = iFilters.find(key);
    if (it == iFilters.end()) {
        // designed once per conversion, on the host (Kaiser-windowed sinc, Q28: DESIGN.md section 4), uploaded once
        SrcFilter f;
        uint32_t L = 0, M = 0;
        int err = ohgpu_src_design(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz, nullptr, 0, &L, &M);
        ASSERT(err == OHGPU_OK);
        std::vector<int32_t> coef((size_t)L * aTapsPerPhase);
        err = ohgpu_src_design(aRateIn, aRateOut, aTapsPerPhase, aBeta, aPassHz, coef.data(), coef.size(), &L, &M);
        ASSERT(err == OHGPU_OK);
        err = ohgpu_src_create(Gpu(), L, M, aTapsPerPhase, coef.data(), &f.handle);
        ASSERT(err == OHGPU_OK);
        f.L = L;
        f.M = M;
        f.T = aTapsPerPhase;
        it = iFilters.emplace(key, f).first;                               // (std::map: the reference handed out stays valid)
    }
    return it->second;
}

TUint MsgFactory::FilterCount() const
{
    std::lock_guard<std::mutex> hold(iFilterLock);
    return (TUint)iFilters.size();
}

void MsgFactory::ReserveArena(size_t aSrcBytes, size_t aDstBytes, TByte*& aSrc, TByte*& aDst)
{
    auto grow = [this](TByte*& aBuf, size_t& aHave, size_t aWant) {
        if (aHave >= aWant && aBuf != nullptr) {
            return;
        }
        if (aBuf != nullptr) {
            ohgpu_free_host(Gpu(), aBuf);
            aBuf = nullptr;
        }
        size_t bytes = aWant + aWant / 2;
        if (bytes < 65536) bytes = 65536;
        void* p = nullptr;
        const int err = ohgpu_malloc_host(Gpu(), bytes, &p);
        ASSERT(err == OHGPU_OK);
        aBuf = (TByte*)p;
        aHave = bytes;
    };
    grow(iArenaSrc, iArenaSrcBytes, aSrcBytes);
    grow(iArenaDst, iArenaDstBytes, aDstBytes);
    aSrc = iArenaSrc;
    aDst = iArenaDst;
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
