// RampGenerator.cpp -- see RampGenerator.h.
#include "RampGenerator.h"

#include <algorithm>
#include <cstring>

#include "../../include/ohgpu.h"
#include "FlywheelRamper.h"

namespace OpenHome {
namespace Media {

// ---------------------------------------------------------------- FlywheelInput
FlywheelInput::FlywheelInput(MsgFactory& aFactory, TUint aMaxJiffies)
    : iFactory(aFactory)
{
    (void)aMaxJiffies;                                   // the reference sizes a fixed buffer from it (:76-83); vectors grow
}

const Brx& FlywheelInput::Prepare(std::deque<MsgAudio*>& aAudio, TUint aJiffies, TUint aSampleRate, TUint aBitDepth, TUint aNumChannels)
{
    ASSERT(aNumChannels >= 1 && aNumChannels <= 10);
    iPacked.clear();
    iChannels = aNumChannels;
    iSubsampleBytes = aBitDepth / 8;
    PlayableBatch batch(iFactory);                       // :99-108: every queued message becomes a playable and is read here
    while (!aAudio.empty()) {
        MsgAudio* msg = aAudio.front();
        aAudio.pop_front();
        MsgPlayable* playable = nullptr;
        if (MsgAudioPcm* pcm = dynamic_cast<MsgAudioPcm*>(msg)) {
            playable = pcm->CreatePlayable();
        }
        else {
            MsgSilence* silence = dynamic_cast<MsgSilence*>(msg);
            ASSERT(silence != nullptr);
            playable = silence->CreatePlayable();
        }
        batch.Add(playable, *this);
    }
    batch.Run();
    // keep the newest aJiffies worth of frames, then de-interleave on the device (a11)
    const TUint frameBytes = iSubsampleBytes * aNumChannels;
    const TUint wanted = Jiffies::ToSamples(aJiffies, aSampleRate);
    const TUint have = (TUint)(iPacked.size() / frameBytes);
    const TUint frames = std::min(wanted, have);
    const TByte* newest = iPacked.data() + (size_t)(have - frames) * frameBytes;
    iPlanar.assign((size_t)frames * kSubsampleBytes * aNumChannels, 0);
    if (frames > 0) {
        ohgpu_fmt_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = OHGPU_FMT_UNPACK_PLANAR;
        d.channels = (uint8_t)aNumChannels;
        d.src_bits = (uint8_t)aBitDepth;
        d.n_frames = frames;
        d.dst_plane_stride = (uint64_t)frames * kSubsampleBytes;
        ohgpu_ctx* ctx = iFactory.Gpu();
        ohgpu_batch* b = nullptr;
        void *dSrc = nullptr, *dDst = nullptr;
        const size_t srcBytes = (size_t)frames * frameBytes;
        int err = ohgpu_fmt_batch_create(ctx, &d, 1, srcBytes, iPlanar.size(), &b);
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, srcBytes, &dSrc);
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, iPlanar.size(), &dDst);
        if (err == OHGPU_OK) err = ohgpu_memcpy_h2d(ctx, dSrc, newest, srcBytes, nullptr);
        if (err == OHGPU_OK) err = ohgpu_fmt_batch_run(ctx, b, dSrc, dDst, nullptr);
        if (err == OHGPU_OK) err = ohgpu_memcpy_d2h(ctx, iPlanar.data(), dDst, iPlanar.size(), nullptr);
        if (err == OHGPU_OK) err = ohgpu_stream_sync(ctx, nullptr);
        if (dSrc) ohgpu_free(ctx, dSrc);
        if (dDst) ohgpu_free(ctx, dDst);
        if (b) ohgpu_batch_destroy(ctx, b);
        ASSERT(err == OHGPU_OK);
    }
    iResult.Set(iPlanar.data(), (TUint)iPlanar.size());
    return iResult;
}

void FlywheelInput::ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes)
{
    ASSERT(aNumChannels == iChannels && aSubsampleBytes == iSubsampleBytes);
    iPacked.insert(iPacked.end(), aData.Ptr(), aData.Ptr() + aData.Bytes());
}

void FlywheelInput::ProcessSilence(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes)
{
    ProcessFragment(aData, aNumChannels, aSubsampleBytes);   // :154-157
}

// ---------------------------------------------------------------- RampGenerator
RampGenerator::RampGenerator(MsgFactory& aFactory, TUint aInputJiffies, TUint aRampJiffies)
    : iFactory(aFactory)
    , iInputJiffies(aInputJiffies)
    , iRampJiffies(aRampJiffies)
{
}

RampGenerator::~RampGenerator()
{
    // the reference asserts the queue was drained (:227); a destructor must not throw, so undelivered messages are released
    while (!iQueue.empty()) {
        iQueue.front()->RemoveRef();
        iQueue.pop_front();
    }
}

void RampGenerator::Start(const Brx& aRecentAudio, TUint aSampleRate, TUint aNumChannels, TUint aBitDepth, TUint aCurrentRampValue)
{
    ASSERT(aNumChannels >= 1 && aNumChannels <= 10);
    ASSERT(aBitDepth == 8 || aBitDepth == 16 || aBitDepth == 24 || aBitDepth == 32);   // ASSERTS() in ProcessFragment's default, :322-324
    const TUint inSamples = Jiffies::ToSamples(iInputJiffies, aSampleRate);
    const TUint channelBytes = aRecentAudio.Bytes() / aNumChannels;                       // InitChannels, FlywheelRamper.cpp:71
    ASSERT(channelBytes >= inSamples * 4);
    const TUint outFrames = Jiffies::ToSamples(iRampJiffies, aSampleRate);              // genSampleCount, :240
    const TUint blockFrames = Jiffies::ToSamples(FlywheelRamperManager::kMaxOutputJiffiesBlockSize, aSampleRate);
    TUint remainingRampSize = Jiffies::PerSample(aSampleRate) * outFrames;              // :241
    if (outFrames == 0) {
        return;
    }
    const size_t rampBytes = (size_t)outFrames * aNumChannels * 4;
    const TUint outSub = aBitDepth / 8;
    const size_t packedBytes = (size_t)outFrames * aNumChannels * outSub;

    // ---- device chain: training audio -> flywheel (32-bit blocks) -> a12 pack to the stream's depth ----
    ohgpu_ctx* ctx = iFactory.Gpu();
    ohgpu_flywheel_desc fd;
    memset(&fd, 0, sizeof(fd));
    fd.channel_bytes = channelBytes;
    fd.in_samples = inSamples;
    fd.out_frames = outFrames;
    fd.block_frames = blockFrames;
    fd.sample_rate = aSampleRate;
    fd.channels = aNumChannels;
    std::vector<ohgpu_msg_desc> blocks;                  // one 1 ms block = one message, as RenderChannels hands them over
    for (TUint done = 0; done < outFrames; done += blockFrames) {
        ohgpu_msg_desc m;
        memset(&m, 0, sizeof(m));
        m.src_offset = (uint64_t)done * aNumChannels * 4;
        m.dst_offset = (uint64_t)done * aNumChannels * outSub;
        m.n_frames = std::min(blockFrames, outFrames - done);
        m.ramp_start = m.ramp_end = OHGPU_RAMP_MAX;
        m.attenuation = OHGPU_UNITY_ATTENUATION;
        m.channels = (uint8_t)aNumChannels;
        m.src_bits = 32; m.src_endian = OHGPU_ENDIAN_BIG;
        m.dst_bits = (uint8_t)aBitDepth; m.dst_endian = OHGPU_ENDIAN_BIG;
        m.flags = aBitDepth == 32 ? OHGPU_FLAG_ZERO_LSB32 : 0;                            // "discard least significant byte", :311-320
        blocks.push_back(m);
    }
    std::vector<TByte> packed(packedBytes);
    ohgpu_batch *fb = nullptr, *pb = nullptr;
    void *dTrain = nullptr, *dRamp = nullptr, *dPacked = nullptr;
    int err = ohgpu_flywheel_batch_create(ctx, &fd, 1, aRecentAudio.Bytes(), rampBytes, &fb);
    if (err == OHGPU_OK) err = ohgpu_pcm_batch_create(ctx, blocks.data(), blocks.size(), rampBytes, packedBytes, &pb);
    if (err == OHGPU_OK) err = ohgpu_malloc(ctx, aRecentAudio.Bytes(), &dTrain);
    if (err == OHGPU_OK) err = ohgpu_malloc(ctx, rampBytes, &dRamp);
    if (err == OHGPU_OK) err = ohgpu_malloc(ctx, packedBytes, &dPacked);
    if (err == OHGPU_OK) err = ohgpu_memcpy_h2d(ctx, dTrain, aRecentAudio.Ptr(), aRecentAudio.Bytes(), nullptr);
    if (err == OHGPU_OK) err = ohgpu_flywheel_batch_run(ctx, fb, dTrain, dRamp, nullptr);
    if (err == OHGPU_OK) err = ohgpu_pcm_batch_run(ctx, pb, dRamp, dPacked, nullptr);   // same stream: ordered after the flywheel
    if (err == OHGPU_OK) err = ohgpu_memcpy_d2h(ctx, packed.data(), dPacked, packedBytes, nullptr);
    if (err == OHGPU_OK) err = ohgpu_stream_sync(ctx, nullptr);
    if (dTrain) ohgpu_free(ctx, dTrain);
    if (dRamp) ohgpu_free(ctx, dRamp);
    if (dPacked) ohgpu_free(ctx, dPacked);
    if (fb) ohgpu_batch_destroy(ctx, fb);
    if (pb) ohgpu_batch_destroy(ctx, pb);
    ASSERT(err == OHGPU_OK);

    // ---- EndBlock, :337-352: one message per block, ramping down from where the stream's ramp stood ----
    TUint current = aCurrentRampValue;
    for (const ohgpu_msg_desc& m : blocks) {
        MsgAudioPcm* audio = iFactory.CreateMsgAudioPcm(Brn(packed.data() + m.dst_offset, m.n_frames * aNumChannels * outSub), aNumChannels,
                                                        aSampleRate, aBitDepth, AudioDataEndian::Big, MsgAudioPcm::kTrackOffsetInvalid);
        if (current == Ramp::kMin) {
            audio->SetMuted();
        }
        else {
            MsgAudio* split = nullptr;
            current = audio->SetRamp(current, remainingRampSize, Ramp::EDown, split);
            ASSERT(split == nullptr);
        }
        iQueue.push_back(audio);
    }
}

TBool RampGenerator::TryGetAudio(Msg*& aMsg)
{
    if (iQueue.empty()) {
        return false;
    }
    aMsg = iQueue.front();
    iQueue.pop_front();
    return true;
}

} // namespace Media
} // namespace OpenHome
