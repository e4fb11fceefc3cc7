// FlywheelRamper.cpp -- see FlywheelRamper.h.  No audio arithmetic here: the host lays requests out and replays the
// reference's output callbacks.
#include "FlywheelRamper.h"

#include <algorithm>
#include <cstring>

#include "../../include/ohgpu.h"

namespace OpenHome {
namespace Media {

static const TUint kMaxChannelCount = 10;                // FlywheelRamper.cpp:19
static const TUint kMaxSampleRate = 384000;              // :20

const TUint FlywheelRamperManager::kMaxOutputJiffiesBlockSize = Jiffies::kPerMs;   // :22

TUint FlywheelRamper::DecimationFactor(TUint aSampleRate)
{
    switch (aSampleRate) {
    case 192000: case 176400: return 4;
    case 88200: case 96000: return 2;
    default: return 1;
    }
}

FlywheelRamperBatch::FlywheelRamperBatch(MsgFactory& aFactory)
    : iFactory(aFactory)
{
}

void FlywheelRamperBatch::Add(IPcmProcessor& aOutput, const Brx& aSamples, TUint aSampleRate, TUint aChannelCount,
                              TUint aInputJiffies, TUint aOutputJiffies)
{
    ASSERT(aChannelCount >= 1 && aChannelCount <= kMaxChannelCount);
    ASSERT(aSampleRate <= kMaxSampleRate);                                              // :178
    Item it;
    it.output = &aOutput;
    it.sampleRate = aSampleRate;
    it.channels = aChannelCount;
    it.channelBytes = aSamples.Bytes() / aChannelCount;                                 // InitChannels, :71
    it.inSamples = Jiffies::ToSamples(aInputJiffies, aSampleRate);
    ASSERT(it.channelBytes >= it.inSamples * FlywheelRamper::kBytesPerSample);          // :180
    it.outFrames = Jiffies::ToSamples(aOutputJiffies, aSampleRate);
    it.blockFrames = Jiffies::ToSamples(FlywheelRamperManager::kMaxOutputJiffiesBlockSize, aSampleRate);
    it.srcOffset = iSrc.size();
    iSrc.insert(iSrc.end(), aSamples.Ptr(), aSamples.Ptr() + (size_t)it.channelBytes * aChannelCount);
    iItems.push_back(it);
}

void FlywheelRamperBatch::Run()
{
    std::vector<ohgpu_flywheel_desc> descs(iItems.size());
    std::vector<TUint64> outOffset(iItems.size());
    TUint64 dstBytes = 0;
    for (size_t i = 0; i < iItems.size(); i++) {
        const Item& it = iItems[i];
        ohgpu_flywheel_desc& d = descs[i];
        memset(&d, 0, sizeof(d));
        d.src_offset = it.srcOffset;
        d.channel_bytes = it.channelBytes;
        d.dst_offset = outOffset[i] = dstBytes;
        d.in_samples = it.inSamples;
        d.out_frames = it.outFrames;
        d.block_frames = it.blockFrames;
        d.sample_rate = it.sampleRate;
        d.channels = it.channels;
        dstBytes += (TUint64)it.outFrames * it.channels * 4;
    }
    std::vector<TByte> dst(dstBytes);
    if (!descs.empty()) {
        const int err = ohgpu_flywheel_process_host(iFactory.Gpu(), descs.data(), descs.size(), iSrc.data(), iSrc.size(),
                                                    dst.data(), dst.size());
        ASSERT(err == OHGPU_OK);
    }
    for (size_t i = 0; i < iItems.size(); i++) {                                        // Ramp, :52-63 + RenderChannels' tail, :124-130
        const Item& it = iItems[i];
        const TByte* out = dst.data() + outOffset[i];
        for (TUint done = 0; done < it.outFrames; done += it.blockFrames) {
            const TUint n = std::min(it.blockFrames, it.outFrames - done);
            it.output->BeginBlock();
            it.output->ProcessFragment(Brn(out + (size_t)done * it.channels * 4, n * it.channels * 4), it.channels, 4);
            it.output->EndBlock();
        }
    }
    iItems.clear();
    iSrc.clear();
}

FlywheelRamperManager::FlywheelRamperManager(MsgFactory& aFactory, IPcmProcessor& aOutput, TUint aInputJiffies, TUint aOutputJiffies)
    : iFactory(aFactory)
    , iOutput(aOutput)
    , iInputJiffies(aInputJiffies)
    , iOutputJiffies(aOutputJiffies)
{
}

void FlywheelRamperManager::Ramp(const Brx& aSamples, TUint aSampleRate, TUint aChannelCount)
{
    FlywheelRamperBatch batch(iFactory);
    batch.Add(iOutput, aSamples, aSampleRate, aChannelCount, iInputJiffies, iOutputJiffies);
    batch.Run();
}

} // namespace Media
} // namespace OpenHome
