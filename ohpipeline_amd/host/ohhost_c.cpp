// ohhost_c.cpp -- C entry points over the host adapter, for scripted callers (tests, bench.py).
// Exceptions never cross this boundary: AssertionFailed -> -1, SampleRateInvalid -> -2.
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "../../include/ohgpu.h"
#include "Msg.h"
#include "Ramp.h"
#include "SampleRateConverter.h"

using namespace OpenHome;
using namespace OpenHome::Media;

#define OHHOST_TRY(...)                                    \
    try { __VA_ARGS__; }                                   \
    catch (const AssertionFailed&) { return -1; }          \
    catch (const SampleRateInvalid&) { return -2; }        \
    catch (...) { return -3; }

struct ohhost_ramp {
    uint32_t start, end, direction, enabled;
};

static void ToC(const Ramp& r, ohhost_ramp* o)
{
    o->start = r.Start(); o->end = r.End(); o->direction = (uint32_t)r.Direction(); o->enabled = r.IsEnabled() ? 1 : 0;
}

extern "C" {

int ohhost_jiffies_per_sample(uint32_t rate)
{
    OHHOST_TRY(return (int)Jiffies::PerSample(rate));
}

// Ramp::Set on a ramp that currently holds (cur) [pass enabled = 0 for a fresh ramp]; outputs the resulting
// ramp, the split ramp and split position.  Returns 1 iff split is set.
int ohhost_ramp_set(const ohhost_ramp* cur, uint32_t start, uint32_t fragment, uint32_t remaining, uint32_t direction,
                    ohhost_ramp* out, ohhost_ramp* split, uint32_t* split_pos)
{
    OHHOST_TRY({
        Ramp r, s;
        if (cur != nullptr && cur->enabled) {
            // rebuild the existing ramp through the public surface: a fresh Set over a degenerate duration
            Ramp tmp; TUint pos;
            if (cur->direction == Ramp::EMute) { r.SetMuted(); }
            else if (cur->start == cur->end) { return -1; }
            else {
                const Ramp::EDirection d = cur->start > cur->end ? Ramp::EDown : Ramp::EUp;
                const TUint togo = (d == Ramp::EDown) ? cur->start : Ramp::kMax - cur->start;
                const TUint span = (d == Ramp::EDown) ? cur->start - cur->end : cur->end - cur->start;
                // choose fragment/duration so that ceil(togo*f/D) == span: f = span, D = togo
                r.Set(cur->start, span, togo, d, tmp, pos);
            }
        }
        TUint pos = 0;
        const TBool has = r.Set(start, fragment, remaining, (Ramp::EDirection)direction, s, pos);
        ToC(r, out);
        ToC(s, split);
        *split_pos = pos;
        return has ? 1 : 0;
    });
}

// Ramp endpoints for every message of one stream that ramps up from silence over its first up_jiffies and
// down to silence over its last down_jiffies -- what Ramper::ProcessAudio (Ramper.cpp:114-134) does at a stream
// start and Stopper does at its end, each driving MsgAudio::SetRamp (Msg.cpp:1989-2046) message by message.
// sizes[i] = message i's length in jiffies.  flags[i] = 1 when message i carries an enabled ramp.
int ohhost_stream_ramp_schedule(const uint32_t* sizes, uint32_t n, uint32_t up_jiffies, uint32_t down_jiffies,
                                uint8_t* flags, uint16_t* starts, uint16_t* ends)
{
    OHHOST_TRY({
        for (uint32_t i = 0; i < n; i++) { flags[i] = 0; starts[i] = (uint16_t)Ramp::kMax; ends[i] = (uint16_t)Ramp::kMax; }
        auto run = [&](uint32_t first, uint32_t last, TUint current, Ramp::EDirection dir) {
            TUint remaining = 0;
            for (uint32_t i = first; i < last; i++) remaining += sizes[i];
            for (uint32_t i = first; i < last && remaining != 0; i++) {
                Ramp ramp, split;
                TUint splitPos;
                const TBool hasSplit = ramp.Set(current, sizes[i], remaining, dir, split, splitPos);
                ASSERT(!hasSplit);                      // a fresh message cannot intersect an older ramp
                remaining -= sizes[i];                   // MsgAudio::SetRamp, Msg.cpp:2031
                if (dir == Ramp::EDown && ramp.End() == Ramp::kMin) remaining = 0;       // Msg.cpp:2037-2043
                else if (dir == Ramp::EUp && ramp.End() == Ramp::kMax) remaining = 0;
                current = ramp.End();
                flags[i] = 1;
                starts[i] = (uint16_t)ramp.Start();
                ends[i] = (uint16_t)ramp.End();
            }
        };
        uint32_t upLast = 0;
        for (TUint acc = 0; upLast < n && acc < up_jiffies; upLast++) acc += sizes[upLast];
        uint32_t downFirst = n;
        for (TUint acc = 0; downFirst > upLast && acc < down_jiffies; downFirst--) acc += sizes[downFirst - 1];
        run(0, upLast, Ramp::kMin, Ramp::EUp);
        run(downFirst, n, Ramp::kMax, Ramp::EDown);
        return 0;
    });
}

}  // extern "C"

// ---- a driver thread's period over many rate-converted streams, for scripted callers (bench.py's cadence.adapter): `lanes`
// chains of SampleRateConverter -> CreatePlayable (PreDriver.cpp:115-133's one line) behind one factory = one GPU context, read
// with ONE PlayableBatch::Run per tick, the way AnimatorBasic.cpp:77-142 reads one.
namespace {

class CopyOut : public IPcmProcessor {
public:
    TByte* iDst = nullptr;
    uint32_t iBytes = 0, iCap = 0;
    void BeginBlock() override {}
    void ProcessFragment(const Brx& aData, TUint, TUint) override { Take(aData); }
    void ProcessSilence(const Brx& aData, TUint, TUint) override { Take(aData); }
    void EndBlock() override {}
    void Flush() override {}
private:
    void Take(const Brx& aData)
    {
        ASSERT(iBytes + aData.Bytes() <= iCap);
        memcpy(iDst + iBytes, aData.Ptr(), aData.Bytes());
        iBytes += aData.Bytes();
    }
};

struct LiveLane : public IPipelineElementUpstream {
    LiveLane(MsgFactory& aFactory, TUint aRateOut, TUint aTaps) : iSrc(aFactory, *this, aRateOut, aTaps) {}
    Msg* Pull() override { ASSERT(!iPending.empty()); Msg* m = iPending.front(); iPending.pop_front(); return m; }
    SampleRateConverter iSrc;
    std::deque<Msg*> iPending;
    CopyOut iSink;
};

}  // namespace

struct ohhost_live {
    std::unique_ptr<MsgFactory> factory;
    std::unique_ptr<PlayableBatch> batch;
    std::vector<std::unique_ptr<LiveLane>> lanes;
    TUint rateIn = 0, channels = 0, bits = 0;
    AudioDataEndian endian = AudioDataEndian::Little;
};

extern "C" {

int ohhost_live_create(int device, uint32_t lanes, uint32_t rate_in, uint32_t rate_out, uint32_t channels, uint32_t bits,
                       uint32_t little_endian, uint32_t out_bits, ohhost_live** out)
{
    OHHOST_TRY({
        std::unique_ptr<ohhost_live> live(new ohhost_live());
        live->factory.reset(new MsgFactory(device));
        live->batch.reset(new PlayableBatch(*live->factory));
        live->batch->SetOutputFormat(out_bits, AudioDataEndian::Big);
        live->rateIn = rate_in; live->channels = channels; live->bits = bits;
        live->endian = little_endian ? AudioDataEndian::Little : AudioDataEndian::Big;
        for (uint32_t l = 0; l < lanes; l++) {
            live->lanes.emplace_back(new LiveLane(*live->factory, rate_out, rate_in == 2 * rate_out ? 64 : 32));
            DecodedStreamInfo info;
            info.iStreamId = l + 1; info.iBitDepth = bits; info.iSampleRate = rate_in; info.iNumChannels = channels;
            live->lanes.back()->iPending.push_back(live->factory->CreateMsgDecodedStream(info));
        }
        *out = live.release();
        return 0;
    });
}

// One driver period.  Lane l is fed `frames` input frames from input + l * in_lane_stride; whatever audio that makes available on
// every lane is read with ONE PlayableBatch::Run and lands at output + l * out_lane_stride (out_bytes[l] bytes of it).
int ohhost_live_tick(ohhost_live* live, const uint8_t* input, uint64_t in_lane_stride, uint32_t frames, uint8_t* output,
                     uint64_t out_lane_stride, uint32_t* out_bytes)
{
    OHHOST_TRY({
        const TUint frameBytes = live->channels * (live->bits / 8);
        ASSERT(frames * frameBytes <= DecodedAudio::kMaxBytes);
        for (size_t l = 0; l < live->lanes.size(); l++) {
            LiveLane& lane = *live->lanes[l];
            lane.iPending.push_back(live->factory->CreateMsgAudioPcm(Brn(input + l * in_lane_stride, frames * frameBytes), live->channels,
                                                                     live->rateIn, live->bits, live->endian, 0));
            lane.iSink.iDst = output + l * out_lane_stride;
            lane.iSink.iBytes = 0;
            lane.iSink.iCap = (uint32_t)out_lane_stride;
            while (!lane.iPending.empty()) {
                Msg* msg = lane.iSrc.Pull();
                if (KindOf(msg) == MsgKind::AudioPcm) live->batch->Add(static_cast<MsgAudioPcm*>(msg)->CreatePlayable(), lane.iSink);
                else msg->RemoveRef();
            }
        }
        live->batch->Run();
        for (size_t l = 0; l < live->lanes.size(); l++) out_bytes[l] = live->lanes[l]->iSink.iBytes;
        return 0;
    });
}

int ohhost_live_stats(ohhost_live* live, uint64_t* src_calls, uint64_t* h2d_bytes, uint64_t* d2h_bytes, uint64_t* device_allocs, uint32_t* filters)
{
    OHHOST_TRY({
        uint64_t calls = 0;
        if (ohgpu_host_transfer_stats(live->factory->Gpu(), &calls, src_calls, h2d_bytes, d2h_bytes) != OHGPU_OK) return -4;
        if (ohgpu_device_allocations(live->factory->Gpu(), device_allocs) != OHGPU_OK) return -4;
        *filters = live->factory->FilterCount();
        return 0;
    });
}

int ohhost_live_destroy(ohhost_live* live)
{
    OHHOST_TRY({
        if (live != nullptr) {
            for (auto& lane : live->lanes) for (Msg* m : lane->iPending) m->RemoveRef();
            live->lanes.clear();
            live->batch.reset();
            delete live;
        }
        return 0;
    });
}

}  // extern "C"
