// test_host.cpp -- tests of the C++ host adapter (ohpipeline_amd/host), written the way the reference's own suites are
// (OpenHome/Media/Tests/TestMsg.cpp SuiteMsgAudio / SuiteMsgPlayable / SuiteRamp, Tests/TestRamper.cpp): hand-built
// messages, elements pulled through a suite that acts as the upstream element, results inspected through an
// IPcmProcessor.  `test_host cpu` runs the control-plane checks (no GPU); `test_host gpu` also reads audio through the
// C ABI and compares the bytes with the CPU oracle.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ohgpu.h"
#include "../../ohpipeline_amd/host/DecodedAudioAggregator.h"
#include "../../ohpipeline_amd/host/FlywheelRamper.h"
#include "../../ohpipeline_amd/host/Msg.h"
#include "../../ohpipeline_amd/host/SampleRateConverter.h"
#include "../../ohpipeline_amd/host/Sender.h"
#include "../../ohpipeline_amd/host/StarvationRamper.h"
#include "../../oracle/ohp_flywheel.h"
#include "../../oracle/ohp_oracle.h"
#include "../../oracle/ohp_pipeline.h"
#include "../../oracle/ohp_songcast.h"

using namespace OpenHome;
using namespace OpenHome::Media;

static int gFailures = 0, gChecks = 0;

// ---- fixtures of these suites.  They stand where two of the reference's own elements stand in a pipeline -- Ramper
// (Pipeline/Ramper.cpp: a stream that starts live or mid-track fades in) and PreDriver (Pipeline/PreDriver.cpp: audio becomes
// playables for the driver) -- so that the chains under test have the neighbours they have in a product.  They are test code:
// a drop-in keeps the reference's elements, the product ships neither.
class FadeInAtStreamStart : public PipelineElement, public IPipelineElementUpstream {
public:
    FadeInAtStreamStart(IPipelineElementUpstream& aUpstream, TUint aLongJiffies, TUint aShortJiffies)
        : PipelineElement(0xffffffffu), iUpstream(aUpstream), iLong(aLongJiffies), iShort(aShortJiffies), iFade(aLongJiffies) {}
    Msg* Pull() override
    {
        Msg* next = nullptr;
        if (iHeld.empty()) next = iUpstream.Pull();
        else { next = iHeld.front(); iHeld.pop_front(); }
        return next->Process(*this);
    }
    TBool Holding() const { return !iHeld.empty(); }
private:
    Msg* ProcessMsg(MsgMode* aMsg) override { iFade = aMsg->Info().RampPauseResumeLong() ? iLong : iShort; return aMsg; }
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override
    {
        const DecodedStreamInfo& info = aMsg->StreamInfo();
        iLeft = (info.Live() || info.SampleStart() > 0) ? iFade : 0;
        iLevel = Ramp::kMin;
        return aMsg;
    }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override
    {
        if (iLeft == 0) return aMsg;
        if (aMsg->Jiffies() > iLeft) iHeld.push_front(aMsg->Split(iLeft));
        MsgAudio* rest = nullptr;
        iLevel = aMsg->SetRamp(iLevel, iLeft, Ramp::EUp, rest);
        if (rest != nullptr) iHeld.push_front(rest);
        return aMsg;
    }
private:
    IPipelineElementUpstream& iUpstream;
    const TUint iLong, iShort;
    TUint iFade, iLeft = 0, iLevel = Ramp::kMin;
    std::deque<Msg*> iHeld;
};

class DriverEdge : public PipelineElement, public IPipelineElementUpstream {
public:
    explicit DriverEdge(IPipelineElementUpstream& aUpstream) : PipelineElement(0xffffffffu), iUpstream(aUpstream) {}
    Msg* Pull() override { return iUpstream.Pull()->Process(*this); }
private:
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override { return aMsg->CreatePlayable(); }
    Msg* ProcessMsg(MsgSilence* aMsg) override { return aMsg->CreatePlayable(); }
private:
    IPipelineElementUpstream& iUpstream;
};

class Semaphore {                                        // the counting semaphore the reference's suites lean on
public:
    explicit Semaphore(TUint aCount = 0) : iCount(aCount) {}
    void Wait() { std::unique_lock<std::mutex> lock(iLock); iCv.wait(lock, [this] { return iCount > 0; }); iCount--; }
    void Signal() { { std::lock_guard<std::mutex> lock(iLock); iCount++; } iCv.notify_one(); }
    TBool Clear() { std::lock_guard<std::mutex> lock(iLock); const TBool pending = iCount > 0; iCount = 0; return pending; }
private:
    std::mutex iLock;
    std::condition_variable iCv;
    TUint iCount;
};
#define TEST(x) do { gChecks++; if (!(x)) { gFailures++; printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #x); } } while (0)
#define TEST_THROWS(expr, Ex) do { gChecks++; bool thrown_ = false; try { expr; } catch (Ex&) { thrown_ = true; } \
    if (!thrown_) { gFailures++; printf("FAILED %s:%d  %s did not throw %s\n", __FILE__, __LINE__, #expr, #Ex); } } while (0)

// RampValidator's continuity rule (Pipeline/RampValidator.cpp:73-124) as a checker the suites below hang behind the manager's
// lanes: a ramp starts at an end of the scale, every message of a ramp starts where the last one ended (a drain may jump to an
// end), a ramp that reached its end is over; Mode and DecodedStream start afresh.  The reference logs a warning; here it fails.
class RampContinuity {
public:
    void NewStream() { iRamping = false; iLast = 0xffffffffu; iDraining = false; }
    void Drain() { iDraining = true; }
    void Audio(const Media::Ramp& aRamp)
    {
        if (iRamping) {
            if (aRamp.Start() != iLast) {
                TEST(iDraining && (aRamp.Start() == Ramp::kMin || aRamp.Start() == Ramp::kMax));
            }
            iLast = aRamp.End();
            Complete(aRamp);
        }
        else if (aRamp.IsEnabled()) {
            iRamping = true;
            if (aRamp.Direction() == Ramp::EUp) TEST(aRamp.Start() == Ramp::kMin);
            else if (aRamp.Direction() == Ramp::EDown) TEST(aRamp.Start() == Ramp::kMax);
            iLast = aRamp.End();
            Complete(aRamp);
        }
        iDraining = false;
        iChecked++;
    }
    TUint Checked() const { return iChecked; }
private:
    void Complete(const Media::Ramp& aRamp)
    {
        if ((aRamp.Direction() == Ramp::EUp && iLast == Ramp::kMax) || (aRamp.Direction() == Ramp::EDown && iLast == Ramp::kMin)) {
            iRamping = false;
            iLast = 0xffffffffu;
        }
    }
    TBool iRamping = false, iDraining = false;
    TUint iLast = 0xffffffffu, iChecked = 0;
};

// ------------------------------------------------------------------------------------------- control plane
static void SuiteRampControl()
{   // TestMsg.cpp:1391-1443, 1593-1625
    TUint jiffies = Jiffies::kPerMs;
    Ramp ramp, split;
    TUint splitPos;
    TEST(!ramp.Set(Ramp::kMax, jiffies, jiffies, Ramp::EDown, split, splitPos));
    TEST(ramp.Start() == Ramp::kMax && ramp.End() == Ramp::kMin && ramp.Direction() == Ramp::EDown);
    ramp.Reset();
    TEST_THROWS(ramp.Set(Ramp::kMax, jiffies, jiffies, Ramp::EUp, split, splitPos), AssertionFailed);
    ramp.Reset();
    TEST(!ramp.Set(Ramp::kMin, jiffies, jiffies, Ramp::EUp, split, splitPos));
    TEST(ramp.Start() == Ramp::kMin && ramp.End() == Ramp::kMax && ramp.Direction() == Ramp::EUp);
    ramp.Reset();
    TEST(!ramp.Set(Ramp::kMax, jiffies, 2 * jiffies, Ramp::EDown, split, splitPos));
    TEST(ramp.End() == (Ramp::kMax - Ramp::kMin) / 2);
    ramp.Reset();
    TUint start = (Ramp::kMax - Ramp::kMin) / 2;
    TEST(!ramp.Set(start, jiffies, 2 * jiffies, Ramp::EUp, split, splitPos));
    TEST(ramp.End() == Ramp::kMax - ((Ramp::kMax - Ramp::kMin) / 4));
    // [50%..Min] then [Min..50%]: split into [Min..25%], [25%..Min]
    ramp.Reset();
    TEST(!ramp.Set(Ramp::kMax / 2, jiffies, jiffies, Ramp::EDown, split, splitPos));
    TEST(ramp.Set(Ramp::kMin, jiffies, 2 * jiffies, Ramp::EUp, split, splitPos));
    TEST(ramp.Start() == 0 && ramp.End() == Ramp::kMax / 4 && ramp.Direction() == Ramp::EUp);
    TEST(split.Start() == ramp.End() && split.End() == 0 && split.Direction() == Ramp::EDown);
    // same direction: lower points win
    ramp.Reset();
    TEST(!ramp.Set(Ramp::kMax / 2, jiffies, 2 * jiffies, Ramp::EDown, split, splitPos));
    start = (TUint)(((TUint64)2 * Ramp::kMax) / 5);
    TEST(!ramp.Set(start, jiffies, jiffies, Ramp::EDown, split, splitPos));
    TEST(ramp.Start() == start && ramp.End() == 0);
    ramp.Reset();
    ramp.SetMuted();
    TEST(ramp.Direction() == Ramp::EMute && ramp.Start() == Ramp::kMin && ramp.End() == Ramp::kMin);
    TEST_THROWS(Jiffies::PerSample(44101), SampleRateInvalid);
    TEST(Jiffies::PerSample(44100) == 1280 && Jiffies::PerSample(48000) == 1176);
}

static void SuiteMsgAudioControl(MsgFactory& f)
{   // TestMsg.cpp:780-836, 926-945
    std::vector<TByte> data(1200, 0xde);
    Brn buf(data.data(), (TUint)data.size());
    const TUint rates[] = { 7350, 8000, 11025, 12000, 14700, 16000, 22050, 24000, 29400, 32000, 44100, 48000, 88200, 96000, 176400, 192000 };
    TUint prev = 0xffffffff;
    for (TUint r : rates) {
        MsgAudio* msg = f.CreateMsgAudioPcm(buf, 2, r, 8, AudioDataEndian::Little, 0);
        TEST(prev > msg->Jiffies());
        prev = msg->Jiffies();
        msg->RemoveRef();
    }
    MsgAudioPcm* msg = f.CreateMsgAudioPcm(buf, 2, 44100, 8, AudioDataEndian::Little, Jiffies::kPerSecond);
    const TUint jiffies = msg->Jiffies();
    MsgAudio* remaining = msg->Split(800);
    TEST(msg->Jiffies() + remaining->Jiffies() == jiffies);
    TEST(static_cast<MsgAudioPcm*>(remaining)->TrackOffset() == msg->TrackOffset() + msg->Jiffies());
    remaining->RemoveRef();
    TEST_THROWS(remaining = msg->Split(0), AssertionFailed);
    TEST_THROWS(remaining = msg->Split(msg->Jiffies()), AssertionFailed);
    TEST_THROWS(remaining = msg->Split(msg->Jiffies() + 1), AssertionFailed);
    MsgAudio* clone = msg->Clone();
    TEST(clone->Jiffies() == msg->Jiffies());
    msg->RemoveRef();
    clone->RemoveRef();
    TEST_THROWS(f.CreateMsgAudioPcm(Brn(data.data(), 0), 2, 44100, 8, AudioDataEndian::Little, 0), AssertionFailed);
    // aggregate: lengths add, mismatches assert
    std::vector<TByte> d1(4596, 1), d2(4596, 2);
    MsgAudioPcm* a1 = f.CreateMsgAudioPcm(Brn(d1.data(), 4596), 2, 44100, 8, AudioDataEndian::Little, 0);
    MsgAudioPcm* a2 = f.CreateMsgAudioPcm(Brn(d2.data(), 4596), 2, 44100, 8, AudioDataEndian::Little, a1->Jiffies());
    const TUint expected = a1->Jiffies() + a2->Jiffies();
    a1->Aggregate(a2);
    TEST(a1->Jiffies() == expected);
    MsgAudioPcm* a3 = f.CreateMsgAudioPcm(Brn(d2.data(), 4596), 1, 44100, 8, AudioDataEndian::Little, a1->Jiffies());
    TEST_THROWS(a1->Aggregate(a3), AssertionFailed);
    a3->RemoveRef();
    a1->RemoveRef();
    // silence
    TUint sj = Jiffies::kPerMs;
    MsgSilence* silence = f.CreateMsgSilence(sj, 44100, 8, 2);
    TEST(sj == silence->Jiffies());
    MsgAudio* rest = silence->Split(sj / 4);
    TEST(silence->Jiffies() + rest->Jiffies() == sj);
    MsgPlayable* p = silence->CreatePlayable();
    MsgPlayable* rp = static_cast<MsgSilence*>(rest)->CreatePlayable();
    TEST(p->Bytes() + rp->Bytes() == (sj / Jiffies::PerSample(44100)) * 2);
    p->RemoveRef();
    rp->RemoveRef();
}

static void SuitePlayableControl(MsgFactory& f)
{   // TestMsg.cpp:1106-1251 (sizes and split points; contents are checked in the gpu pass)
    TByte data[256];
    for (TUint i = 0; i < 256; i++) data[i] = (TByte)(0xff - i);
    Brn buf(data, 256);
    MsgAudioPcm* pcm = f.CreateMsgAudioPcm(buf, 2, 44100, 8, AudioDataEndian::Little, 0);
    MsgAudioPcm* rem = static_cast<MsgAudioPcm*>(pcm->Split(pcm->Jiffies() / 4));
    MsgPlayable* p = pcm->CreatePlayable();
    MsgPlayable* rp = rem->CreatePlayable();
    TEST(rp->Bytes() == 3 * p->Bytes());
    MsgPlayable* tail = rp->Split(rp->Bytes() / 2);
    TEST(tail != nullptr && tail->Bytes() + rp->Bytes() == 3 * p->Bytes());
    TEST(rp->Split(rp->Bytes()) == nullptr);
    TEST_THROWS(rp->Split(0), AssertionFailed);
    TEST_THROWS(rp->Split(rp->Bytes() + 1), AssertionFailed);
    p->RemoveRef(); rp->RemoveRef(); tail->RemoveRef();
    pcm = f.CreateMsgAudioPcm(buf, 2, 44100, 8, AudioDataEndian::Little, 0);      // split at 1 jiffy
    rem = static_cast<MsgAudioPcm*>(pcm->Split(1));
    p = pcm->CreatePlayable();
    rp = rem->CreatePlayable();
    TEST(p->Bytes() == 0 && rp->Bytes() == 256);
    p->RemoveRef(); rp->RemoveRef();
    pcm = f.CreateMsgAudioPcm(buf, 2, 44100, 8, AudioDataEndian::Little, 0);      // muted -> silence playable
    pcm->SetMuted();
    p = pcm->CreatePlayable();
    TEST(p->Work().silence && p->Bytes() == 256);
    p->RemoveRef();
}

// ------------------------------------------------------------------------------------------- data plane (GPU)
static std::vector<TByte> OracleRead(const PlayableWork& w, TUint outBits, int outEndian)
{
    ohp_msg_desc d;
    memset(&d, 0, sizeof(d));
    d.src_offset = w.offsetBytes;
    d.n_frames = w.frames;
    d.ramp_start = (uint16_t)w.ramp.Start();
    d.ramp_end = (uint16_t)w.ramp.End();
    d.attenuation = (uint16_t)w.attenuation;
    d.channels = (uint8_t)w.channels;
    d.src_bits = (uint8_t)w.bitDepth;
    d.src_endian = (w.audio && w.audio->Endian() == AudioDataEndian::Little) ? OHP_ENDIAN_LITTLE : OHP_ENDIAN_BIG;
    d.dst_bits = (uint8_t)outBits;
    d.dst_endian = (uint8_t)outEndian;
    d.flags = (uint8_t)((w.silence ? OHP_FLAG_SILENCE : 0) | (w.ramp.IsEnabled() && !w.silence ? OHP_FLAG_RAMP : 0));
    std::vector<TByte> out((size_t)w.frames * w.channels * (outBits / 8));
    const TByte* src = w.audio ? w.audio->Ptr(0) : nullptr;
    TEST(ohp_msg_process(&d, src, out.data()) == 0);
    return out;
}

class SuiteRamperGpu : public IPipelineElementUpstream, private IMsgProcessor {
    // Tests/TestRamper.cpp:18-105, 252-360 restated: the suite is the upstream element and the inspecting processor
    static const TUint kRampLong = Jiffies::kPerMs * 50, kRampShort = Jiffies::kPerMs * 20;
public:
    explicit SuiteRamperGpu(MsgFactory& aFactory) : iFactory(aFactory), iRamper(*this, kRampLong, kRampShort) {}
    void Run()
    {
        // not live, starts at 0: no ramp, audio comes through untouched
        Push(iFactory.CreateMsgMode(ModeInfo()));
        Push(Stream(false, 0));
        Push(Audio());
        iRamping = false;
        for (int i = 0; i < 3; i++) PullNext();
        TEST(iJiffies > 0);
        // live: ramps up over kRampLong, then stays at full level
        iJiffies = 0;
        Push(Stream(true, 0));
        PullNext();
        iRamping = true;
        iLastSubsample = 0;
        while (iRamping) {
            Push(Audio());
            PullNextAll();
        }
        TEST(iJiffies >= kRampLong);
        TEST(iRampedJiffies == kRampLong);
        Push(Audio());
        PullNextAll();
        // mid-track start of a new stream: ramp; short mode: short ramp
        ModeInfo shortMode;
        shortMode.iRampPauseResumeLong = false;
        Push(iFactory.CreateMsgMode(shortMode));
        PullNext();
        iJiffies = 0; iRampedJiffies = 0;
        Push(Stream(false, 100));
        PullNext();
        iRamping = true;
        iLastSubsample = 0;
        while (iRamping) {
            Push(Audio());
            PullNextAll();
        }
        TEST(iRampedJiffies == kRampShort);
    }
private:
    void Push(Msg* aMsg) { iPending.push_back(aMsg); }
    Msg* Pull() override { ASSERT(!iPending.empty()); Msg* m = iPending.front(); iPending.pop_front(); return m; }
    void PullNext() { Msg* m = iRamper.Pull(); m = m->Process(*this); m->RemoveRef(); }
    void PullNextAll() { do { PullNext(); } while (!iPending.empty() || iRamper.Holding()); }
    Msg* Stream(TBool aLive, TUint64 aSampleStart)
    {
        DecodedStreamInfo info;
        info.iStreamId = iNextStreamId++; info.iBitDepth = 24; info.iSampleRate = 44100; info.iNumChannels = 2;
        info.iLive = aLive; info.iSampleStart = aSampleStart;
        return iFactory.CreateMsgDecodedStream(info);
    }
    Msg* Audio()
    {
        TByte data[3 * 1024];
        memset(data, 0x7f, sizeof data);
        MsgAudioPcm* audio = iFactory.CreateMsgAudioPcm(Brn(data, sizeof data), 2, 44100, 24, AudioDataEndian::Little, iTrackOffset);
        iTrackOffset += audio->Jiffies();
        return audio;
    }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override
    {
        iJiffies += aMsg->Jiffies();
        if (aMsg->Ramp().IsEnabled()) iRampedJiffies += aMsg->Jiffies();
        iMoreQueued = false;
        MsgPlayable* playable = aMsg->CreatePlayable();
        const std::vector<TByte> expected = OracleRead(playable->Work(), 24, OHP_ENDIAN_BIG);
        ProcessorPcmBufTest pcm;
        playable->Read(pcm);
        const Brn buf(pcm.Buf());
        TEST(buf.Bytes() == expected.size() && memcmp(buf.Ptr(), expected.data(), expected.size()) == 0);   // bit-exact vs oracle
        const TByte* ptr = buf.Ptr();
        const TUint bytes = buf.Bytes();
        const TUint first = (ptr[0] << 16) | (ptr[1] << 8) | ptr[2];
        if (iRamping) {
            TEST(first >= iLastSubsample || iLastSubsample == 0);
            iLastSubsample = (ptr[bytes - 3] << 16) | (ptr[bytes - 2] << 8) | ptr[bytes - 1];
            TEST(iLastSubsample >= first);
            if (!playable->Ramp().IsEnabled()) iRamping = false;     // ramp finished: this message is at full level
            else iMoreQueued = playable->Ramp().End() != Ramp::kMax ? false : true;
            if (playable->Ramp().IsEnabled()) {
                const TUint perFrag = 256 / 6;
                TEST(pcm.Fragments().size() == (bytes / 6 + perFrag - 1) / perFrag);   // 42-frame fragments (Msg.cpp:2766)
            }
        }
        else {
            TEST(first == 0x7f7f7f);
            TEST(pcm.Fragments().size() == 1);
        }
        return playable;
    }
    Msg* ProcessMsg(MsgMode* m) override { return m; }
    Msg* ProcessMsg(MsgTrack* m) override { return m; }
    Msg* ProcessMsg(MsgDrain* m) override { return m; }
    Msg* ProcessMsg(MsgDelay* m) override { return m; }
    Msg* ProcessMsg(MsgEncodedStream* m) override { return m; }
    Msg* ProcessMsg(MsgStreamSegment* m) override { return m; }
    Msg* ProcessMsg(MsgAudioEncoded* m) override { return m; }
    Msg* ProcessMsg(MsgMetaText* m) override { return m; }
    Msg* ProcessMsg(MsgStreamInterrupted* m) override { return m; }
    Msg* ProcessMsg(MsgHalt* m) override { return m; }
    Msg* ProcessMsg(MsgFlush* m) override { return m; }
    Msg* ProcessMsg(MsgWait* m) override { return m; }
    Msg* ProcessMsg(MsgDecodedStream* m) override { return m; }
    Msg* ProcessMsg(MsgAudioDsd* m) override { return m; }
    Msg* ProcessMsg(MsgSilence* m) override { return m; }
    Msg* ProcessMsg(MsgPlayable* m) override { return m; }
    Msg* ProcessMsg(MsgQuit* m) override { return m; }
private:
    MsgFactory& iFactory;
    FadeInAtStreamStart iRamper;
    std::deque<Msg*> iPending;
    TUint iNextStreamId = 1;
    TUint64 iTrackOffset = 0;
    TUint iJiffies = 0, iRampedJiffies = 0, iLastSubsample = 0;
    TBool iRamping = false, iMoreQueued = false;
};

static void SuitePlayableGpu(MsgFactory& f)
{   // TestMsg.cpp:1133-1236: byte-exact pass-through across split points; attenuation KAT :982-996; batch read
    TByte data[256];
    for (TUint i = 0; i < 256; i++) data[i] = (TByte)(0xff - i);
    MsgAudioPcm* pcm = f.CreateMsgAudioPcm(Brn(data, 256), 2, 44100, 8, AudioDataEndian::Little, 0);
    MsgAudioPcm* rem = static_cast<MsgAudioPcm*>(pcm->Split(pcm->Jiffies() / 4 - 1));       // non-sample boundary
    MsgPlayable* p = pcm->CreatePlayable();
    MsgPlayable* rp = rem->CreatePlayable();
    ProcessorPcmBufTest a, b;
    PlayableBatch batch(f);
    batch.Add(p, a);
    batch.Add(rp, b);
    batch.Run();
    TUint v = 0xff;
    for (TUint i = 0; i < a.Buf().Bytes(); i++, v--) TEST(a.Buf()[i] == v);
    for (TUint i = 0; i < b.Buf().Bytes(); i++, v--) TEST(b.Buf()[i] == v);
    TEST(a.Buf().Bytes() + b.Buf().Bytes() == 256);
    const TByte s = 0x7f;
    TByte sample[] = { s, s, s, s };
    MsgAudioPcm* att = f.CreateMsgAudioPcm(Brn(sample, 4), 2, 44100, 16, AudioDataEndian::Little, 0);
    att->SetAttenuation(MsgAudioPcm::kUnityAttenuation / 4);
    MsgPlayable* ap = att->CreatePlayable();
    ProcessorPcmBufTest c;
    ap->Read(c);
    ap->RemoveRef();
    const TInt16 sub = (TInt16)((c.Ptr()[0] << 8) + c.Ptr()[1]);
    TEST(sub == ((s << 8) + s) / 4);
    // silence, 6 channels, 32 bit: zeros plus the channel-id bytes (Msg.cpp:2877)
    TUint sj = Jiffies::kPerMs;
    MsgSilence* sil = f.CreateMsgSilence(sj, 192000, 32, 6);
    MsgPlayable* sp = sil->CreatePlayable();
    const std::vector<TByte> expected = OracleRead(sp->Work(), 32, OHP_ENDIAN_BIG);
    ProcessorPcmBufTest d;
    sp->Read(d);
    sp->RemoveRef();
    TEST(d.Buf().Bytes() == expected.size() && memcmp(d.Ptr(), expected.data(), expected.size()) == 0);
    TEST(d.Ptr()[7] == 0x10 && d.Ptr()[31] == 0x70 && d.Ptr()[35] == 0x00);
}

class SuiteSrcGpu : public IPipelineElementUpstream {
    // SampleRateConverter -> Ramper -> PreDriver, 44.1 kHz S24LE stereo in, 48 kHz out: the chain of BASELINE config 2.
public:
    explicit SuiteSrcGpu(MsgFactory& aFactory)
        : iFactory(aFactory), iSrc(aFactory, *this, 48000), iRamper(iSrc, Jiffies::kPerMs * 50, Jiffies::kPerMs * 20), iPreDriver(iRamper) {}
    void Run()
    {
        DecodedStreamInfo info;
        info.iStreamId = 7; info.iBitDepth = 24; info.iSampleRate = 44100; info.iNumChannels = 2; info.iLive = true;
        iPending.push_back(iFactory.CreateMsgDecodedStream(info));
        uint32_t x = 12345;
        const TUint kMsgs = 40;                       // 200 ms of 220-frame messages
        for (TUint m = 0; m < kMsgs; m++) {
            TByte data[220 * 6];
            for (TUint i = 0; i < sizeof data; i += 3) {
                x = x * 1664525u + 1013904223u;
                data[i] = (TByte)(x >> 8); data[i + 1] = (TByte)(x >> 16); data[i + 2] = (TByte)(x >> 24);
            }
            iInput.insert(iInput.end(), data, data + sizeof data);
            iPending.push_back(iFactory.CreateMsgAudioPcm(Brn(data, sizeof data), 2, 44100, 24, AudioDataEndian::Little, 0));
        }
        iPending.push_back(iFactory.CreateMsgQuit());
        // drive like an animator: pull, batch the playables of this "period", read them in one launch
        std::vector<TByte> got;
        std::vector<ProcessorPcmBufTest> sinks(64);
        TUint64 outFrames = 0;
        TBool quit = false, sawStream = false;
        std::vector<ohp_src_msg_desc> descs;
        while (!quit) {
            PlayableBatch batch(iFactory);
            size_t n = 0;
            while (n < 8 && !quit) {
                Msg* msg = iPreDriver.Pull();
                if (MsgPlayable* p = dynamic_cast<MsgPlayable*>(msg)) {
                    ohp_src_msg_desc d;
                    memset(&d, 0, sizeof(d));
                    d.src_frames = iInput.size() / 6; d.out_frame0 = outFrames; d.dst_offset = outFrames * 6;
                    d.n_frames = p->Bytes() / 6; d.ramp_start = (uint16_t)p->Ramp().Start(); d.ramp_end = (uint16_t)p->Ramp().End();
                    d.attenuation = 256; d.channels = 2; d.src_bits = 24; d.src_endian = OHP_ENDIAN_LITTLE; d.dst_bits = 24;
                    d.dst_endian = OHP_ENDIAN_BIG; d.flags = p->Ramp().IsEnabled() ? OHP_FLAG_RAMP : 0;
                    descs.push_back(d);
                    outFrames += d.n_frames;
                    batch.Add(p, sinks[n++]);
                }
                else if (MsgDecodedStream* s = dynamic_cast<MsgDecodedStream*>(msg)) {
                    TEST(s->StreamInfo().SampleRate() == 48000 && s->StreamInfo().BitDepth() == 24);
                    sawStream = true;
                    msg->RemoveRef();
                }
                else {
                    quit = dynamic_cast<MsgQuit*>(msg) != nullptr;
                    msg->RemoveRef();
                }
            }
            batch.Run();
            for (size_t i = 0; i < n; i++) got.insert(got.end(), sinks[i].Ptr(), sinks[i].Ptr() + sinks[i].Buf().Bytes());
        }
        TEST(sawStream);
        TEST(outFrames == (kMsgs * 220ull * 160 + 146) / 147);
        // the same messages through the oracle's resample -> ramp -> fmt
        ohp_src* ref = ohp_src_new(44100, 48000, 32, 9.0, 20000.0);
        std::vector<TByte> want(got.size());
        TEST(ohp_src_msg_process_batch(ref, descs.data(), descs.size(), iInput.data(), want.data()) == 0);
        ohp_src_delete(ref);
        TEST(got.size() == outFrames * 6 && memcmp(got.data(), want.data(), got.size()) == 0);
        TUint ramped = 0;
        for (auto& d : descs) if (d.flags & OHP_FLAG_RAMP) ramped += d.n_frames;
        TEST(ramped == 50 * 48);                      // the live stream's 50 ms ramp, at the OUTPUT rate
    }
    Msg* Pull() override { ASSERT(!iPending.empty()); Msg* m = iPending.front(); iPending.pop_front(); return m; }
private:
    MsgFactory& iFactory;
    SampleRateConverter iSrc;
    FadeInAtStreamStart iRamper;
    DriverEdge iPreDriver;
    std::deque<Msg*> iPending;
    std::vector<TByte> iInput;
};

// A rate-converted stream's history is a ring, and a reader is handed the window of input its outputs are made of -- not the
// history (round 5).  CPU: the ring against a linear copy of everything appended, over several turns of the ring, in ragged pieces;
// the window against the specification's two lines; what is gone, or not there yet, asserts.
static void SuiteSrcStreamRing()
{
    SrcFilter flt;                                   // (no device behind it: the ring and the window arithmetic need none)
    flt.L = 160; flt.M = 147; flt.T = 32;
    SampleRateConverterStream st(flt, 44100, 2, 24, AudioDataEndian::Little, 100);      // 100 ms = 4410 frames of history
    std::vector<TByte> all;
    uint32_t x = 99;
    TUint64 outSeen = 0;
    for (TUint piece = 0; piece < 400; piece++) {
        x = x * 1664525u + 1013904223u;
        const TUint frames = 1 + (x >> 16) % 300;
        std::vector<TByte> data(frames * 6);
        for (auto& b : data) { x = x * 1664525u + 1013904223u; b = (TByte)(x >> 24); }
        all.insert(all.end(), data.begin(), data.end());
        const TUint64 avail = st.Append(data.data(), (TUint)data.size());
        TEST(avail == (all.size() / 6 * 160 + 146) / 147);
        TEST(st.InputFrames() == all.size() / 6);
        if (avail == outSeen) continue;
        // the window of the outputs that have just become available, against first principles
        TUint64 first = 0; TUint n = 0;
        st.Window(outSeen, (TUint)(avail - outSeen), first, n);
        const TUint64 n0First = outSeen * 147 / 160, n0Last = (avail - 1) * 147 / 160;
        TEST(first == (n0First >= 31 ? n0First - 31 : 0) && first + n == n0Last + 1);
        TEST(first + n <= all.size() / 6);                        // an output exists only once its newest input has arrived
        std::vector<TByte> got((size_t)n * 6);
        st.CopyFrames(first, n, got.data());
        TEST(memcmp(got.data(), &all[(size_t)first * 6], got.size()) == 0);
        outSeen = avail;
    }
    TEST(all.size() / 6 > 10 * 4410);                             // the ring went round many times
    const TUint64 frames = all.size() / 6;
    std::vector<TByte> buf(4410 * 6);
    st.CopyFrames(frames - 4410, 4410, buf.data());               // the whole ring, across its seam
    TEST(memcmp(buf.data(), &all[(size_t)(frames - 4410) * 6], buf.size()) == 0);
    TEST_THROWS(st.CopyFrames(frames - 4411, 10, buf.data()), AssertionFailed);     // overwritten
    TEST_THROWS(st.CopyFrames(frames - 5, 10, buf.data()), AssertionFailed);        // not appended yet
    TUint64 f0 = 0; TUint n0 = 0;
    st.Window(0, 1, f0, n0);
    TEST(f0 == 0 && n0 == 1);                                     // the stream's first output: one input frame, zeros before it
    st.Window(35, 240, f0, n0);
    TEST(f0 == 35 * 147 / 160 - 31 && f0 + n0 == (35 + 239) * 147 / 160 + 1);
}

// 256 live rate-converted streams behind ONE driver thread (round 5): every lane is SampleRateConverter -> the fade-in fixture ->
// the driver edge, a tick feeds every lane 5 ms of input and reads what comes out of all of them with ONE PlayableBatch::Run.
// Three layouts over two conversions (44.1 -> 48 kHz: S24LE and S16BE stereo; 96 -> 48 kHz: S24LE stereo), so the factory holds two
// filters whatever the number of streams, and a tick is two resampler calls through the C ABI, not 256; what crosses the link
// per tick is the messages' windows, not the streams' histories; nothing is allocated on the device after the first ticks.
// Every lane's output is bit-exact against the oracle run over that lane's whole input.
class SuiteManyResampledLanesGpu {
    struct Lane : public IPipelineElementUpstream {
        Lane(MsgFactory& aFactory, TUint aRateIn, TUint aBits, AudioDataEndian aEndian)
            : rateIn(aRateIn), bits(aBits), endian(aEndian), src(aFactory, *this, 48000, aRateIn == 96000 ? 64 : 32)
            , fade(src, Jiffies::kPerMs * 50, Jiffies::kPerMs * 20), edge(fade) {}
        Msg* Pull() override { ASSERT(!pending.empty()); Msg* m = pending.front(); pending.pop_front(); return m; }
        TUint rateIn, bits;
        AudioDataEndian endian;
        SampleRateConverter src;
        FadeInAtStreamStart fade;
        DriverEdge edge;
        std::deque<Msg*> pending;
        std::vector<TByte> input, got;
        std::vector<ohp_src_msg_desc> descs;
        TUint64 outFrames = 0;
        uint32_t x = 0;
        TUint filled = 0;
        ProcessorPcmBufTest sinks[4];
    };
public:
    explicit SuiteManyResampledLanesGpu(MsgFactory& aFactory) : iFactory(aFactory) {}
    void Run()
    {
        const TUint kLanes = 256, kTicks = 200;
        const TUint filtersBefore = iFactory.FilterCount();
        std::vector<std::unique_ptr<Lane>> lanes;
        for (TUint l = 0; l < kLanes; l++) {
            const TUint kind = l % 8;                        // 6 of 8: 44.1 kHz S24LE; 1: 44.1 kHz S16BE; 1: 96 kHz S24LE
            lanes.emplace_back(new Lane(iFactory, kind == 7 ? 96000 : 44100, kind == 6 ? 16 : 24,
                                        kind == 6 ? AudioDataEndian::Big : AudioDataEndian::Little));
            Lane& lane = *lanes.back();
            lane.x = 0x9E3779B9u * (l + 1);
            DecodedStreamInfo info;
            info.iStreamId = l + 1; info.iBitDepth = lane.bits; info.iSampleRate = lane.rateIn; info.iNumChannels = 2; info.iLive = true;
            lane.pending.push_back(iFactory.CreateMsgDecodedStream(info));
        }
        PlayableBatch batch(iFactory);                        // one object, reused tick after tick
        uint64_t allocsAfterWarmup = 0, allocsSeen = 0, windowBytes = 0, h2dTotal = 0;
        TBool allocsFlat = true;
        TUint srcCallsPerTickMax = 0, srcCallsPerTickMin = 1000, sawStreams = 0, maxPlayablesPerTick = 0;
        TBool h2dWithinBudget = true;
        for (TUint tick = 0; tick < kTicks; tick++) {
            uint64_t tickWindowBytes = 0;
            TUint playables = 0;
            for (auto& lp : lanes) {
                Lane& lane = *lp;
                const TUint frames = lane.rateIn / 200;       // 5 ms
                const TUint sb = lane.bits / 8;
                std::vector<TByte> data(frames * 2 * sb);
                for (size_t i = 0; i < data.size(); i += sb) {
                    lane.x = lane.x * 1664525u + 1013904223u;
                    for (TUint b = 0; b < sb; b++) {          // the sample's bytes, least significant first or last
                        const TByte v = (TByte)(lane.x >> (32 - 8 * sb + 8 * b));
                        data[i + (lane.endian == AudioDataEndian::Little ? b : sb - 1 - b)] = v;
                    }
                }
                lane.input.insert(lane.input.end(), data.begin(), data.end());
                lane.pending.push_back(iFactory.CreateMsgAudioPcm(Brn(data.data(), (TUint)data.size()), 2, lane.rateIn, lane.bits, lane.endian, 0));
                TUint n = 0;
                while (!lane.pending.empty() || lane.fade.Holding()) {
                    Msg* msg = lane.edge.Pull();
                    if (MsgPlayable* p = dynamic_cast<MsgPlayable*>(msg)) {
                        ohp_src_msg_desc d;
                        memset(&d, 0, sizeof(d));
                        d.src_frames = lane.input.size() / (2 * sb); d.out_frame0 = lane.outFrames; d.dst_offset = lane.outFrames * 6;
                        d.n_frames = p->Bytes() / 6; d.ramp_start = (uint16_t)p->Ramp().Start(); d.ramp_end = (uint16_t)p->Ramp().End();
                        d.attenuation = 256; d.channels = 2; d.src_bits = (uint8_t)lane.bits;
                        d.src_endian = lane.endian == AudioDataEndian::Little ? OHP_ENDIAN_LITTLE : OHP_ENDIAN_BIG;
                        d.dst_bits = 24; d.dst_endian = OHP_ENDIAN_BIG; d.flags = p->Ramp().IsEnabled() ? OHP_FLAG_RAMP : 0;
                        lane.descs.push_back(d);
                        // what this message needs of its stream's input (the specification's window), for the link's budget
                        const TUint64 L = lane.rateIn == 96000 ? 1 : 160, M = lane.rateIn == 96000 ? 2 : 147, T = lane.rateIn == 96000 ? 64 : 32;
                        const TUint64 hi = (d.out_frame0 + d.n_frames - 1) * M / L, n0 = d.out_frame0 * M / L;
                        tickWindowBytes += (hi - (n0 >= T - 1 ? n0 - (T - 1) : 0) + 1) * 2 * sb;
                        lane.outFrames += d.n_frames;
                        ASSERT(n < 4);
                        batch.Add(p, lane.sinks[n++]);
                        playables++;
                    }
                    else {
                        if (MsgDecodedStream* s = dynamic_cast<MsgDecodedStream*>(msg)) {
                            TEST(s->StreamInfo().SampleRate() == 48000 && s->StreamInfo().BitDepth() == 24);
                            sawStreams++;
                        }
                        msg->RemoveRef();
                    }
                }
                lane.filled = n;                              // (how many sinks of this lane the tick filled)
            }
            uint64_t calls0 = 0, src0 = 0, h2d0 = 0, d2h0 = 0, calls1 = 0, src1 = 0, h2d1 = 0, d2h1 = 0;
            ohgpu_host_transfer_stats(iFactory.Gpu(), &calls0, &src0, &h2d0, &d2h0);
            batch.Run();                                      // THE tick: every lane's audio in one go
            ohgpu_host_transfer_stats(iFactory.Gpu(), &calls1, &src1, &h2d1, &d2h1);
            for (auto& lp : lanes) {
                Lane& lane = *lp;
                const TUint n = lane.filled;
                for (TUint k = 0; k < n; k++) lane.got.insert(lane.got.end(), lane.sinks[k].Ptr(), lane.sinks[k].Ptr() + lane.sinks[k].Buf().Bytes());
            }
            srcCallsPerTickMax = std::max(srcCallsPerTickMax, (TUint)(src1 - src0));
            srcCallsPerTickMin = std::min(srcCallsPerTickMin, (TUint)(src1 - src0));
            TEST(calls1 - calls0 == src1 - src0);             // nothing but resampled audio here: no other device call
            if ((h2d1 - h2d0) * 10 > tickWindowBytes * 12) h2dWithinBudget = false;
            TEST(d2h1 - d2h0 <= (uint64_t)playables * 241 * 6 + 256);      // the outputs and nothing else
            windowBytes += tickWindowBytes;
            h2dTotal += h2d1 - h2d0;
            maxPlayablesPerTick = std::max(maxPlayablesPerTick, playables);
            uint64_t allocs = 0;
            ohgpu_device_allocations(iFactory.Gpu(), &allocs);
            if (allocs != allocsSeen) { printf("SuiteManyResampledLanesGpu: tick %u: %llu device allocations so far\n", tick, (unsigned long long)allocs); allocsSeen = allocs; }
            if (tick == 11) allocsAfterWarmup = allocs;       // (the fade-in's ten messages split one of them: the arenas have seen their largest tick)
            if (tick > 11 && allocs != allocsAfterWarmup) allocsFlat = false;
        }
        TEST(allocsFlat);                                     // nothing is allocated on the device once the first ticks have sized the arenas
        TEST(sawStreams == kLanes);
        TEST(iFactory.FilterCount() == filtersBefore + 2 || iFactory.FilterCount() == 2);      // two conversions, 256 streams
        TEST(srcCallsPerTickMax == 2 && srcCallsPerTickMin == 2);                             // one C-ABI resampler call per filter per tick
        TEST(h2dWithinBudget);                                                                // windows only: <= 1.2 x their bytes, every tick
        TEST(h2dTotal * 10 <= windowBytes * 11);
        TEST(maxPlayablesPerTick >= kLanes);
        printf("SuiteManyResampledLanesGpu: %u lanes x %u ticks, %.1f KB of windows and %.1f KB over the link per tick, 2 resampler calls per tick\n",
               kLanes, kTicks, windowBytes / 1024.0 / kTicks, h2dTotal / 1024.0 / kTicks);
        // every lane against the oracle over the lane's whole input
        ohp_src* ref441 = ohp_src_new(44100, 48000, 32, 9.0, 20000.0);
        ohp_src* ref96 = ohp_src_new(96000, 48000, 64, 9.0, 20000.0);
        TUint lanesOk = 0, rampedOk = 0;
        for (auto& lp : lanes) {
            Lane& lane = *lp;
            for (auto& d : lane.descs) d.src_frames = lane.input.size() / (2 * (lane.bits / 8));
            std::vector<TByte> want(lane.got.size());
            const int rc = ohp_src_msg_process_batch(lane.rateIn == 96000 ? ref96 : ref441, lane.descs.data(), lane.descs.size(), lane.input.data(), want.data());
            const TUint64 expect = lane.rateIn == 96000 ? (TUint64)kTicks * 240 : ((TUint64)kTicks * 220 * 160 + 146) / 147;
            if (rc == 0 && lane.outFrames == expect && lane.got.size() == lane.outFrames * 6 && memcmp(lane.got.data(), want.data(), want.size()) == 0) lanesOk++;
            TUint ramped = 0;
            for (auto& d : lane.descs) if (d.flags & OHP_FLAG_RAMP) ramped += d.n_frames;
            if (ramped == 50 * 48) rampedOk++;                // the live stream's 50 ms fade-in, at the OUTPUT rate
        }
        ohp_src_delete(ref441);
        ohp_src_delete(ref96);
        TEST(lanesOk == kLanes);
        TEST(rampedOk == kLanes);
        for (auto& lp : lanes) { lp->pending.push_back(iFactory.CreateMsgQuit()); lp->edge.Pull()->RemoveRef(); }
    }
private:
    MsgFactory& iFactory;
};

// ------------------------------------------------------------------------------------------- FlywheelRamper (N1)
// Driven the way Tests/TestFlywheelRamper.cpp:619-671 drives the original: a manager with (generation, ramp) jiffies, a
// block of planar 32-bit audio, output collected by an IPcmProcessor -- here checked against the oracle, which the
// reference's known-answer tests pin.
class ProcessorBlocks : public IPcmProcessor {
public:
    void BeginBlock() override { iOpen = true; }
    void ProcessFragment(const Brx& aData, TUint aNumChannels, TUint aSubsampleBytes) override
    {
        TEST(iOpen && aSubsampleBytes == 4);
        iChannels = aNumChannels;
        iBlockBytes.push_back(aData.Bytes());
        iBuf.insert(iBuf.end(), aData.Ptr(), aData.Ptr() + aData.Bytes());
    }
    void ProcessSilence(const Brx&, TUint, TUint) override { TEST(false); }
    void EndBlock() override { iOpen = false; }
    void Flush() override {}
    std::vector<TByte> iBuf;
    std::vector<TUint> iBlockBytes;
    TUint iChannels = 0;
    TBool iOpen = false;
};

static void SuiteFlywheelGpu(MsgFactory& aFactory)
{
    const TUint kGenJiffies = Jiffies::kPerMs, kRampJiffies = 20 * Jiffies::kPerMs;   // StarvationRamper.cpp:374-375
    const struct { TUint rate, channels; } cases[] = { {44100, 2}, {192000, 8}, {96000, 6}, {48000, 1} };
    FlywheelRamperBatch batch(aFactory);
    std::vector<ProcessorBlocks> sinks(4);
    std::vector<std::vector<TByte>> inputs(4);
    uint32_t x = 99;
    for (int k = 0; k < 4; k++) {
        const TUint rate = cases[k].rate, ch = cases[k].channels;
        const TUint n = FlywheelRamper::SampleCount(rate, kGenJiffies) + (k == 1 ? 2 : 0);   // one request gets "slightly too much data"
        std::vector<TByte>& in = inputs[k];
        for (TUint c = 0; c < ch; c++) {
            int32_t v = (int32_t)(0x20000000u + c * 0x01000000u);
            for (TUint i = 0; i < n; i++) {
                x = x * 1664525u + 1013904223u;
                v += (int32_t)(x >> 8) - (1 << 23) - (int32_t)(i * 0x00040000u);          // a noisy, decaying run
                in.push_back((TByte)((uint32_t)v >> 24)); in.push_back((TByte)((uint32_t)v >> 16));
                in.push_back((TByte)((uint32_t)v >> 8)); in.push_back((TByte)v);
            }
        }
        if (k == 0) {                                                                    // the drop-in shape: one manager, one Ramp()
            FlywheelRamperManager manager(aFactory, sinks[0], kGenJiffies, kRampJiffies);
            manager.Ramp(Brn(in.data(), (TUint)in.size()), rate, ch);
        } else {
            batch.Add(sinks[k], Brn(in.data(), (TUint)in.size()), rate, ch, kGenJiffies, kRampJiffies);
        }
    }
    TEST(batch.Count() == 3);
    batch.Run();
    for (int k = 0; k < 4; k++) {
        const TUint rate = cases[k].rate, ch = cases[k].channels;
        const TUint inSamples = FlywheelRamper::SampleCount(rate, kGenJiffies), outFrames = FlywheelRamper::SampleCount(rate, kRampJiffies);
        const TUint block = FlywheelRamper::SampleCount(rate, FlywheelRamperManager::kMaxOutputJiffiesBlockSize);
        std::vector<TByte> want((size_t)outFrames * ch * 4);
        TEST(ohp_flywheel_ramp(inputs[k].data(), inputs[k].size() / ch, inSamples, rate, ch, outFrames, block, want.data()) == 0);
        TEST(sinks[k].iChannels == ch && sinks[k].iBuf.size() == want.size());
        TEST(sinks[k].iBuf.size() == want.size() && memcmp(sinks[k].iBuf.data(), want.data(), want.size()) == 0);
        TEST(sinks[k].iBlockBytes.size() == (outFrames + block - 1) / block);           // 1 ms blocks, FlywheelRamper.cpp:52-63
        TEST(sinks[k].iBlockBytes.front() == block * ch * 4);
    }
    TEST(FlywheelRamper::DecimationFactor(176400) == 4 && FlywheelRamper::DecimationFactor(96000) == 2 && FlywheelRamper::DecimationFactor(44100) == 1);
    ProcessorBlocks sink;
    FlywheelRamperManager manager(aFactory, sink, kGenJiffies, kRampJiffies);
    TByte few[8 * 4] = { 0 };
    TEST_THROWS(manager.Ramp(Brn(few, sizeof few), 44100, 2), AssertionFailed);          // less than the generation period: ASSERT, :180
    TEST_THROWS(manager.Ramp(Brn(few, sizeof few), 44100, 11), AssertionFailed);
}

// ------------------------------------------------------------------------------------------- starvation rescue (a7 + a11 + N1 + a12)
struct RescueCase { TUint rate, channels, bits, startRamp; };

// `count` frames of a slowly decaying tone plus noise, big-endian as the pipeline holds it; aFirst = index of the first frame
static std::vector<TByte> RescueSignal(const RescueCase& cs, uint32_t& x, TUint aFirst, TUint aCount)
{
    const TUint sb = cs.bits / 8, frameBytes = cs.channels * sb;
    std::vector<TByte> data((size_t)aCount * frameBytes);
    for (TUint f = 0; f < aCount; f++) {
        for (TUint c = 0; c < cs.channels; c++) {
            x = x * 1664525u + 1013904223u;
            const int32_t v = (int32_t)(0x30000000 - (int32_t)((aFirst + f) * 0x00080000u)) + (int32_t)(x >> 10) - (1 << 21);
            for (TUint b = 0; b < sb; b++) data[(size_t)f * frameBytes + c * sb + b] = (TByte)((uint32_t)v >> (24 - 8 * b));
        }
    }
    return data;
}

// The oracle's view of one rescue: a11 on the newest millisecond (aNewest: packed big-endian frames), the flywheel, the a12
// pack, then every message read with the ramp the host algebra gave it.  aMsgs: what the rescue queued (consumed here).
static void CheckRescued(MsgFactory& aFactory, const RescueCase& cs, const TByte* aNewest, std::deque<Msg*>& aMsgs)
{
    const TUint kTraining = Jiffies::kPerMs, kRampDown = 20 * Jiffies::kPerMs;         // StarvationRamper.cpp:374-375
    const TUint frameBytes = cs.channels * cs.bits / 8;
    const TUint inSamples = Jiffies::ToSamples(kTraining, cs.rate);
    std::vector<TByte> wantPlanar((size_t)inSamples * 4 * cs.channels);
    std::vector<uint32_t> pos(cs.channels, 0);
    TEST(ohp_flywheel_unpack(aNewest, inSamples * frameBytes, cs.channels, cs.bits / 8, wantPlanar.data(), inSamples * 4, pos.data()) == 0);
    const TUint outFrames = Jiffies::ToSamples(kRampDown, cs.rate), block = Jiffies::ToSamples(Jiffies::kPerMs, cs.rate);
    std::vector<TByte> ramp32((size_t)outFrames * cs.channels * 4), packed((size_t)outFrames * frameBytes);
    TEST(ohp_flywheel_ramp(wantPlanar.data(), inSamples * 4, inSamples, cs.rate, cs.channels, outFrames, block, ramp32.data()) == 0);
    uint32_t packedBytes = 0;
    TEST(ohp_rampgen_pack(ramp32.data(), (uint32_t)ramp32.size(), cs.bits, packed.data(), &packedBytes) == 0 && packedBytes == packed.size());
    PlayableBatch batch(aFactory);
    std::vector<ProcessorPcmBufTest> sinks((outFrames + block - 1) / block);
    std::vector<ohp_msg_desc> descs;
    TUint k = 0, frames = 0, lastEnd = cs.startRamp;
    TEST(aMsgs.size() == sinks.size());
    while (!aMsgs.empty() && k < sinks.size()) {
        Msg* msg = aMsgs.front();
        aMsgs.pop_front();
        MsgAudioPcm* pcm = dynamic_cast<MsgAudioPcm*>(msg);
        TEST(pcm != nullptr);
        if (pcm == nullptr) { msg->RemoveRef(); continue; }
        TEST(pcm->Ramp().IsEnabled() && pcm->Ramp().Direction() == Ramp::EDown && pcm->Ramp().Start() == lastEnd);
        lastEnd = pcm->Ramp().End();
        MsgPlayable* playable = pcm->CreatePlayable();
        ohp_msg_desc d;
        memset(&d, 0, sizeof(d));
        d.src_offset = (uint64_t)frames * frameBytes; d.dst_offset = d.src_offset; d.n_frames = playable->Bytes() / frameBytes;
        d.ramp_start = (uint16_t)playable->Ramp().Start(); d.ramp_end = (uint16_t)playable->Ramp().End(); d.attenuation = 256;
        d.channels = (uint8_t)cs.channels; d.src_bits = d.dst_bits = (uint8_t)cs.bits; d.src_endian = d.dst_endian = OHP_ENDIAN_BIG;
        d.flags = OHP_FLAG_RAMP;
        descs.push_back(d);
        frames += d.n_frames;
        batch.Add(playable, sinks[k++]);
    }
    TEST(frames == outFrames && lastEnd == Ramp::kMin);                          // the whole 20 ms, down to silence
    batch.Run();
    std::vector<TByte> got, want(packed.size());
    for (TUint j = 0; j < k; j++) got.insert(got.end(), sinks[j].Ptr(), sinks[j].Ptr() + sinks[j].Buf().Bytes());
    TEST(ohp_msg_process_batch(descs.data(), descs.size(), packed.data(), want.data()) == 0);
    TEST(got.size() == want.size() && memcmp(got.data(), want.data(), want.size()) == 0);
}

// What happens when streams run dry (StarvationRamper.cpp:491-537 -> FlywheelRamper.cpp:45-131 -> :281-364), for SEVERAL streams of
// different formats in ONE RescueBatch: the last millisecond of each is read, unpacked, extrapolated and packed in the same
// device passes; the generated messages are then read like any other audio.
static void SuiteStarvationRescueGpu(MsgFactory& aFactory)
{
    const RescueCase cases[] = {
        {44100, 2, 24, Ramp::kMax}, {48000, 2, 16, Ramp::kMax}, {96000, 6, 32, Ramp::kMax}, {44100, 1, 8, Ramp::kMax}, {192000, 8, 24, 9000} };
    const size_t n = sizeof cases / sizeof cases[0];
    uint32_t x = 4242;
    std::vector<std::vector<TByte>> newest(n);
    std::vector<std::deque<Msg*>> out(n);
    const TUint64 launches = RescueBatch::FlywheelLaunches();
    RescueBatch rescue(aFactory);
    for (size_t i = 0; i < n; i++) {
        const RescueCase& cs = cases[i];
        // the manager hands over one training window: here the window's one message, after 2 ms of earlier audio
        const TUint perMs = Jiffies::ToSamples(Jiffies::kPerMs, cs.rate);
        (void)RescueSignal(cs, x, 0, 2 * perMs);
        newest[i] = RescueSignal(cs, x, 2 * perMs, perMs);
        MsgAudioPcm* msg = aFactory.CreateMsgAudioPcm(Brn(newest[i].data(), (TUint)newest[i].size()), cs.channels, cs.rate, cs.bits, AudioDataEndian::Big, 0);
        RescueRequest rq;
        rq.audio.push_back(msg);
        rq.jiffies = msg->Jiffies();
        rq.sampleRate = cs.rate; rq.bitDepth = cs.bits; rq.channels = cs.channels; rq.rampValue = cs.startRamp;
        rq.out = &out[i];
        rescue.Add(std::move(rq));
    }
    TEST(rescue.Count() == n);
    rescue.Run();
    TEST(RescueBatch::FlywheelLaunches() == launches + 1);                         // five streams, one extrapolation launch
    for (size_t i = 0; i < n; i++) {
        CheckRescued(aFactory, cases[i], newest[i].data(), out[i]);
    }
}

// ------------------------------------------------------------------------------------------- many lanes, one tick (N4)
// 64 streams of mixed formats play through one StarvationManager and ALL run dry in the same driver period: the manager
// must rescue them with one chain of device passes (one flywheel launch), every lane's 20 ms must be what the reference
// would have produced for that lane alone, and every lane must then halt.
class ScriptedSource : public IPipelineElementUpstream {
public:
    void Push(Msg* aMsg) { { std::lock_guard<std::mutex> lock(iLock); iQueue.push_back(aMsg); } iCv.notify_one(); }
    Msg* Pull() override
    {
        std::unique_lock<std::mutex> lock(iLock);
        iCv.wait(lock, [this] { return !iQueue.empty(); });
        Msg* msg = iQueue.front();
        iQueue.pop_front();
        return msg;
    }
private:
    std::mutex iLock;
    std::condition_variable iCv;
    std::deque<Msg*> iQueue;
};

class CountingObserver : public IStarvationRamperObserver {
public:
    void NotifyStarvationRamperBuffering(TBool aBuffering) override { aBuffering ? iStarted++ : iStopped++; }
    std::atomic<TUint> iStarted{0}, iStopped{0};
};

// A stream that starves twice: the second rescue works in the device buffers the first one allocated (RescueArena) -- a rescue
// happens in the period in which its lane has nothing to play, and allocating in it was the advisor's finding.
static void SuiteRescueBuffersPersistGpu(MsgFactory& aFactory)
{
    ScriptedSource source;
    CountingObserver observer;
    StarvationManager manager(aFactory);
    StarvationManager::LaneConfig cfg;
    cfg.upstream = &source; cfg.observer = &observer;
    cfg.sizeJiffies = 200 * Jiffies::kPerMs; cfg.rampUpJiffies = 50 * Jiffies::kPerMs; cfg.maxStreamCount = 10;
    TEST(manager.AddLane(cfg) == 0);
    source.Push(aFactory.CreateMsgMode(ModeInfo()));
    DecodedStreamInfo info;
    info.iStreamId = 3; info.iBitDepth = 16; info.iSampleRate = 48000; info.iNumChannels = 2;
    source.Push(aFactory.CreateMsgDecodedStream(info));
    uint32_t x = 9;
    std::vector<TByte> pcm(4 * 96);                          // 2 ms of 48 kHz stereo S16
    auto feed = [&](TUint aMsgs) {
        TUint fed = 0;
        for (TUint m = 0; m < aMsgs; m++) {
            for (auto& b : pcm) { x = x * 1664525u + 1013904223u; b = (TByte)(x >> 24); }
            MsgAudioPcm* audio = aFactory.CreateMsgAudioPcm(Brn(pcm.data(), (TUint)pcm.size()), 2, 48000, 16, AudioDataEndian::Big, 0);
            fed += audio->Jiffies();
            source.Push(audio);
        }
        while (manager.SizeInJiffies(0) != fed) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    };
    std::vector<Msg*> out;
    auto play_until_halt = [&](TUint aAudioBefore) {        // the fed audio, then 20 ms of extrapolated audio, then the halt
        TUint audio = 0;
        for (TUint t = 0; t < 200; t++) {
            manager.Tick(out);
            TEST(out.size() == 1 && out[0] != nullptr);
            if (out[0] == nullptr) return;
            const MsgKind k = KindOf(out[0]);
            out[0]->RemoveRef();
            if (k == MsgKind::Halt) { TEST(audio == aAudioBefore + 20); return; }
            if (k == MsgKind::AudioPcm) audio++;
        }
        TEST(false);
    };
    feed(3);
    play_until_halt(3);
    TEST(manager.RescueLaunches() == 1 && manager.RescueAllocations() == 4);
    const TUint64 batchAllocs = manager.DeviceAllocations(); // what the first rescue's three batch objects took for descriptors and plans
    TEST(batchAllocs > 0);
    feed(40);                                                // 80 ms: through the 50 ms ramp up and on
    play_until_halt(40);
    TEST(manager.RescueLaunches() == 2);
    TEST(manager.RescueAllocations() == 4);                  // the second rescue allocated nothing: not its buffers,
    TEST(manager.DeviceAllocations() == batchAllocs);        // ... and not its batches' descriptors or plan arrays (the context's cache)
    source.Push(aFactory.CreateMsgQuit());
    for (TUint t = 0; t < 20000 && !manager.Finished(0); t++) {
        manager.Tick(out);
        if (out[0] != nullptr) out[0]->RemoveRef();
        else std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    TEST(manager.Finished(0));
}

// A playing lane whose inbox holds nothing but a message the StarvationRamper consumes (MsgMetaText: :718-724 keep it from
// downstream) has run dry all the same: the non-blocking Tick pops it, finds nothing behind it, and must rescue the lane in THAT
// period -- the reference blocks in DoDequeue here and ramps when the wait times out; a lane that sat the period out would drop to
// silence unramped and get its flywheel ramp a tick late.
static void SuiteLaneLeftWithMetaTextIsRescuedGpu(MsgFactory& aFactory)
{
    ScriptedSource source;
    CountingObserver observer;
    StarvationManager manager(aFactory);
    StarvationManager::LaneConfig cfg;
    cfg.upstream = &source; cfg.observer = &observer;
    cfg.sizeJiffies = 200 * Jiffies::kPerMs; cfg.rampUpJiffies = 50 * Jiffies::kPerMs; cfg.maxStreamCount = 10;
    TEST(manager.AddLane(cfg) == 0);
    source.Push(aFactory.CreateMsgMode(ModeInfo()));
    DecodedStreamInfo info;
    info.iStreamId = 5; info.iBitDepth = 16; info.iSampleRate = 48000; info.iNumChannels = 2;
    source.Push(aFactory.CreateMsgDecodedStream(info));
    uint32_t x = 77;
    std::vector<TByte> pcm(4 * 96);                          // 2 ms of 48 kHz stereo S16
    TUint fed = 0;
    for (TUint m = 0; m < 40; m++) {                         // 80 ms: through the 50 ms ramp up and on
        for (auto& b : pcm) { x = x * 1664525u + 1013904223u; b = (TByte)(x >> 24); }
        MsgAudioPcm* audio = aFactory.CreateMsgAudioPcm(Brn(pcm.data(), (TUint)pcm.size()), 2, 48000, 16, AudioDataEndian::Big, 0);
        fed += audio->Jiffies();
        source.Push(audio);
    }
    while (manager.SizeInJiffies(0) != fed) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    std::vector<Msg*> out;
    TUint audio = 0;
    for (TUint t = 0; t < 42; t++) {                         // mode, stream, 40 x audio
        manager.Tick(out);
        TEST(out.size() == 1 && out[0] != nullptr);
        if (out[0] == nullptr) return;
        audio += KindOf(out[0]) == MsgKind::AudioPcm ? 1 : 0;
        out[0]->RemoveRef();
    }
    TEST(audio == 40 && manager.RescueLaunches() == 0 && manager.State(0) == LaneState::Running);
    source.Push(aFactory.CreateMsgMetaText());
    while (manager.IsEmpty(0)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    manager.Tick(out);                                       // pops the meta text, finds the inbox dry, rescues: extrapolated audio, this period
    TEST(out.size() == 1 && out[0] != nullptr && KindOf(out[0]) == MsgKind::AudioPcm);
    TEST(manager.RescueLaunches() == 1);
    if (out[0] != nullptr) out[0]->RemoveRef();
    TUint more = 1;
    for (TUint t = 0; t < 100; t++) {
        manager.Tick(out);
        TEST(out[0] != nullptr);
        if (out[0] == nullptr) break;
        const MsgKind k = KindOf(out[0]);
        out[0]->RemoveRef();
        if (k == MsgKind::Halt) break;
        more++;
    }
    TEST(more == 20);                                        // the flywheel's 20 ms, then the halt
    source.Push(aFactory.CreateMsgQuit());
    for (TUint t = 0; t < 20000 && !manager.Finished(0); t++) {
        manager.Tick(out);
        if (out[0] != nullptr) out[0]->RemoveRef();
        else std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    TEST(manager.Finished(0));
}

// An idle stream must not hold the others' period up (the reference has one StarvationRamper and one driver thread per
// pipeline: a halted pipeline blocks only itself).  Lane 0 plays; lane 1 was halted by its source and then hears nothing;
// lane 2 is held at an occupancy gate its feeder cannot reach.  Every tick must come back with lane 0's message.  No device
// work: nothing here runs dry while it is playing.
static void SuiteIdleLaneDoesNotStallTheTick(MsgFactory& aFactory)
{
    const TUint kAudio = 4;
    std::vector<std::unique_ptr<ScriptedSource>> sources;
    CountingObserver observer;
    StarvationManager manager(aFactory);
    for (TUint l = 0; l < 3; l++) {
        sources.emplace_back(new ScriptedSource());
        StarvationManager::LaneConfig cfg;
        cfg.upstream = sources.back().get(); cfg.observer = &observer;
        cfg.sizeJiffies = 100 * Jiffies::kPerMs; cfg.rampUpJiffies = 50 * Jiffies::kPerMs; cfg.maxStreamCount = 10;
        TEST(manager.AddLane(cfg) == l);
        sources[l]->Push(aFactory.CreateMsgMode(ModeInfo()));
        DecodedStreamInfo info;
        info.iStreamId = 7 + l; info.iBitDepth = 16; info.iSampleRate = 44100; info.iNumChannels = 2;
        sources[l]->Push(aFactory.CreateMsgDecodedStream(info));
    }
    std::vector<TByte> pcm(4 * 88, 0x11);                    // 2 ms of 44.1 kHz stereo S16
    TUint fed = 0;
    for (TUint m = 0; m < kAudio; m++) {
        MsgAudioPcm* audio = aFactory.CreateMsgAudioPcm(Brn(pcm.data(), (TUint)pcm.size()), 2, 44100, 16, AudioDataEndian::Big, 0);
        fed += audio->Jiffies();
        sources[0]->Push(audio);
    }
    sources[0]->Push(aFactory.CreateMsgHalt());              // lane 0 ends in a halt of its own: it never runs dry while playing
    sources[1]->Push(aFactory.CreateMsgHalt());
    while (manager.SizeInJiffies(0) != fed || manager.IsEmpty(1) || manager.IsEmpty(2)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    std::this_thread::sleep_for(std::chrono::milliseconds(50));   // (the halts carry no jiffies: give the feeders a moment to pass them on)
    std::vector<Msg*> out;
    auto kinds = [&](MsgKind a, bool b_null, MsgKind b, bool c_null, MsgKind c) {
        manager.Tick(out);
        TEST(out.size() == 3);
        TEST(out[0] != nullptr && KindOf(out[0]) == a);
        TEST(b_null ? out[1] == nullptr : (out[1] != nullptr && KindOf(out[1]) == b));
        TEST(c_null ? out[2] == nullptr : (out[2] != nullptr && KindOf(out[2]) == c));
        for (Msg* m : out) if (m != nullptr) m->RemoveRef();
    };
    kinds(MsgKind::Mode, false, MsgKind::Mode, false, MsgKind::Mode);
    kinds(MsgKind::DecodedStream, false, MsgKind::DecodedStream, false, MsgKind::DecodedStream);
    manager.WaitForOccupancy(2, 10 * Jiffies::kPerMs);      // lane 2: a gate nothing will open for a while
    kinds(MsgKind::AudioPcm, false, MsgKind::Halt, true, MsgKind::Halt);
    for (TUint m = 1; m < kAudio; m++) {
        kinds(MsgKind::AudioPcm, true, MsgKind::Halt, true, MsgKind::Halt);   // lane 1 halted and silent, lane 2 gated: lane 0 still gets its period
    }
    TEST(manager.State(1) == LaneState::Halted && manager.RescueLaunches() == 0);
    kinds(MsgKind::Halt, true, MsgKind::Halt, true, MsgKind::Halt);
    // the idle lanes wake when something arrives: 12 ms of silence open lane 2's gate (silence never makes a lane "playing", so
    // it cannot run dry into a rescue), a quit ends each lane
    TUint fed2 = 0;
    for (TUint m = 0; m < 3; m++) {
        TUint size = 4 * Jiffies::kPerMs;
        MsgSilence* silence = aFactory.CreateMsgSilence(size, 44100, 16, 2);
        fed2 += silence->Jiffies();
        sources[2]->Push(silence);
    }
    for (TUint l = 0; l < 3; l++) sources[l]->Push(aFactory.CreateMsgQuit());
    while (manager.SizeInJiffies(2) != fed2) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    TUint quits = 0, silence2 = 0;
    for (TUint t = 0; t < 20000 && quits < 3; t++) {
        manager.Tick(out);
        for (TUint l = 0; l < 3; l++) {
            if (out[l] == nullptr) continue;
            const MsgKind k = KindOf(out[l]);
            TEST(k == MsgKind::Quit || (l == 2 && k == MsgKind::Silence));
            quits += k == MsgKind::Quit;
            silence2 += k == MsgKind::Silence;
            out[l]->RemoveRef();
        }
        if (quits < 3) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    TEST(quits == 3 && silence2 == 3 && manager.RescueLaunches() == 0);
    for (TUint l = 0; l < 3; l++) TEST(manager.Finished(l));
    manager.Tick(out);
    TEST(out[0] == nullptr && out[1] == nullptr && out[2] == nullptr);
}

static void SuiteManyLanesStarveTogetherGpu(MsgFactory& aFactory)
{
    const TUint kLanes = 64, kMsgs = 3;
    const RescueCase formats[] = { {44100, 2, 16, Ramp::kMax}, {48000, 2, 24, Ramp::kMax}, {96000, 2, 24, Ramp::kMax}, {44100, 1, 24, Ramp::kMax},
                                   {48000, 6, 16, Ramp::kMax}, {192000, 2, 32, Ramp::kMax}, {88200, 2, 8, Ramp::kMax} };
    const size_t kFormats = sizeof formats / sizeof formats[0];
    std::vector<std::unique_ptr<ScriptedSource>> sources;
    CountingObserver observer;
    std::vector<std::vector<TByte>> sent(kLanes);
    uint32_t x = 77;
    {
        StarvationManager manager(aFactory);
        TUint total[64];
        for (TUint l = 0; l < kLanes; l++) {
            sources.emplace_back(new ScriptedSource());
            StarvationManager::LaneConfig cfg;
            cfg.upstream = sources.back().get(); cfg.observer = &observer;
            cfg.sizeJiffies = 100 * Jiffies::kPerMs; cfg.rampUpJiffies = 50 * Jiffies::kPerMs; cfg.maxStreamCount = 10;
            TEST(manager.AddLane(cfg) == l);
            const RescueCase& cs = formats[l % kFormats];
            sources[l]->Push(aFactory.CreateMsgMode(ModeInfo()));
            DecodedStreamInfo info;
            info.iStreamId = 100 + l; info.iBitDepth = cs.bits; info.iSampleRate = cs.rate; info.iNumChannels = cs.channels;
            sources[l]->Push(aFactory.CreateMsgDecodedStream(info));
            const TUint perMsg = 2 * Jiffies::ToSamples(Jiffies::kPerMs, cs.rate) + l % 5;     // 2 ms and a few frames: ragged windows
            total[l] = 0;
            for (TUint m = 0; m < kMsgs; m++) {
                const std::vector<TByte> data = RescueSignal(cs, x, m * perMsg, perMsg);
                sent[l].insert(sent[l].end(), data.begin(), data.end());
                MsgAudioPcm* audio = aFactory.CreateMsgAudioPcm(Brn(data.data(), (TUint)data.size()), cs.channels, cs.rate, cs.bits, AudioDataEndian::Big, 0);
                total[l] += audio->Jiffies();
                sources[l]->Push(audio);
            }
        }
        for (TUint l = 0; l < kLanes; l++) {                 // every lane's feeder has taken what its source holds
            while (manager.SizeInJiffies(l) != total[l]) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        std::vector<Msg*> out;
        std::vector<RampContinuity> continuity(kLanes);      // RampValidator's rule behind every lane
        for (TUint t = 0; t < 2 + kMsgs; t++) {              // mode, stream, then the audio: nothing to rescue yet
            manager.Tick(out);
            TEST(out.size() == kLanes);
            for (TUint l = 0; l < kLanes; l++) {
                Msg* m = out[l];
                TEST(KindOf(m) == (t == 0 ? MsgKind::Mode : t == 1 ? MsgKind::DecodedStream : MsgKind::AudioPcm));
                if (t >= 2) continuity[l].Audio(static_cast<MsgAudioPcm*>(m)->Ramp());
                m->RemoveRef();
            }
        }
        TEST(manager.RescueLaunches() == 0);
        for (TUint l = 0; l < kLanes; l++) TEST(manager.State(l) == LaneState::Running && manager.IsEmpty(l));
        const TUint64 launches = RescueBatch::FlywheelLaunches();
        std::vector<std::deque<Msg*>> rescued(kLanes);
        // the driver's next periods: 1 ms of extrapolated audio per lane and tick (20 of them, 21 where a millisecond is not a
        // whole number of frames), then the lane's halt; a lane that has passed its quit on has nothing more to give
        std::vector<TUint> halted(kLanes, 0), quit(kLanes, 0);
        auto all_quit = [&] { for (TUint l = 0; l < kLanes; l++) if (!quit[l]) return false; return true; };
        for (TUint t = 0; t < 23 || (!all_quit() && t < 20000); t++) {   // (a tick waits for no feeder: a halted lane's quit may take a few more periods to arrive)
            if (t >= 23) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            manager.Tick(out);
            if (t == 0) {                                    // (the sources end only now: a lane whose quit is in sight is not rescued)
                for (TUint l = 0; l < kLanes; l++) sources[l]->Push(aFactory.CreateMsgQuit());
            }
            TEST(out.size() == kLanes);
            for (TUint l = 0; l < kLanes; l++) {
                if (out[l] == nullptr) { TEST(quit[l] == 1 || (halted[l] == 1 && !manager.Finished(l))); continue; }   // done, or halted and waiting for its feeder
                const MsgKind kind = KindOf(out[l]);
                if (kind == MsgKind::AudioPcm) {
                    TEST(!halted[l] && manager.State(l) == LaneState::FlywheelRamping);
                    continuity[l].Audio(static_cast<MsgAudioPcm*>(out[l])->Ramp());
                    rescued[l].push_back(out[l]);
                    continue;
                }
                if (kind == MsgKind::Halt) { TEST(!halted[l] && !quit[l] && manager.State(l) == LaneState::RampingUp); halted[l]++; }
                else if (kind == MsgKind::Quit) { TEST(halted[l] == 1); quit[l]++; }
                else TEST(false);
                out[l]->RemoveRef();
            }
        }
        TEST(manager.RescueLaunches() == 1);                 // ONE rescue for 64 lanes...
        TEST(manager.RescueAllocations() == 4);              // ... in the manager's four persistent device buffers (allocated by this first rescue)
        TEST(RescueBatch::FlywheelLaunches() == launches + 1);                     // ...and one flywheel launch on the device
        for (TUint l = 0; l < kLanes; l++) TEST(halted[l] == 1 && quit[l] == 1 && continuity[l].Checked() >= kMsgs + 20);
        TEST(observer.iStarted.load() == 2 * kLanes && observer.iStopped.load() == kLanes);     // buffering at start, playing, buffering again
        for (TUint l = 0; l < kLanes; l++) {
            const RescueCase& cs = formats[l % kFormats];
            const size_t window = (size_t)Jiffies::ToSamples(Jiffies::kPerMs, cs.rate) * cs.channels * cs.bits / 8;
            CheckRescued(aFactory, cs, sent[l].data() + sent[l].size() - window, rescued[l]);
        }
    }
}


// ------------------------------------------------------------------------------------------- Songcast sender (N3)
// Messages pushed at the Sender element the way the pipeline does (Sender.cpp:117-123); the datagrams it hands to the sink
// are compared with the oracle's Sender + OhmSenderDriver + OhmMsgAudio fed the same messages.
namespace {

struct DatagramCollector : public Av::IOhmDatagramSink {
    std::vector<std::vector<TByte>> grams;
    void Send(const Brx& aDatagram) override { grams.emplace_back(aDatagram.Ptr(), aDatagram.Ptr() + aDatagram.Bytes()); }
};

struct SongcastStream {
    TUint rate, channels, bits;
    AudioDataEndian endian;
    const char* codec;
    TUint latencyMs;
    TUint64 sampleStart, trackLengthJiffies;
    // what was pushed, for the oracle
    std::vector<ohp_msg_audio> msgs;
    std::vector<std::vector<TByte>> audio;               // big endian, empty for silence
    DatagramCollector sink;
};

// Pushes a MsgDecodedStream and n audio messages (some ramped across message boundaries, some silence) at aSender and
// records the same messages for the oracle.
void FeedStream(MsgFactory& aFactory, Av::Sender& aSender, SongcastStream& aStream, TUint aMsgs, uint32_t& aSeed)
{
    DecodedStreamInfo info;
    info.iBitDepth = aStream.bits; info.iSampleRate = aStream.rate; info.iNumChannels = aStream.channels;
    info.iBitRate = aStream.rate * aStream.bits * aStream.channels;
    info.iTrackLength = aStream.trackLengthJiffies; info.iSampleStart = aStream.sampleStart;
    info.iLossless = true; info.iCodecName = aStream.codec;
    aSender.Push(aFactory.CreateMsgDecodedStream(info));
    const TUint frameBytes = aStream.channels * aStream.bits / 8;
    TUint rampRemaining = 0, rampValue = Ramp::kMax;
    Ramp::EDirection rampDir = Ramp::ENone;
    auto record = [&](MsgAudio* aMsg, const std::vector<TByte>& aBigEndian, TUint aOffsetBytes, TBool aSilence) {
        ohp_msg_audio m;
        if (aSilence) {
            uint32_t j = aMsg->Jiffies();
            TEST(ohp_msg_audio_init_silence(&m, &j, aStream.rate, aStream.bits, aStream.channels) == 0 && j == aMsg->Jiffies());
        }
        else {
            TEST(ohp_msg_audio_init_pcm(&m, (uint32_t)aBigEndian.size(), aStream.channels, aStream.rate, aStream.bits) == 0);
            m.offset_jiffies = (aOffsetBytes / frameBytes) * Jiffies::PerSample(aStream.rate);
            m.size_jiffies = aMsg->Jiffies();
        }
        m.ramp.start = aMsg->Ramp().Start(); m.ramp.end = aMsg->Ramp().End();
        m.ramp.direction = (uint32_t)aMsg->Ramp().Direction(); m.ramp.enabled = aMsg->Ramp().IsEnabled() ? 1 : 0;
        aStream.msgs.push_back(m);
        aStream.audio.push_back(aSilence ? std::vector<TByte>() : aBigEndian);
    };
    for (TUint k = 0; k < aMsgs; k++) {
        aSeed = aSeed * 1664525u + 1013904223u;
        const TUint maxFrames = std::min(aStream.rate / 60, DecodedAudio::kMaxBytes / frameBytes);
        const TUint frames = 1 + (aSeed >> 8) % maxFrames;
        const TUint kind = (aSeed >> 3) % 9;
        MsgAudio* msg;
        std::vector<TByte> be;
        const TBool silence = (kind == 0);
        if (silence) {
            TUint jiffies = frames * Jiffies::PerSample(aStream.rate);
            msg = aFactory.CreateMsgSilence(jiffies, aStream.rate, aStream.bits, aStream.channels);
        }
        else {
            std::vector<TByte> data((size_t)frames * frameBytes);
            for (auto& b : data) { aSeed = aSeed * 1664525u + 1013904223u; b = (TByte)(aSeed >> 24); }
            be.resize(data.size());
            TEST(ohp_construct_pcm(data.data(), (uint32_t)data.size(), aStream.bits,
                                   aStream.endian == AudioDataEndian::Little ? OHP_ENDIAN_LITTLE : OHP_ENDIAN_BIG, be.data()) == 0);
            msg = aFactory.CreateMsgAudioPcm(Brn(data.data(), (TUint)data.size()), aStream.channels, aStream.rate, aStream.bits, aStream.endian, 0);
        }
        if (rampRemaining == 0 && kind == 1) {           // start a ~12 ms ramp down, then one back up: they cross message boundaries
            rampRemaining = (aStream.rate / 83) * Jiffies::PerSample(aStream.rate);        // whole samples, so every split lands on one
            rampDir = (rampValue == Ramp::kMax) ? Ramp::EDown : Ramp::EUp;
        }
        if (rampRemaining > 0) {
            if (msg->Jiffies() > rampRemaining) {        // the ramp ends inside this message: what Ramper / Stopper do
                MsgAudio* rest = msg->Split(rampRemaining);
                MsgAudio* split = nullptr;
                rampValue = msg->SetRamp(rampValue, rampRemaining, rampDir, split);
                TEST(split == nullptr && rampRemaining == 0);
                const TUint firstBytes = (TUint)((TUint64)msg->Jiffies() / Jiffies::PerSample(aStream.rate)) * frameBytes;
                record(msg, be, 0, silence);
                aSender.Push(msg);
                record(rest, be, firstBytes, silence);
                aSender.Push(rest);
                continue;
            }
            MsgAudio* split = nullptr;
            rampValue = msg->SetRamp(rampValue, rampRemaining, rampDir, split);
            TEST(split == nullptr);
        }
        record(msg, be, 0, silence);
        aSender.Push(msg);
    }
}

// The oracle's datagrams for what FeedStream recorded.
std::vector<std::vector<TByte>> OracleDatagrams(const SongcastStream& aStream, bool aFlush)
{
    std::vector<std::vector<TByte>> out;
    ohp_ohm_driver d;
    ohp_ohm_driver_init(&d, aStream.latencyMs);
    const TUint64 samplesTotal = aStream.trackLengthJiffies / Jiffies::PerSample(aStream.rate);
    ohp_ohm_driver_set_track_position(&d, samplesTotal, aStream.sampleStart);
    const TUint wireCh = aStream.channels < 2 ? aStream.channels : 2, wireBits = aStream.bits < 24 ? aStream.bits : 24;
    TEST(ohp_ohm_driver_set_audio_format(&d, aStream.rate, aStream.rate * aStream.bits * aStream.channels, wireCh, wireBits, 1,
                                         (const uint8_t*)aStream.codec, (uint32_t)strlen(aStream.codec), aStream.sampleStart) == 0);
    std::vector<ohp_sender_fragment> frags(aStream.msgs.size() * 8 + 64);
    std::vector<ohp_sender_packet> packs(aStream.msgs.size() * 8 + 64);
    uint32_t nf = 0, np = 0;
    TEST(ohp_sender_packetise(aStream.msgs.data(), (uint32_t)aStream.msgs.size(), aFlush ? 1 : 0, frags.data(), (uint32_t)frags.size(), &nf,
                              packs.data(), (uint32_t)packs.size(), &np) == 0);
    for (uint32_t k = 0; k < np; k++) {
        std::vector<TByte> payload;
        for (uint32_t g = 0; g < packs[k].n_fragments; g++) {
            const ohp_sender_fragment& f = frags[packs[k].first_fragment + g];
            if (f.playable.size_bytes == 0) continue;
            std::vector<TByte> audio = aStream.audio[f.msg];                          // Read attenuates in place: a copy
            std::vector<TByte> pcm(f.playable.size_bytes), packed(f.playable.size_bytes);
            uint32_t n_frags = 0, bytes = 0, packedBytes = 0;
            TEST(ohp_playable_read(&f.playable, audio.empty() ? nullptr : audio.data(), pcm.data(), (uint32_t)pcm.size(), nullptr, 0, &n_frags, &bytes) == 0);
            TEST(ohp_sender_pack(pcm.data(), bytes, f.playable.channels, f.playable.bit_depth / 8, packed.data(), &packedBytes) == 0);
            payload.insert(payload.end(), packed.begin(), packed.begin() + packedBytes);
        }
        std::vector<TByte> gram(8192);
        const bool halt = aFlush && k == np - 1;                                        // ProcessMsg(MsgQuit*) sends with aHalt = true
        const int n = ohp_ohm_driver_send_audio(&d, payload.empty() ? gram.data() : payload.data(), (uint32_t)payload.size(), halt ? 1 : 0,
                                                gram.data(), (uint32_t)gram.size());
        TEST(n >= 0);
        if (n > 0) { gram.resize((size_t)n); out.push_back(gram); }
    }
    return out;
}

} // namespace

static void SuiteSongcastSenderControl(MsgFactory& aControl)
{   // no GPU: the packetiser and the driver's counters against the oracle; nothing is read
    SongcastStream s = { 44100, 2, 16, AudioDataEndian::Big, "FLAC", 100, 1000, 0, {}, {}, {} };
    s.trackLengthJiffies = (TUint64)Jiffies::kPerSecond * 60;
    Av::Sender sender(aControl, s.sink, s.latencyMs);
    sender.SetBatching(0);                                // never run the device
    uint32_t seed = 99;
    FeedStream(aControl, sender, s, 50, seed);
    std::vector<ohp_sender_fragment> frags(1024);
    std::vector<ohp_sender_packet> packs(1024);
    uint32_t nf = 0, np = 0;
    TEST(ohp_sender_packetise(s.msgs.data(), (uint32_t)s.msgs.size(), 0, frags.data(), 1024, &nf, packs.data(), 1024, &np) == 0);
    uint64_t samples = 0; uint32_t sent = 0;
    for (uint32_t k = 0; k < np; k++) {
        uint32_t bytes = 0;
        for (uint32_t g = 0; g < packs[k].n_fragments; g++) bytes += frags[packs[k].first_fragment + g].playable.size_bytes;
        samples += bytes / 4;
        sent += bytes ? 1 : 0;
    }
    TEST(np > 20 && sender.Driver().Frame() == sent && sender.Driver().SampleStart() == s.sampleStart + samples);
    TEST(s.sink.grams.empty());
    TByte b[4] = { 0 };
    {   // (a message the element refuses stays the caller's: released here, so that the suite is clean under LeakSanitizer too)
        Msg* refused = aControl.CreateMsgAudioPcm(Brn(b, 4), 2, 44100, 16, AudioDataEndian::Big, 0)->CreatePlayable();
        TEST_THROWS(sender.Push(refused), AssertionFailed);                             // Sender.cpp:264-268
        refused->RemoveRef();
    }

    // the messages that cut a packet short, and what they do to the frame counter
    std::vector<TByte> ms(48 * 4, 0x11);                                               // 1 ms of 48 kHz 16-bit stereo
    auto audio = [&]() { return aControl.CreateMsgAudioPcm(Brn(ms.data(), (TUint)ms.size()), 2, 48000, 16, AudioDataEndian::Big, 0); };
    DatagramCollector sink2;
    Av::Sender s2(aControl, sink2, 50);
    s2.SetBatching(0);
    {
        Msg* early = audio();
        TEST_THROWS(s2.Push(early), AssertionFailed);                                  // audio before any MsgDecodedStream: ASSERT(iSampleRate != 0), Sender.cpp:246
        early->RemoveRef();
    }
    DecodedStreamInfo info;
    info.iBitDepth = 16; info.iSampleRate = 48000; info.iNumChannels = 2; info.iSampleStart = 7;
    s2.Push(aControl.CreateMsgDecodedStream(info));
    TEST(s2.Driver().Frame() == 0 && s2.Driver().SampleStart() == 7);                  // SetAudioFormat: iSampleStart (OhmSender.cpp:334); nothing pending, nothing sent
    for (int k = 0; k < 3; k++) s2.Push(audio());                                      // 3 ms pending: below a packet
    TEST(s2.Driver().Frame() == 0);
    s2.Push(aControl.CreateMsgHalt());                                                 // SendPendingAudio(true): the 3 ms leave as a (halt) frame, Sender.cpp:196-201
    TEST(s2.Driver().Frame() == 1 && s2.Driver().SampleStart() == 7 + 3 * 48);
    s2.Push(aControl.CreateMsgHalt());                                                 // nothing pending, but a halt frame is still sent (samples == 0 && aHalt), OhmSender.cpp:434
    TEST(s2.Driver().Frame() == 2 && s2.Driver().SampleStart() == 7 + 3 * 48);
    s2.Push(audio());
    s2.Push(aControl.CreateMsgStreamInterrupted());                                    // pending audio out, then the gap receivers resync on (:187-193, OhmSender.cpp:482-488)
    TEST(s2.Driver().Frame() == 3 + 250);
    for (int k = 0; k < 12; k++) s2.Push(audio());                                     // 12 ms: two whole packets, 2 ms left pending
    TEST(s2.Driver().Frame() == 3 + 250 + 2);
    s2.Push(aControl.CreateMsgTrack());                                                // SendPendingAudio(): the 2 ms go out, Sender.cpp:146-152
    TEST(s2.Driver().Frame() == 3 + 250 + 3);
    info.iMultiroom = Multiroom::Forbidden;                                            // a stream that may not be sent: its audio is dropped (:281-284)
    s2.Push(aControl.CreateMsgDecodedStream(info));
    for (int k = 0; k < 12; k++) s2.Push(audio());
    TEST(s2.Driver().Frame() == 3 + 250 + 3);
    s2.Driver().SetEnabled(false);                                                     // turning the sender off resets the frame counter (ResetLocked, :623-635)
    TEST(s2.Driver().Frame() == 0);
}

static void SuiteSongcastSenderGpu(MsgFactory& aFactory)
{
    SongcastStream streams[] = {
        { 48000, 2, 24, AudioDataEndian::Big, "FLAC", 100, 0, 0, {}, {}, {} },
        { 44100, 2, 16, AudioDataEndian::Little, "WAV", 250, 123456, 0, {}, {}, {} },
        { 96000, 2, 32, AudioDataEndian::Big, "", 50, 7, 0, {}, {}, {} },
        { 48000, 6, 24, AudioDataEndian::Big, "PCM", 100, 0, 0, {}, {}, {} },
        { 44100, 1, 24, AudioDataEndian::Little, "ALAC", 100, 1ull << 33, 0, {}, {}, {} },
        { 48000, 8, 16, AudioDataEndian::Little, "AIFF", 100, 0, 0, {}, {}, {} },
    };
    uint32_t seed = 2024;
    // ---- each stream through its own Sender, the device running after every packet (the reference's cadence) ----
    {
        SongcastStream& s = streams[0];
        s.trackLengthJiffies = (TUint64)Jiffies::kPerSecond * 200;
        Av::Sender sender(aFactory, s.sink, s.latencyMs);
        FeedStream(aFactory, sender, s, 30, seed);
        sender.Push(aFactory.CreateMsgQuit());
        const auto want = OracleDatagrams(s, true);
        TEST(s.sink.grams.size() == want.size() && !want.empty());
        for (size_t i = 0; i < want.size() && i < s.sink.grams.size(); i++) TEST(s.sink.grams[i] == want[i]);
    }
    // ---- all streams sharing one device pass ----
    Av::OhmFrameBatch batch(aFactory);
    std::vector<Av::Sender*> senders;
    for (auto& s : streams) {
        s.msgs.clear(); s.audio.clear(); s.sink.grams.clear();
        s.trackLengthJiffies = (TUint64)Jiffies::kPerSecond * 300;
        senders.push_back(new Av::Sender(aFactory, s.sink, s.latencyMs, &batch));
        FeedStream(aFactory, *senders.back(), s, 40, seed);
        senders.back()->Push(aFactory.CreateMsgQuit());
    }
    TEST(batch.Count() > 100 && streams[1].sink.grams.empty());
    batch.Run();
    TEST(batch.Count() == 0);
    for (auto& s : streams) {
        const auto want = OracleDatagrams(s, true);
        TEST(s.sink.grams.size() == want.size() && !want.empty());
        size_t bad = 0;
        for (size_t i = 0; i < want.size() && i < s.sink.grams.size(); i++) bad += (s.sink.grams[i] == want[i]) ? 0 : 1;
        TEST(bad == 0);
        if (bad) printf("  stream %u Hz %u ch %u bit: %zu of %zu datagrams differ\n", s.rate, s.channels, s.bits, bad, want.size());
    }
    for (auto* p : senders) delete p;
}

// ------------------------------------------------------------------------------------------- batch builder (N4, first half)
// Tests/TestDecodedAudioAggregator.cpp restated: the suite is the downstream element and inspects what comes out.
namespace {

class SuiteDecodedAudioAggregator : public IPipelineElementDownstream, private IMsgProcessor {
    static const TUint kSampleRate = 44100, kChannels = 2, kBitDepth = 16;     // TestDecodedAudioAggregator.cpp:95-98
    enum EMsgType { ENone, EMsgMode, EMsgTrack, EMsgDrain, EMsgEncodedStream, EMsgDecodedStream, EMsgAudioPcm, EMsgHalt, EMsgFlush, EMsgWait, EMsgQuit, EMsgOther };
public:
    explicit SuiteDecodedAudioAggregator(MsgFactory& aFactory) : iFactory(aFactory) {}
    void Run()
    {
        TestStreamSuccessful(); TestNoDataAfterDecodedStream(); TestShortStream(); TestTrackEncodedStreamTrack();
        TestPcmIsExpectedSize(); TestRawPcmNotAggregated(); TestCodecControllerChunks();
    }
private:
    void Push(Msg* aMsg) override { iReceived.push_back(aMsg); }
    void Setup()
    {
        for (auto* m : iReceived) m->RemoveRef();
        iReceived.clear();
        iAggregator.reset(new DecodedAudioAggregator(*this));
        iTrackOffset = 0; iTrackOffsetBytes = 0; iJiffies = 0; iLast = ENone;
    }
    void Queue(Msg* aMsg) { iAggregator->Push(aMsg); }
    void PullNext(EMsgType aExpected)
    {
        TEST(!iReceived.empty());
        if (iReceived.empty()) return;
        Msg* msg = iReceived.front();
        iReceived.pop_front();
        msg = msg->Process(*this);
        msg->RemoveRef();
        TEST(iLast == aExpected);
    }
    void PullNext(EMsgType aExpected, TUint64 aExpectedJiffies)
    {
        const TUint64 start = iJiffies;
        PullNext(aExpected);
        TEST(iJiffies - start == aExpectedJiffies);
    }
    MsgDecodedStream* CreateDecodedStream()
    {
        DecodedStreamInfo info;
        info.iBitRate = 256; info.iBitDepth = kBitDepth; info.iSampleRate = kSampleRate; info.iNumChannels = kChannels; info.iCodecName = "Dummy";
        return iFactory.CreateMsgDecodedStream(info);
    }
    MsgAudioPcm* CreateAudio(TUint aBytes, TUint aSampleRate = kSampleRate, TUint aBitDepth = kBitDepth, TUint aNumChannels = kChannels)
    {
        std::vector<TByte> data(aBytes, 0x7f);
        MsgAudioPcm* audio = iFactory.CreateMsgAudioPcm(Brn(data.data(), aBytes), aNumChannels, aSampleRate, aBitDepth, AudioDataEndian::Little, iTrackOffset);
        const TUint samples = aBytes / (aNumChannels * (aBitDepth / 8));
        iTrackOffset += (TUint64)samples * (Jiffies::kPerSecond / aSampleRate);
        iTrackOffsetBytes += aBytes;
        return audio;
    }
    void StartStream()
    {
        Queue(iFactory.CreateMsgTrack());          PullNext(EMsgTrack);
        Queue(iFactory.CreateMsgEncodedStream());  PullNext(EMsgEncodedStream);
        Queue(CreateDecodedStream());              PullNext(EMsgDecodedStream);
    }
    void TestStreamSuccessful()
    {   // :426-448
        Setup();
        const TUint kMaxMsgBytes = DecodedAudio::kMaxBytes, kAudioBytes = DecodedAudio::kMaxBytes * 5;
        StartStream();
        while (iTrackOffsetBytes < kAudioBytes) Queue(CreateAudio(kMaxMsgBytes));
        for (int i = 0; i < 5; i++) PullNext(EMsgAudioPcm);
        TEST(iTrackOffsetBytes == kAudioBytes && iJiffies == iTrackOffset && iReceived.empty());
    }
    void TestNoDataAfterDecodedStream()
    {   // :450-464
        Setup();
        StartStream();
        Queue(iFactory.CreateMsgTrack()); PullNext(EMsgTrack);
        TEST(iJiffies == iTrackOffset);
    }
    void TestShortStream()
    {   // :466-481
        Setup();
        StartStream();
        Queue(CreateAudio(DecodedAudio::kMaxBytes));
        PullNext(EMsgAudioPcm);
        TEST(iJiffies == iTrackOffset);
    }
    void TestTrackEncodedStreamTrack()
    {   // :483-502
        Setup();
        Queue(iFactory.CreateMsgTrack()); PullNext(EMsgTrack);
        Queue(iFactory.CreateMsgTrack()); PullNext(EMsgTrack);
        Queue(iFactory.CreateMsgEncodedStream()); PullNext(EMsgEncodedStream);
        Queue(iFactory.CreateMsgTrack()); PullNext(EMsgTrack);
    }
    void TestPcmIsExpectedSize()
    {   // :504-546
        Setup();
        const TUint kMaxMsgBytes = 64, kSamplesPerMsg = 16;
        const TUint kAudioBytes = DecodedAudio::kMaxBytes - (DecodedAudio::kMaxBytes % kMaxMsgBytes);
        const TUint64 kJiffiesPerMsg = (TUint64)(Jiffies::kPerSecond / kSampleRate) * kSamplesPerMsg;
        const TUint kMaxDecodedBufferedJiffies = Jiffies::kPerMs * 5;
        const TUint kRemainderJiffies = (TUint)(kMaxDecodedBufferedJiffies % kJiffiesPerMsg);
        const TUint64 kExpectedJiffiesPerMsg = kMaxDecodedBufferedJiffies - kRemainderJiffies + kJiffiesPerMsg;
        TEST(kMaxDecodedBufferedJiffies % kJiffiesPerMsg != 0);
        StartStream();
        while (iTrackOffsetBytes < kAudioBytes) Queue(CreateAudio(kMaxMsgBytes));
        Queue(iFactory.CreateMsgEncodedStream());       // flush out remaining audio
        while (iJiffies < iTrackOffset - kMaxDecodedBufferedJiffies) PullNext(EMsgAudioPcm, kExpectedJiffiesPerMsg);
        if (iJiffies < iTrackOffset) PullNext(EMsgAudioPcm, iTrackOffset - iJiffies);
        PullNext(EMsgEncodedStream);
        TEST(iTrackOffsetBytes == kAudioBytes && iJiffies == iTrackOffset);
    }
    void TestRawPcmNotAggregated()
    {   // :548-567
        Setup();
        ModeInfo info;
        info.SetLatencyMode(Latency::Internal);
        Queue(iFactory.CreateMsgMode(info));
        Queue(iFactory.CreateMsgTrack());
        Queue(iFactory.CreateMsgEncodedStream(MsgEncodedStream::Format::Pcm));
        DecodedStreamInfo ds;
        ds.iBitDepth = 32; ds.iSampleRate = 48000; ds.iNumChannels = 2;
        Queue(iFactory.CreateMsgDecodedStream(ds));
        Queue(CreateAudio(8, 48000, 32, 2));             // one sample for 32-bit stereo
        PullNext(EMsgMode); PullNext(EMsgTrack); PullNext(EMsgEncodedStream); PullNext(EMsgDecodedStream);
        PullNext(EMsgAudioPcm);
        TEST(iJiffies == Jiffies::PerSample(48000) && iJiffies == iTrackOffset);
    }
    void TestCodecControllerChunks()
    {   // CodecController::OutputAudioPcm, CodecController.cpp:799-826: a codec's block leaves in pieces of at most
        // iMaxOutputJiffies (whole samples), the track offset running through; then the aggregator sees them
        Setup();
        const TUint kMaxOutputJiffies = 2 * Jiffies::kPerMs;
        CodecController controller(iFactory, *iAggregator, kMaxOutputJiffies);
        controller.OutputDecodedStream(1411200, 16, 44100, 2, Brn((const TByte*)"WAV", 3), 0, 0, true);
        PullNext(EMsgDecodedStream);
        TEST(controller.MaxOutputBytes() == Jiffies::ToSamples(kMaxOutputJiffies, 44100) * 4);            // 88 samples
        std::vector<TByte> block(4000 * 4, 0x11);                                                          // 4000 samples in one call
        const TUint64 out = controller.OutputAudioPcm(Brn(block.data(), (TUint)block.size()), 2, 44100, 16, AudioDataEndian::Little, 0);
        TEST(out == (TUint64)4000 * Jiffies::PerSample(44100));
        TEST_THROWS(controller.OutputAudioPcm(Brn(block.data(), 16), 2, 48000, 16, AudioDataEndian::Little, 0), AssertionFailed);
        TEST(controller.OutputAudioPcm(Brn(block.data(), 0), 2, 44100, 16, AudioDataEndian::Little, 0) == 0);
        Queue(iFactory.CreateMsgQuit());
        // 88-sample pieces aggregate to 3 x 88 = 264 samples >= kMaxJiffies (5 ms less one 7350 Hz sample = 212.8 samples at 44.1 kHz)
        TUint64 total = 0; TUint n = 0;
        while (iReceived.size() > 1) {
            const TUint64 before = iJiffies;
            PullNext(EMsgAudioPcm);
            const TUint64 got = iJiffies - before;
            total += got; n++;
            if (iReceived.size() > 1) TEST(got == (TUint64)264 * Jiffies::PerSample(44100));
        }
        PullNext(EMsgQuit);
        TEST(total == out && n == (4000 + 263) / 264);
        TEST_THROWS(controller.OutputDecodedStream(0, 16, 44101, 2, Brn(), 0, 0, true), CodecStreamFeatureUnsupported);
    }
private: // IMsgProcessor
    Msg* ProcessMsg(MsgMode* aMsg) override { iLast = EMsgMode; return aMsg; }
    Msg* ProcessMsg(MsgTrack* aMsg) override { iLast = EMsgTrack; return aMsg; }
    Msg* ProcessMsg(MsgDrain* aMsg) override { iLast = EMsgDrain; return aMsg; }
    Msg* ProcessMsg(MsgDelay* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override { iLast = EMsgEncodedStream; return aMsg; }
    Msg* ProcessMsg(MsgStreamSegment* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgAudioEncoded* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgMetaText* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgHalt* aMsg) override { iLast = EMsgHalt; return aMsg; }
    Msg* ProcessMsg(MsgFlush* aMsg) override { iLast = EMsgFlush; return aMsg; }
    Msg* ProcessMsg(MsgWait* aMsg) override { iLast = EMsgWait; return aMsg; }
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override { iLast = EMsgDecodedStream; return aMsg; }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override
    {
        iLast = EMsgAudioPcm;
        TEST(aMsg->TrackOffset() == iJiffies);           // aggregated messages keep the running track offset
        iJiffies += aMsg->Jiffies();
        return aMsg;
    }
    Msg* ProcessMsg(MsgAudioDsd* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgSilence* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgPlayable* aMsg) override { iLast = EMsgOther; return aMsg; }
    Msg* ProcessMsg(MsgQuit* aMsg) override { iLast = EMsgQuit; return aMsg; }
private:
    MsgFactory& iFactory;
    std::unique_ptr<DecodedAudioAggregator> iAggregator;
    std::deque<Msg*> iReceived;
    TUint64 iTrackOffset = 0, iTrackOffsetBytes = 0, iJiffies = 0;
    EMsgType iLast = ENone;
};

} // namespace

// ------------------------------------------------------------------------------------------- batch builder (N4, second half)
// Tests/TestStarvationRamper.cpp restated (the PCM cases): the suite is the upstream element (its Pull blocks until the
// test has queued a message), the stream handler and the observer; messages pulled from the StarvationRamper are
// inspected one by one.  Cases that starve need the device (FlywheelInput / RampGenerator run there).
namespace {

class SuiteStarvationRamper : private IPipelineElementUpstream, private IMsgProcessor, private IStreamHandler,
                              private IStarvationRamperObserver {
    static const TUint kMaxAudioBuffer = Jiffies::kPerMs * 100;          // TestStarvationRamper.cpp:25-33
    static const TUint kRampUpDuration = Jiffies::kPerMs * 50;
    static const TUint kSampleRateDefault = 48000, kBitDepthDefault = 16, kNumChannels = 2;
    static const TUint kAudioPcmBytesDefault = 960;                       // 5 ms of 48k, 16-bit stereo
    enum EMsgType { ENone, EMsgMode, EMsgTrack, EMsgDrain, EMsgDelay, EMsgEncodedStream, EMsgMetaText, EMsgStreamInterrupted,
                    EMsgDecodedStream, EMsgAudioPcm, EMsgAudioDsd, EMsgSilence, EMsgHalt, EMsgFlush, EMsgWait, EMsgQuit };
    typedef StarvationRamper::State State;
public:
    explicit SuiteStarvationRamper(MsgFactory& aFactory) : iFactory(aFactory) {}
    void RunControl()
    {   // no starvation in these: nothing is read, no device needed
        Run(&SuiteStarvationRamper::TestMsgsPassWhenRunning);
        Run(&SuiteStarvationRamper::TestBlocksWhenHasMaxAudio);
        Run(&SuiteStarvationRamper::TestNoRampAroundHalt);
        Run(&SuiteStarvationRamper::TestFlush);
        Run(&SuiteStarvationRamper::TestPruneMsgsNotReqdDownstream);
    }
    void RunGpu()
    {
        Run(&SuiteStarvationRamper::TestRampBeforeDrain);
        Run(&SuiteStarvationRamper::TestRampsAroundStarvation);
        Run(&SuiteStarvationRamper::TestNotifyStarvingAroundStarvation);
        Run(&SuiteStarvationRamper::TestReportsBuffering);
        Run(&SuiteStarvationRamper::TestDrainAllAudio);
        Run(&SuiteStarvationRamper::TestAllSampleRates);
    }
private:
    void Run(void (SuiteStarvationRamper::*aTest)()) { Setup(); (this->*aTest)(); TearDown(); }
    void Setup()
    {
        iStreamId = UINT32_MAX; iTrackOffset = 0; iJiffies = 0;
        iRampingUp = iRampingDown = iBuffering = false;
        iLastRampPos = Ramp::kMax; iNextStreamId = 1; iStarving = false; iStarvingStreamId = IStreamHandler::kStreamIdInvalid;
        iSampleRate = kSampleRateDefault; iBitDepth = kBitDepthDefault; iLastPulledMsg = ENone;
        iPcmData.assign(kAudioPcmBytesDefault, 0);                         // left = 0x7f7f, right = 0x0000
        for (size_t i = 0; i < iPcmData.size(); i += 4) { iPcmData[i] = 0x7f; iPcmData[i + 1] = 0x7f; }
        (void)iMsgAvailable.Clear();
        iStarvationRamper = new StarvationRamper(iFactory, *this, *this, kMaxAudioBuffer, kRampUpDuration, 10);
    }
    void TearDown()
    {
        delete iStarvationRamper;
        for (auto* m : iPendingMsgs) m->RemoveRef();
        iPendingMsgs.clear();
    }
private: // from IPipelineElementUpstream
    Msg* Pull() override
    {
        iMsgAvailable.Wait();
        std::lock_guard<std::mutex> lock(iPendingMsgLock);
        Msg* msg = iPendingMsgs.front();
        iPendingMsgs.pop_front();
        return msg;
    }
private: // from IStreamHandler / IStarvationRamperObserver
    void NotifyStarving(const Brx& aMode, TUint aStreamId, TBool aStarving) override
    {
        TEST(aMode.Bytes() == 9 && memcmp(aMode.Ptr(), "DummyMode", 9) == 0);
        iStarving = aStarving;
        iStarvingStreamId = aStreamId;
    }
    void NotifyStarvationRamperBuffering(TBool aBuffering) override { iBuffering = aBuffering; }
private: // from IMsgProcessor
    Msg* ProcessMsg(MsgMode* aMsg) override { iLastPulledMsg = EMsgMode; iContinuity.NewStream(); return aMsg; }
    Msg* ProcessMsg(MsgTrack* aMsg) override { iLastPulledMsg = EMsgTrack; return aMsg; }
    Msg* ProcessMsg(MsgDrain* aMsg) override { iLastPulledMsg = EMsgDrain; iContinuity.Drain(); return aMsg; }
    Msg* ProcessMsg(MsgDelay* aMsg) override { iLastPulledMsg = EMsgDelay; return aMsg; }
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override { iLastPulledMsg = EMsgEncodedStream; return aMsg; }
    Msg* ProcessMsg(MsgStreamSegment* aMsg) override { ASSERTS(); return aMsg; }
    Msg* ProcessMsg(MsgAudioEncoded* aMsg) override { ASSERTS(); return aMsg; }
    Msg* ProcessMsg(MsgMetaText* aMsg) override { iLastPulledMsg = EMsgMetaText; return aMsg; }
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override { iLastPulledMsg = EMsgStreamInterrupted; return aMsg; }
    Msg* ProcessMsg(MsgHalt* aMsg) override { iLastPulledMsg = EMsgHalt; return aMsg; }
    Msg* ProcessMsg(MsgFlush* aMsg) override { iLastPulledMsg = EMsgFlush; return aMsg; }
    Msg* ProcessMsg(MsgWait* aMsg) override { iLastPulledMsg = EMsgWait; return aMsg; }
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override { iLastPulledMsg = EMsgDecodedStream; iContinuity.NewStream(); return aMsg; }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override
    {   // ProcessAudio, :323-346
        iLastPulledMsg = EMsgAudioPcm;
        iJiffies += aMsg->Jiffies();
        const Media::Ramp& ramp = aMsg->Ramp();
        iContinuity.Audio(ramp);
        if (iRampingDown) {
            TEST(ramp.Direction() == Ramp::EDown);
            TEST(ramp.Start() == iLastRampPos);
            if (ramp.End() == Ramp::kMin) iRampingDown = false;
        }
        else if (iRampingUp) {
            TEST(ramp.Direction() == Ramp::EUp);
            TEST(ramp.Start() == iLastRampPos);
            if (ramp.End() == Ramp::kMax) iRampingUp = false;
        }
        else {
            TEST(ramp.Direction() == Ramp::ENone);
        }
        iLastRampPos = ramp.End();
        return aMsg;
    }
    Msg* ProcessMsg(MsgAudioDsd* aMsg) override { iLastPulledMsg = EMsgAudioDsd; return aMsg; }
    Msg* ProcessMsg(MsgSilence* aMsg) override { iLastPulledMsg = EMsgSilence; return aMsg; }
    Msg* ProcessMsg(MsgPlayable* aMsg) override { ASSERTS(); return aMsg; }
    Msg* ProcessMsg(MsgQuit* aMsg) override { iLastPulledMsg = EMsgQuit; return aMsg; }
private:
    void AddPending(Msg* aMsg)
    {
        { std::lock_guard<std::mutex> lock(iPendingMsgLock); iPendingMsgs.push_back(aMsg); }
        iMsgAvailable.Signal();
    }
    size_t PendingCount() { std::lock_guard<std::mutex> lock(iPendingMsgLock); return iPendingMsgs.size(); }
    void PullNext(TBool aWait = true)
    {
        if (aWait && !iRampingDown) {
            // no ramping => we expect a msg to be available: poll until the StarvationRamper has pulled something
            int retries = 1000;
            while (iStarvationRamper->IsEmpty() && retries-- > 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        Msg* msg = iStarvationRamper->Pull();
        msg = msg->Process(*this);
        msg->RemoveRef();
    }
    void PullNext(EMsgType aExpectedMsg, TBool aWait = true)
    {
        PullNext(aWait);
        if (iLastPulledMsg != aExpectedMsg) printf("  expected msg type %d, got %d\n", (int)aExpectedMsg, (int)iLastPulledMsg);
        TEST(iLastPulledMsg == aExpectedMsg);
    }
    Msg* CreateMode() { return iFactory.CreateMsgMode(ModeInfo(), "DummyMode"); }
    Msg* CreateDecodedStream()
    {
        DecodedStreamInfo info;
        info.iStreamId = iNextStreamId; info.iBitRate = 100; info.iBitDepth = iBitDepth; info.iSampleRate = iSampleRate;
        info.iNumChannels = kNumChannels; info.iCodecName = "notARealCodec"; info.iTrackLength = 1ull << 38; info.iStreamHandler = this;
        return iFactory.CreateMsgDecodedStream(info);
    }
    Msg* CreateAudio()
    {
        MsgAudioPcm* audio = iFactory.CreateMsgAudioPcm(Brn(iPcmData.data(), (TUint)iPcmData.size()), kNumChannels, iSampleRate, iBitDepth,
                                                        AudioDataEndian::Big, iTrackOffset);
        iTrackOffset += audio->Jiffies();
        return audio;
    }
    void Quit(TBool aRampDown = true)
    {
        iRampingDown = aRampDown;   // in case Pull() is called before StarvationRamper pulls the Halt below (causing SR to start a ramp down)
        AddPending(iFactory.CreateMsgHalt());
        AddPending(iFactory.CreateMsgQuit());
        do { PullNext(); } while (iLastPulledMsg != EMsgQuit);
    }
private:
    void TestMsgsPassWhenRunning()
    {   // :511-546 without its DSD leg
        AddPending(CreateMode());
        AddPending(iFactory.CreateMsgDelay(Jiffies::kPerMs * 20));
        AddPending(iFactory.CreateMsgDrain());
        AddPending(CreateDecodedStream());
        AddPending(CreateAudio());
        PullNext(EMsgMode); PullNext(EMsgDelay); PullNext(EMsgDrain); PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TUint size = Jiffies::kPerMs * 3;
        AddPending(iFactory.CreateMsgSilence(size, 44100, 8, 2));
        do { PullNext(EMsgSilence); } while (!iStarvationRamper->IsEmpty());
        AddPending(iFactory.CreateMsgHalt());
        AddPending(iFactory.CreateMsgQuit());
        PullNext(EMsgHalt);
        PullNext(EMsgQuit);
    }
    void TestBlocksWhenHasMaxAudio()
    {   // :548-578
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        do { AddPending(CreateAudio()); } while (iTrackOffset < kMaxAudioBuffer);
        AddPending(CreateAudio());
        int retries = 100;
        while (retries-- > 0) {                           // wait for expected number of pending msgs to be pulled
            if (PendingCount() == 1) break;                // 1 == the MsgAudioPcm that doesn't yet fit into SR
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));      // long enough for it to be pulled if SR were running
        TEST(PendingCount() == 1);
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);    // (the reference polls pending/IsEmpty, which has a window)
        TEST(PendingCount() == 0 && iStarvationRamper->IsEmpty());
        AddPending(iFactory.CreateMsgQuit());
        PullNext(EMsgQuit);
    }
    void TestNoRampAroundHalt()
    {   // :580-604
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        AddPending(CreateAudio());
        AddPending(CreateAudio());
        AddPending(iFactory.CreateMsgHalt());
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
        PullNext(EMsgHalt);
        AddPending(CreateAudio());
        AddPending(CreateAudio());
        AddPending(iFactory.CreateMsgQuit());
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
        PullNext(EMsgQuit);
    }
    void TestRampBeforeDrain()
    {   // :606-632
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        AddPending(CreateAudio());
        AddPending(CreateAudio());
        AddPending(iFactory.CreateMsgDrain());
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
        // ramp down then Halt should be generated before Drain is passed on
        iRampingDown = true;
        do { PullNext(EMsgAudioPcm); } while (iRampingDown);
        PullNext(EMsgHalt);
        PullNext(EMsgDrain);
        AddPending(iFactory.CreateMsgQuit());
        PullNext(EMsgQuit);
    }
    void TestRampsAroundStarvation()
    {   // :634-690
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        do { AddPending(CreateAudio()); } while (iTrackOffset < StarvationRamper::kTrainingJiffies);
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
        iRampingDown = true;
        iJiffies = 0;
        while (iRampingDown) PullNext(EMsgAudioPcm);
        TEST(iJiffies == StarvationRamper::kRampDownJiffies);
        PullNext(EMsgHalt, false);
        TEST(iStarvationRamper->CurrentState() == State::RampingUp);
        // ramps up once audio is available, ramp up takes kRampUpDuration
        iRampingUp = true;
        iJiffies = 0;
        const TUint64 trackOffsetStart = iTrackOffset;
        do { AddPending(CreateAudio()); } while (iTrackOffset - trackOffsetStart < kRampUpDuration);
        while (iRampingUp) PullNext(EMsgAudioPcm);
        TEST(iJiffies == kRampUpDuration);
        TEST(iStarvationRamper->CurrentState() == State::Running);
        if (!iStarvationRamper->IsEmpty()) PullNext(EMsgAudioPcm);          // clear any split msg at the end of the ramp up
        // ramps down after < kMaxAudioBuffer of prior audio, ramp down takes StarvationRamper::kRampDownJiffies
        AddPending(CreateDecodedStream());
        PullNext(EMsgDecodedStream);
        AddPending(CreateAudio());
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        iRampingDown = true;
        iJiffies = 0;
        while (iRampingDown) PullNext(EMsgAudioPcm);
        TEST(iJiffies == StarvationRamper::kRampDownJiffies);
        PullNext(EMsgHalt, false);
        TEST(iStarvationRamper->CurrentState() == State::RampingUp);
        Quit();
    }
    void TestNotifyStarvingAroundStarvation()
    {   // :692-724
        TEST(!iStarving);
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        TEST(!iStarving);
        AddPending(CreateAudio());
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        iRampingDown = true;
        PullNext(EMsgAudioPcm);
        TEST(iStarving && iStarvingStreamId == iNextStreamId);
        while (iRampingDown) PullNext(EMsgAudioPcm);
        TEST(iStarving);
        PullNext(EMsgHalt, false);
        iRampingUp = true;
        AddPending(CreateAudio());
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TEST(!iStarving);
        Quit();
    }
    void TestReportsBuffering()
    {   // :726-781
        TEST(iBuffering);
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        PullNext(EMsgMode);
        TEST(iBuffering);
        PullNext(EMsgDecodedStream);
        TEST(iBuffering);
        AddPending(CreateAudio());
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TEST(!iBuffering);
        iRampingDown = true;
        while (iRampingDown) { PullNext(EMsgAudioPcm); TEST(iBuffering); }
        PullNext(EMsgHalt, false);
        AddPending(CreateAudio());
        iRampingUp = true;
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TEST(!iBuffering);
        iRampingUp = false;
        iRampingDown = true;
        PullNext(EMsgAudioPcm);
        TEST(iBuffering);
        AddPending(CreateDecodedStream());
        do { PullNext(); } while (iLastPulledMsg != EMsgDecodedStream);
        iRampingDown = false;
        TEST(iBuffering);
        AddPending(CreateAudio());
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TEST(!iBuffering);
        AddPending(CreateDecodedStream());
        AddPending(CreateAudio());
        std::this_thread::sleep_for(std::chrono::milliseconds(50));       // short wait to allow StarvationRamper to pull the above msgs
        PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (!iStarvationRamper->IsEmpty());
        TEST(!iBuffering);
        Quit();
    }
    void TestFlush()
    {   // :783-809
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        for (TUint i = 0; i < 50; i++) AddPending(CreateAudio());
        const TUint kFlushId = 42;
        AddPending(iFactory.CreateMsgFlush(kFlushId));
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        PullNext(EMsgAudioPcm);
        iJiffies = 0;
        iStarvationRamper->Flush(kFlushId);
        TEST(iStarvationRamper->CurrentState() == State::RampingDown);
        iRampingDown = true;
        do { PullNext(EMsgAudioPcm); } while (iJiffies < StarvationRamper::kRampDownJiffies);
        TEST(iJiffies == StarvationRamper::kRampDownJiffies);
        TEST(iStarvationRamper->CurrentState() == State::Flushing);
        iRampingDown = false;
        PullNext(EMsgHalt);
        TEST(iStarvationRamper->IsEmpty());
        TEST(iStarvationRamper->CurrentState() == State::Halted);
        Quit(false);
    }
    void TestDrainAllAudio()
    {   // :811-858 without the DSD message
        AddPending(CreateMode());
        AddPending(CreateDecodedStream());
        do { AddPending(CreateAudio()); } while (iTrackOffset < StarvationRamper::kTrainingJiffies);
        PullNext(EMsgMode);
        PullNext(EMsgDecodedStream);
        do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
        AddPending(CreateAudio());
        AddPending(CreateDecodedStream());
        TUint size = Jiffies::kPerMs * 5;
        AddPending(iFactory.CreateMsgSilence(size, 44100, 8, 2));
        AddPending(iFactory.CreateMsgHalt());
        AddPending(iFactory.CreateMsgDrain());
        TEST(!iStarvationRamper->Draining());
        iJiffies = 0;
        iStarvationRamper->DrainAllAudio();
        TEST(!iStarvationRamper->Draining());
        TEST(iStarvationRamper->DrainRequested());
        iRampingDown = true;
        do {
            PullNext(EMsgAudioPcm);
            TEST(!iStarvationRamper->DrainRequested());
            TEST(iStarvationRamper->Draining());
        } while (iJiffies < StarvationRamper::kRampDownJiffies);
        TEST(iJiffies == StarvationRamper::kRampDownJiffies);
        iRampingDown = false;
        PullNext(EMsgHalt);
        TEST(iStarvationRamper->Draining());
        PullNext(EMsgDecodedStream);
        PullNext(EMsgHalt);
        TEST(!iStarvationRamper->DrainRequested() && iStarvationRamper->Draining());
        PullNext(EMsgDrain);
        TEST(!iStarvationRamper->DrainRequested() && !iStarvationRamper->Draining());
        Quit(false);
    }
    void TestAllSampleRates()
    {   // :860-914
        const TUint kSampleRates[] = { 7350, 8000, 11025, 12000, 14700, 16000, 22050, 24000, 29400, 32000, 44100, 48000, 88200, 96000, 176400, 192000 };
        const TUint kBitDepths[] = { 8, 16, 24, 32 };
        for (TUint bits : kBitDepths) {
            iBitDepth = bits;
            const TUint byteDepth = bits / 8;
            const TUint samples = DecodedAudio::kMaxBytes / (byteDepth * kNumChannels);
            iPcmData.assign((size_t)samples * byteDepth * kNumChannels, 0);
            for (TUint j = 0; j < samples; j++) for (TUint k = 0; k < byteDepth; k++) iPcmData[(size_t)j * byteDepth * 2 + k] = 0x7f;
            for (TUint rate : kSampleRates) {
                iSampleRate = rate;
                iTrackOffset = 0;
                iJiffies = 0;
                AddPending(CreateMode());
                AddPending(CreateDecodedStream());
                do { AddPending(CreateAudio()); } while (iTrackOffset < StarvationRamper::kTrainingJiffies);
                PullNext(EMsgMode);
                PullNext(EMsgDecodedStream);
                do { PullNext(EMsgAudioPcm); } while (iJiffies < iTrackOffset);
                iRampingDown = true;
                iJiffies = 0;
                while (iRampingDown) PullNext(EMsgAudioPcm);
                TUint expected = StarvationRamper::kRampDownJiffies;
                Jiffies::RoundDown(expected, iSampleRate);
                TEST(iJiffies == expected);
                PullNext(EMsgHalt, false);
            }
        }
        Quit();
    }
    void TestPruneMsgsNotReqdDownstream()
    {   // :916-930
        AddPending(iFactory.CreateMsgTrack());
        AddPending(iFactory.CreateMsgDelay(Jiffies::kPerMs * 20));
        AddPending(CreateDecodedStream());
        AddPending(iFactory.CreateMsgMetaText());
        AddPending(iFactory.CreateMsgWait());
        AddPending(iFactory.CreateMsgHalt());
        PullNext(EMsgDelay);
        PullNext(EMsgDecodedStream);
        PullNext(EMsgHalt);
        Quit(false);
    }
private:
    MsgFactory& iFactory;
    StarvationRamper* iStarvationRamper = nullptr;
    std::mutex iPendingMsgLock;
    Semaphore iMsgAvailable;
    std::deque<Msg*> iPendingMsgs;
    EMsgType iLastPulledMsg = ENone;
    TBool iRampingUp = false, iRampingDown = false, iBuffering = false, iStarving = false;
    TUint iStreamId = 0, iLastRampPos = 0, iNextStreamId = 1, iStarvingStreamId = 0, iSampleRate = 0, iBitDepth = 0;
    RampContinuity iContinuity;
    TUint64 iTrackOffset = 0, iJiffies = 0;
    std::vector<TByte> iPcmData;
};

} // namespace

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
    try {
        SuiteRampControl();
        {
            MsgFactory control(-1);
            SuiteMsgAudioControl(control);
            SuitePlayableControl(control);
            MsgAudioPcm* pcm = nullptr;
            TByte b[4] = { 1, 2, 3, 4 };
            pcm = control.CreateMsgAudioPcm(Brn(b, 4), 2, 44100, 16, AudioDataEndian::Little, 0);
            MsgPlayable* p = pcm->CreatePlayable();
            ProcessorPcmBufTest sink;
            TEST_THROWS(p->Read(sink), AssertionFailed);      // no GPU context: reading audio fails loudly, no CPU fallback
            p->RemoveRef();
            SuiteSongcastSenderControl(control);
            SuiteDecodedAudioAggregator aggregator(control);
            aggregator.Run();
            SuiteStarvationRamper starvation(control);
            starvation.RunControl();
            SuiteIdleLaneDoesNotStallTheTick(control);
            SuiteSrcStreamRing();
        }
        if (gpu) {
            MsgFactory f(0);
            SuitePlayableGpu(f);
            SuiteRamperGpu ramper(f);
            ramper.Run();
            SuiteSrcGpu src(f);
            src.Run();
            SuiteManyResampledLanesGpu manyResampled(f);
            manyResampled.Run();
            SuiteFlywheelGpu(f);
            SuiteStarvationRescueGpu(f);
            SuiteManyLanesStarveTogetherGpu(f);
            SuiteRescueBuffersPersistGpu(f);
            SuiteLaneLeftWithMetaTextIsRescuedGpu(f);
            SuiteSongcastSenderGpu(f);
            SuiteStarvationRamper starvation(f);
            starvation.RunControl();
            starvation.RunGpu();
        }
    }
    catch (const std::exception& e) {
        printf("UNEXPECTED EXCEPTION %s\n", e.what());
        gFailures++;
    }
    printf("%s: %d checks, %d failures\n", gpu ? "gpu" : "cpu", gChecks, gFailures);
    return gFailures == 0 ? 0 : 1;
}
