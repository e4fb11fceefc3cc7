// Sender.cpp -- see Sender.h.  File:line comments are relative to the reference tree (OpenHome/Av/Songcast/).
#include "Sender.h"

#include <algorithm>
#include <map>
#include <tuple>

#include "ohgpu.h"

using namespace OpenHome;
using namespace OpenHome::Av;
using namespace OpenHome::Media;

// ---------------------------------------------------------------------------------------------- OhmFrameBatch
OhmFrameBatch::OhmFrameBatch(MsgFactory& aFactory)
    : iFactory(aFactory)
{
}

OhmFrameBatch::~OhmFrameBatch()
{
    for (auto& f : iFrames) {
        for (auto* p : f.playables) {
            p->RemoveRef();
        }
    }
}

void OhmFrameBatch::Add(OhmFrameWork&& aWork)
{
    ASSERT(aWork.sink != nullptr);
    iFrames.push_back(std::move(aWork));
}

void OhmFrameBatch::Run()
{
    if (iFrames.empty()) {
        return;
    }
    ohgpu_ctx* ctx = iFactory.Gpu();
    typedef std::tuple<TUint, TUint, TUint, TUint, TUint64, std::string, TUint> StreamKey;
    std::map<StreamKey, TUint> streamIndex;
    std::vector<ohgpu_ohm_stream> streams;
    std::vector<ohgpu_ohm_frame_desc> frames;
    std::vector<ohgpu_ohm_fragment> fragments;
    std::vector<TByte> src;
    std::map<const DecodedAudio*, TUint64> audioBase;
    std::vector<TUint> frameBytes(iFrames.size());
    TUint64 dstBytes = 0;
    for (size_t i = 0; i < iFrames.size(); i++) {
        const OhmFrameWork& w = iFrames[i];
        // ---- the stream this frame belongs to ----
        TUint endian = OHGPU_ENDIAN_BIG;
        TBool endianKnown = false;
        for (auto* p : w.playables) {
            const PlayableWork& pw = p->Work();
            ASSERT(!pw.resampled);                       // a Sender downstream of the SampleRateConverter reads its output first (not wired up)
            if (!pw.silence && pw.frames > 0) {
                const TUint e = (pw.audio->Endian() == AudioDataEndian::Little) ? OHGPU_ENDIAN_LITTLE : OHGPU_ENDIAN_BIG;
                ASSERT(!endianKnown || e == endian);
                endian = e;
                endianKnown = true;
            }
        }
        const StreamKey key(w.sampleRate, w.bitRate, w.numChannels, w.bitDepth, w.samplesTotal, w.codecName, endian);
        auto it = streamIndex.find(key);
        if (it == streamIndex.end()) {
            ohgpu_ohm_stream s;
            memset(&s, 0, sizeof(s));
            s.samples_total = w.samplesTotal;
            s.sample_rate = w.sampleRate;
            s.bit_rate = w.bitRate;
            s.volume_offset = 0;                         // OhmSender.cpp:337
            s.src_channels = (uint8_t)w.numChannels;
            s.src_bits = (uint8_t)w.bitDepth;
            s.src_endian = (uint8_t)endian;
            ASSERT(w.codecName.size() <= OHGPU_OHM_MAX_CODEC_BYTES);
            s.codec_bytes = (uint8_t)w.codecName.size();
            memcpy(s.codec, w.codecName.data(), w.codecName.size());
            it = streamIndex.emplace(key, (TUint)streams.size()).first;
            streams.push_back(s);
        }
        // ---- the frame and its fragments ----
        ohgpu_ohm_frame_desc fr;
        memset(&fr, 0, sizeof(fr));
        fr.dst_offset = dstBytes;
        fr.sample_start = w.sampleStart;
        fr.stream = it->second;
        fr.frame = w.frame;
        fr.network_timestamp = 0;                        // no timestamper: OhmSender.cpp:443-454 leaves it 0 / not timestamped
        fr.media_latency = w.mediaLatency;
        fr.first_fragment = (uint32_t)fragments.size();
        fr.flags = (uint8_t)((w.halt ? OHGPU_OHM_FLAG_HALT : 0) | (w.lossless ? OHGPU_OHM_FLAG_LOSSLESS : 0));
        TUint samples = 0;
        for (auto* p : w.playables) {
            const PlayableWork& pw = p->Work();
            if (pw.frames == 0) {
                continue;
            }
            ASSERT(pw.channels == w.numChannels && pw.bitDepth == w.bitDepth);
            ohgpu_ohm_fragment g;
            memset(&g, 0, sizeof(g));
            g.n_frames = pw.frames;
            g.attenuation = (uint16_t)pw.attenuation;
            if (pw.silence) {
                g.flags = OHGPU_FLAG_SILENCE;
            }
            else {
                auto a = audioBase.find(pw.audio.get());
                if (a == audioBase.end()) {
                    a = audioBase.emplace(pw.audio.get(), (TUint64)src.size()).first;
                    src.insert(src.end(), pw.audio->Ptr(0), pw.audio->Ptr(0) + pw.audio->Bytes());
                }
                g.src_offset = a->second + pw.offsetBytes;
                if (pw.ramp.IsEnabled()) {
                    g.flags = OHGPU_FLAG_RAMP;
                    g.ramp_start = (uint16_t)pw.ramp.Start();
                    g.ramp_end = (uint16_t)pw.ramp.End();
                }
            }
            fragments.push_back(g);
            samples += pw.frames;
        }
        fr.n_fragments = (uint16_t)(fragments.size() - fr.first_fragment);
        uint32_t headerBytes = 0, bytes = 0;
        ASSERT(ohgpu_ohm_frame_layout(&streams[fr.stream], samples, &headerBytes, &bytes) == OHGPU_OK);   // ASSERT(BytesRemaining() >= ...), Sender.cpp:364
        frameBytes[i] = bytes;
        frames.push_back(fr);
        dstBytes += (bytes + 63u) & ~63u;
    }
    std::vector<TByte> dst((size_t)dstBytes);
    if (src.empty()) {
        src.push_back(0);
    }
    const int err = ohgpu_ohm_process_host(ctx, streams.data(), streams.size(), frames.data(), frames.size(),
                                           fragments.data(), fragments.size(), src.data(), src.size(), dst.data(), dst.size());
    ASSERT(err == OHGPU_OK);
    std::vector<OhmFrameWork> done;
    done.swap(iFrames);                                  // a sink may push more audio from inside Send
    for (size_t i = 0; i < done.size(); i++) {
        done[i].sink->Send(Brn(dst.data() + frames[i].dst_offset, frameBytes[i]));
        for (auto* p : done[i].playables) {
            p->RemoveRef();
        }
    }
}

// ---------------------------------------------------------------------------------------------- OhmSenderDriver
OhmSenderDriver::OhmSenderDriver(OhmFrameBatch& aBatch, IOhmDatagramSink& aSink)
    : iBatch(aBatch)
    , iSink(aSink)
    , iEnabled(false)
    , iActive(false)
    , iSend(false)
    , iFrame(0)
    , iSampleRate(0)
    , iBitRate(0)
    , iTimestampMultiplier(0)
    , iBytesPerSample(0)
    , iLossless(false)
    , iSamplesTotal(0)
    , iSampleStart(0)
    , iLatencyMs(0)
    , iLatencyOhm(0)
    , iFirstFrame(true)
{
}

void OhmSenderDriver::SetAudioFormat(TUint aSampleRate, TUint aBitRate, TUint aChannels, TUint aBitDepth, TBool aLossless,
                                     const Brx& aCodecName, TUint64 aSampleStart)
{
    iSampleRate = aSampleRate;
    iTimestampMultiplier = Jiffies::SongcastTicksPerSecond(aSampleRate);
    UpdateLatencyOhm();
    iBytesPerSample = aChannels * aBitDepth / 8;
    iLossless = aLossless;
    iSampleStart = aSampleStart;
    iBitRate = aBitRate;
    ASSERT(aCodecName.Bytes() <= OHGPU_OHM_MAX_CODEC_BYTES);         // Bws<kMaxCodecBytes> iCodec, OhmMsg.h:108
    iCodecName.assign((const char*)aCodecName.Ptr(), aCodecName.Bytes());
}

void OhmSenderDriver::SendAudio(std::vector<MsgPlayable*>& aPlayables, TUint aSamples, TUint aNumChannels, TUint aBitDepth, TBool aHalt)
{
    auto release = [&aPlayables]() {
        for (auto* p : aPlayables) {
            p->RemoveRef();
        }
        aPlayables.clear();
    };
    const TUint samples = (iBytesPerSample == 0) ? 0 : aSamples;     // OhmSender.cpp:422-428
    if (!iSend) {
        iSampleStart += samples;
        release();
        return;
    }
    if (iSampleRate == 0 || (samples == 0 && !aHalt)) {
        release();                                                    // nothing to usefully communicate to receivers
        return;
    }
    if (iFirstFrame) {
        iFirstFrame = false;
    }
    OhmFrameWork w;
    w.playables.swap(aPlayables);
    w.halt = aHalt;
    w.lossless = iLossless;
    w.frame = iFrame;
    w.mediaLatency = iLatencyOhm;
    w.sampleStart = iSampleStart;
    w.samplesTotal = iSamplesTotal;
    w.sampleRate = iSampleRate;
    w.bitRate = iBitRate;
    w.numChannels = aNumChannels;
    w.bitDepth = aBitDepth;
    w.codecName = iCodecName;
    w.sink = &iSink;
    iBatch.Add(std::move(w));
    iSampleStart += samples;
    iFrame++;
}

void OhmSenderDriver::StreamInterrupted()
{
    iFrame += 250;
}

void OhmSenderDriver::SetEnabled(TBool aValue)
{
    iEnabled = aValue;
    if (iSend) {
        if (!aValue) {
            ResetLocked();
        }
    }
    else if (aValue && iActive) {
        iSend = true;
    }
}

void OhmSenderDriver::SetActive(TBool aValue)
{
    iActive = aValue;
    if (iSend) {
        if (!aValue) {
            ResetLocked();
        }
    }
    else if (aValue && iEnabled) {
        iSend = true;
    }
}

void OhmSenderDriver::SetLatency(TUint aValue)
{
    iLatencyMs = aValue;
    UpdateLatencyOhm();
}

void OhmSenderDriver::SetTrackPosition(TUint64 aSamplesTotal, TUint64 aSampleStart)
{
    iSamplesTotal = aSamplesTotal;
    iSampleStart = aSampleStart;
}

void OhmSenderDriver::ResetLocked()
{
    iSend = false;
    iFrame = 0;
    iFirstFrame = true;
}

// ---------------------------------------------------------------------------------------------- Sender
namespace {

class PlayableCreator : private IMsgProcessor {          // Sender::PlayableCreator, Sender.cpp:401-522
public:
    MsgPlayable* Process(MsgAudio* aMsg) { iPlayable = nullptr; (void)aMsg->Process(*this); return iPlayable; }
private:
    Msg* ProcessMsg(MsgMode*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgTrack*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgDrain*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgDelay*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgEncodedStream*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgStreamSegment*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgAudioEncoded*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgMetaText*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgStreamInterrupted*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgHalt*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgFlush*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgWait*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgDecodedStream*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override { iPlayable = aMsg->CreatePlayable(); return nullptr; }
    Msg* ProcessMsg(MsgAudioDsd*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgSilence* aMsg) override { iPlayable = aMsg->CreatePlayable(); return nullptr; }
    Msg* ProcessMsg(MsgPlayable*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgQuit*) override { ASSERTS(); return nullptr; }
private:
    MsgPlayable* iPlayable = nullptr;
};

} // namespace

Sender::Sender(MsgFactory& aFactory, IOhmDatagramSink& aSink, TUint aMinLatencyMs, OhmFrameBatch* aSharedBatch)
    : iOwnBatch(aSharedBatch == nullptr ? new OhmFrameBatch(aFactory) : nullptr)
    , iBatch(aSharedBatch == nullptr ? *iOwnBatch : *aSharedBatch)
    , iDriver(iBatch, aSink)
    , iSampleRate(0)
    , iNumChannels(0)
    , iBitDepth(0)
    , iMinLatencyMs(aMinLatencyMs)
    , iStreamForbidden(false)
    , iFirstChannelIndex(0)
    , iBatchFrames(1)
{
    // what the OhmSender constructor, the "enabled" configuration value and the first listener do to the driver
    // (OhmSender.cpp:666-692, Sender.cpp:323-331, OhmSender.cpp:1004-1010)
    iDriver.SetLatency(aMinLatencyMs);
    iDriver.SetEnabled(true);
    iDriver.SetActive(true);
    iPendingAudio.reserve(100);
}

Sender::~Sender()
{
    for (auto* m : iPendingAudio) {
        m->RemoveRef();
    }
    delete iOwnBatch;
}

void Sender::Transmit()
{
    iBatch.Run();
}

void Sender::Push(Msg* aMsg)
{
    Msg* msg = aMsg->Process(*this);
    if (msg != nullptr) {
        msg->RemoveRef();
    }
}

Msg* Sender::ProcessMsg(MsgMode* aMsg)
{
    // the reference compares the mode's name with its own Songcast receiver mode (Sender.cpp:125-144); this mirror has no
    // receiver, so every mode may be sent
    aMsg->RemoveRef();
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgTrack* aMsg)
{
    SendPendingAudio();
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgDrain* aMsg)
{
    aMsg->RemoveRef();
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgDelay* aMsg)
{
    SendPendingAudio();
    iDriver.SetLatency(iMinLatencyMs);                   // std::max(latencyMs, iMinLatencyMs), Sender.cpp:163-164; the mirror's MsgDelay carries no delay
    aMsg->RemoveRef();
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgEncodedStream* aMsg) { ASSERTS(); return aMsg; }
Msg* Sender::ProcessMsg(MsgStreamSegment* aMsg) { ASSERTS(); return aMsg; }
Msg* Sender::ProcessMsg(MsgAudioEncoded* aMsg) { ASSERTS(); return aMsg; }

Msg* Sender::ProcessMsg(MsgMetaText* aMsg)
{
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgStreamInterrupted* aMsg)
{
    SendPendingAudio(true);
    iDriver.StreamInterrupted();
    aMsg->RemoveRef();
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgHalt* aMsg)
{
    SendPendingAudio(true);
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgFlush* aMsg)
{
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgWait* aMsg)
{
    SendPendingAudio(true);
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgDecodedStream* aMsg)
{
    // send any pending audio in case the stream msg indicates a discontinuity in the track
    SendPendingAudio();

    const DecodedStreamInfo& streamInfo = aMsg->StreamInfo();
    iSampleRate = streamInfo.SampleRate();
    iStreamForbidden = (streamInfo.Multiroom() == Multiroom::Forbidden);

    const TUint bitDepth = std::min(streamInfo.BitDepth(), (TUint)24);
    const TUint numChannels = streamInfo.NumChannels();
    const TUint64 samplesTotal = streamInfo.TrackLength() / Jiffies::PerSample(iSampleRate);
    iFirstChannelIndex = FirstChannelToSend(numChannels);
    iNumChannels = numChannels;
    iBitDepth = streamInfo.BitDepth();

    iDriver.SetTrackPosition(samplesTotal, streamInfo.SampleStart());
    if (!iStreamForbidden) {
        iDriver.SetAudioFormat(iSampleRate, streamInfo.BitRate(), std::min(numChannels, (TUint)2), bitDepth,
                               streamInfo.Lossless(), streamInfo.CodecName(), streamInfo.SampleStart());
    }
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgAudioPcm* aMsg)
{
    ASSERT(iSampleRate != 0);
    ProcessAudio(aMsg);
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgAudioDsd* aMsg)
{
    ASSERT(iStreamForbidden);
    aMsg->RemoveRef();
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgSilence* aMsg)
{
    ASSERT(iSampleRate != 0);
    ProcessAudio(aMsg);
    return nullptr;
}

Msg* Sender::ProcessMsg(MsgPlayable* aMsg)
{
    ASSERTS(); // don't expect this msg at this stage of the pipeline
    return aMsg;
}

Msg* Sender::ProcessMsg(MsgQuit* aMsg)
{
    SendPendingAudio(true);
    if (iOwnBatch != nullptr) {
        iBatch.Run();                                    // nothing may stay queued behind the last message
    }
    return aMsg;
}

void Sender::ProcessAudio(MsgAudio* aMsg)
{
    if (iStreamForbidden) {
        aMsg->RemoveRef();
        return;
    }
    TUint jiffies = 0;
    for (TUint i = 0; i < iPendingAudio.size(); i++) {
        jiffies += iPendingAudio[i]->Jiffies();
    }
    TUint newJiffies = jiffies + aMsg->Jiffies();
    if (newJiffies < kSongcastPacketJiffies) {
        iPendingAudio.push_back(aMsg);
        return;
    }
    MsgAudio* msg = aMsg;
    MsgAudio* remaining;
    do {
        remaining = (newJiffies == kSongcastPacketJiffies ? nullptr : msg->Split(kSongcastPacketJiffies - jiffies));
        iPendingAudio.push_back(msg);
        SendPendingAudio();
        msg = remaining;
        jiffies = 0;
        newJiffies = (remaining == nullptr ? 0 : remaining->Jiffies());
    } while (remaining != nullptr && newJiffies >= kSongcastPacketJiffies);
    if (remaining != nullptr) {
        iPendingAudio.push_back(remaining);
    }
}

void Sender::SendPendingAudio(TBool aHalt)
{
    PlayableCreator pc;
    std::vector<MsgPlayable*> playables;
    TUint samples = 0;
    for (TUint i = 0; i < iPendingAudio.size(); i++) {
        MsgPlayable* playable = pc.Process(iPendingAudio[i]);    // consumes the pending message's reference
        samples += playable->Work().frames;
        playables.push_back(playable);                            // read on the device, by OhmFrameBatch::Run
    }
    iPendingAudio.clear();
    iDriver.SendAudio(playables, samples, iNumChannels, iBitDepth, aHalt);
    if (iOwnBatch != nullptr && iBatchFrames != 0 && iBatch.Count() >= iBatchFrames) {
        iBatch.Run();
    }
}

TUint Sender::FirstChannelToSend(TUint aNumChannels)
{
    return (aNumChannels < 10) ? 0 : 8;
}
