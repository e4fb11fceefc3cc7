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
    iFrame += 250;                                       // a gap in the frame numbers is what makes receivers resync, OhmSender.cpp:482-488
}

// Frames go out while the sender is both enabled (configuration) and active (somebody listens).  Leaving that state
// restarts the frame numbering (OhmSender.cpp:490-525, 623-635).
void OhmSenderDriver::Gate(TBool aEnabled, TBool aActive)
{
    iEnabled = aEnabled;
    iActive = aActive;
    const TBool send = aEnabled && aActive;
    if (iSend && !send) {
        iFrame = 0;
    }
    iSend = send;
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

// ---------------------------------------------------------------------------------------------- Sender
// A packetiser: audio queues up until 5 ms are there, every full 5 ms leave as one frame; anything that marks a
// discontinuity (new track or stream, halt, wait, interruption, quit) sends what is queued first.  The reference reaches the
// same packets through a visitor with one hook per message type (Sender.cpp:125-321); here the message's kind selects a row
// of a small table and the queue keeps a running length.
Sender::Sender(MsgFactory& aFactory, IOhmDatagramSink& aSink, TUint aMinLatencyMs, OhmFrameBatch* aSharedBatch)
    : iOwnBatch(aSharedBatch == nullptr ? new OhmFrameBatch(aFactory) : nullptr)
    , iBatch(aSharedBatch == nullptr ? *iOwnBatch : *aSharedBatch)
    , iDriver(iBatch, aSink)
    , iQueuedJiffies(0)
    , iSampleRate(0)
    , iNumChannels(0)
    , iBitDepth(0)
    , iMinLatencyMs(aMinLatencyMs)
    , iStreamForbidden(false)
    , iBatchFrames(1)
{
    // what the OhmSender constructor, the "enabled" configuration value and the first listener do to the driver
    // (OhmSender.cpp:666-692, Sender.cpp:323-331, OhmSender.cpp:1004-1010)
    iDriver.SetLatency(aMinLatencyMs);
    iDriver.SetEnabled(true);
    iDriver.SetActive(true);
}

Sender::~Sender()
{
    for (MsgAudio* m : iQueued) {
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
    enum : TUint { Cut = 1, CutWithHalt = 2, Consume = 4, Never = 8 };     // what a kind does to the queue and to the message
    TUint what = 0;
    const MsgKind kind = KindOf(aMsg);
    switch (kind) {
    case MsgKind::Mode:              what = Consume; break;        // (:125-144 compares with the receiver's own mode; no receiver here)
    case MsgKind::Track:             what = Cut; break;            // :146-152
    case MsgKind::Drain:             what = Consume; break;
    case MsgKind::Delay:             what = Cut | Consume; break;  // :160-170
    case MsgKind::StreamInterrupted: what = CutWithHalt | Consume; break;   // :187-193
    case MsgKind::Halt:              what = CutWithHalt; break;    // :196-201
    case MsgKind::Wait:              what = CutWithHalt; break;
    case MsgKind::Quit:              what = CutWithHalt; break;
    case MsgKind::DecodedStream:     what = Cut; break;            // a new stream may be a discontinuity within the track
    case MsgKind::MetaText:
    case MsgKind::Flush:             break;
    case MsgKind::AudioPcm:
    case MsgKind::Silence:
        ASSERT(iSampleRate != 0);                                  // :246, :258
        Queue(static_cast<MsgAudio*>(aMsg));
        return;
    case MsgKind::AudioDsd:
        ASSERT(iStreamForbidden);                                  // :252
        what = Consume;
        break;
    default:                         what = Never; break;          // encoded audio and playables never get this far (:172-176, :264-268)
    }
    ASSERT(!(what & Never));
    if (what & (Cut | CutWithHalt)) {
        SendQueued((what & CutWithHalt) != 0);
    }
    if (kind == MsgKind::Delay) {
        iDriver.SetLatency(iMinLatencyMs);               // std::max(latencyMs, iMinLatencyMs), :163-164; this MsgDelay carries no delay
    }
    else if (kind == MsgKind::StreamInterrupted) {
        iDriver.StreamInterrupted();
    }
    else if (kind == MsgKind::DecodedStream) {
        NewStream(static_cast<MsgDecodedStream*>(aMsg)->StreamInfo());
    }
    else if (kind == MsgKind::Quit && iOwnBatch != nullptr) {
        iBatch.Run();                                    // nothing may stay queued behind the last message
    }
    aMsg->RemoveRef();                                   // the end of the line: nothing is passed on
}

void Sender::NewStream(const DecodedStreamInfo& aInfo)
{   // :203-230: the wire carries at most stereo, at most 24 bits
    iSampleRate = aInfo.SampleRate();
    iNumChannels = aInfo.NumChannels();
    iBitDepth = aInfo.BitDepth();
    iStreamForbidden = aInfo.Multiroom() == Multiroom::Forbidden;
    iDriver.SetTrackPosition(aInfo.TrackLength() / Jiffies::PerSample(iSampleRate), aInfo.SampleStart());
    if (!iStreamForbidden) {
        iDriver.SetAudioFormat(iSampleRate, aInfo.BitRate(), std::min(iNumChannels, (TUint)2), std::min(iBitDepth, (TUint)24),
                               aInfo.Lossless(), aInfo.CodecName(), aInfo.SampleStart());
    }
}

void Sender::Queue(MsgAudio* aAudio)
{   // :277-305: packets are cut at exactly 5 ms of queued audio, wherever the messages' own boundaries fall
    if (iStreamForbidden) {
        aAudio->RemoveRef();
        return;
    }
    while (aAudio != nullptr) {
        const TUint room = kSongcastPacketJiffies - iQueuedJiffies;
        const TUint length = aAudio->Jiffies();
        MsgAudio* rest = nullptr;
        if (length > room) {
            rest = aAudio->Split(room);
        }
        iQueued.push_back(aAudio);
        iQueuedJiffies += aAudio->Jiffies();
        if (length >= room) {
            SendQueued(false);
        }
        aAudio = rest;
    }
}

void Sender::SendQueued(TBool aHalt)
{   // :307-321 with PlayableCreator (:401-522) folded in: the queue only ever holds decoded audio or silence
    std::vector<MsgPlayable*> playables;
    TUint samples = 0;
    for (MsgAudio* m : iQueued) {
        MsgPlayable* playable = KindOf(m) == MsgKind::AudioPcm ? static_cast<MsgAudioPcm*>(m)->CreatePlayable()
                                                               : static_cast<MsgSilence*>(m)->CreatePlayable();
        samples += playable->Work().frames;
        playables.push_back(playable);                   // read on the device, by OhmFrameBatch::Run
    }
    iQueued.clear();
    iQueuedJiffies = 0;
    iDriver.SendAudio(playables, samples, iNumChannels, iBitDepth, aHalt);
    if (iOwnBatch != nullptr && iBatchFrames != 0 && iBatch.Count() >= iBatchFrames) {
        iBatch.Run();
    }
}

