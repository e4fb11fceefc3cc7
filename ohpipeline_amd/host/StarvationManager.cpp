// StarvationManager.cpp -- see StarvationManager.h.  File:line comments are relative to the reference tree
// (OpenHome/Media/Pipeline/StarvationRamper.cpp unless another file is named) and say which behaviour a line answers to.
#include "StarvationManager.h"

#include <algorithm>
#include <cstring>

#include "../../include/ohgpu.h"
#include "FlywheelRamper.h"

namespace OpenHome {
namespace Media {

// ------------------------------------------------------------------------------------------------ message kinds
namespace {
/** Collects what a request's playables deliver: packed big-endian audio at the stream's depth (IPcmProcessor's contract). */
class ByteCollector : public IPcmProcessor {
public:
    std::vector<TByte> iBytes;
    void BeginBlock() override {}
    void ProcessFragment(const Brx& aData, TUint, TUint) override { iBytes.insert(iBytes.end(), aData.Ptr(), aData.Ptr() + aData.Bytes()); }
    void ProcessSilence(const Brx& aData, TUint aCh, TUint aSb) override { ProcessFragment(aData, aCh, aSb); }
    void EndBlock() override {}
    void Flush() override {}
};
} // namespace

// ------------------------------------------------------------------------------------------------ RescueBatch
// Per request, what the reference does one stream at a time: FlywheelInput::Prepare reads the queued messages through their
// playables into planar 32-bit samples (:90-111, 159-186), FlywheelRamperManager::Ramp extrapolates 20 ms from the last
// millisecond (FlywheelRamper.cpp:45-131), RampGenerator cuts the 32-bit blocks back to the stream's depth and ramps them
// down from where the stream's ramp stood (:281-364).  Here: every request in the same four passes.
static std::atomic<TUint64> gFlywheelLaunches{0};
static std::atomic<TUint64> gRescueFailures{0};

TUint64 RescueBatch::Failures() { return gRescueFailures.load(); }

RescueArena::~RescueArena()
{
    for (void* b : iBuf) {
        if (b != nullptr) ohgpu_free(iCtx, b);
    }
}

void* RescueArena::Get(ohgpu_ctx* aCtx, TUint aWhich, size_t aBytes)
{
    ASSERT(aWhich < 4 && (iCtx == nullptr || iCtx == aCtx));
    iCtx = aCtx;
    if (aBytes > iCap[aWhich]) {                         // grow with headroom: the next, larger rescue should not allocate again
        if (iBuf[aWhich] != nullptr) ohgpu_free(aCtx, iBuf[aWhich]);
        iBuf[aWhich] = nullptr;
        iCap[aWhich] = 0;
        const size_t want = std::max<size_t>(2 * aBytes, 64 * 1024);
        void* fresh = nullptr;
        if (ohgpu_malloc(aCtx, want, &fresh) != OHGPU_OK) return nullptr;
        iBuf[aWhich] = fresh;
        iCap[aWhich] = want;
        iAllocations++;
    }
    return iBuf[aWhich];
}

TUint64 RescueBatch::FlywheelLaunches()
{
    return gFlywheelLaunches.load();
}

void RescueBatch::Run()
{
    if (iRequests.empty()) {
        return;
    }
    const size_t n = iRequests.size();
    // pass 1 (row a7 and friends): every message of every request becomes a playable, all read in one launch
    std::vector<ByteCollector> read(n);
    {
        PlayableBatch playables(iFactory);
        for (size_t i = 0; i < n; i++) {
            RescueRequest& rq = iRequests[i];
            ASSERT(rq.channels >= 1 && rq.channels <= 10);
            ASSERT(rq.bitDepth == 8 || rq.bitDepth == 16 || rq.bitDepth == 24 || rq.bitDepth == 32);
            while (!rq.audio.empty()) {
                MsgAudio* m = rq.audio.front();
                rq.audio.pop_front();
                MsgPlayable* p = nullptr;
                if (KindOf(m) == MsgKind::AudioPcm) {
                    p = static_cast<MsgAudioPcm*>(m)->CreatePlayable();
                }
                else {
                    p = static_cast<MsgSilence*>(m)->CreatePlayable();
                }
                playables.Add(p, read[i]);
            }
        }
        playables.Run();
    }
    // layout of the four arenas: newest training frames (packed) | planes (32-bit) | extrapolated blocks (32-bit) | packed output
    struct Place { size_t packedOff, planarOff, rampOff, outOff; TUint frames, inSamples, outFrames, blockFrames, outSub; size_t firstBlock, blocks; };
    std::vector<Place> place(n);
    std::vector<ohgpu_fmt_desc> unpack(n);
    std::vector<ohgpu_flywheel_desc> fly(n);
    std::vector<ohgpu_msg_desc> pack;
    std::vector<TByte> packedIn;
    size_t planarBytes = 0, rampBytes = 0, outBytes = 0;
    for (size_t i = 0; i < n; i++) {
        const RescueRequest& rq = iRequests[i];
        Place& pl = place[i];
        const TUint frameBytes = (rq.bitDepth / 8) * rq.channels;
        const TUint have = (TUint)(read[i].iBytes.size() / frameBytes);
        pl.frames = std::min(Jiffies::ToSamples(rq.jiffies, rq.sampleRate), have);           // the newest frames, :99-108
        pl.inSamples = Jiffies::ToSamples(kTrainingJiffies, rq.sampleRate);
        pl.outFrames = Jiffies::ToSamples(kRampDownJiffies, rq.sampleRate);
        pl.blockFrames = Jiffies::ToSamples(FlywheelRamperManager::kMaxOutputJiffiesBlockSize, rq.sampleRate);
        pl.outSub = rq.bitDepth / 8;
        ASSERT(pl.frames >= pl.inSamples && pl.frames > 0);                                    // InitChannels, FlywheelRamper.cpp:71
        pl.packedOff = packedIn.size();
        const TByte* newest = read[i].iBytes.data() + (size_t)(have - pl.frames) * frameBytes;
        packedIn.insert(packedIn.end(), newest, newest + (size_t)pl.frames * frameBytes);
        while (packedIn.size() % 16) packedIn.push_back(0);
        pl.planarOff = planarBytes;
        planarBytes += (((size_t)pl.frames * 4 * rq.channels) + 15) & ~(size_t)15;
        pl.rampOff = rampBytes;
        rampBytes += (((size_t)pl.outFrames * 4 * rq.channels) + 15) & ~(size_t)15;
        pl.outOff = outBytes;
        outBytes += (((size_t)pl.outFrames * pl.outSub * rq.channels) + 15) & ~(size_t)15;

        ohgpu_fmt_desc& u = unpack[i];
        memset(&u, 0, sizeof(u));
        u.kind = OHGPU_FMT_UNPACK_PLANAR;
        u.channels = (uint8_t)rq.channels;
        u.src_bits = (uint8_t)rq.bitDepth;
        u.n_frames = pl.frames;
        u.src_offset = pl.packedOff;
        u.dst_offset = pl.planarOff;
        u.dst_plane_stride = (uint64_t)pl.frames * 4;
        ohgpu_flywheel_desc& f = fly[i];
        memset(&f, 0, sizeof(f));
        f.src_offset = pl.planarOff;
        f.channel_bytes = (uint64_t)pl.frames * 4;
        f.dst_offset = pl.rampOff;
        f.in_samples = pl.inSamples;
        f.out_frames = pl.outFrames;
        f.block_frames = pl.blockFrames;
        f.sample_rate = rq.sampleRate;
        f.channels = rq.channels;
        pl.firstBlock = pack.size();
        for (TUint done = 0; done < pl.outFrames; done += pl.blockFrames) {                 // one 1 ms block = one message, :337-352
            ohgpu_msg_desc m;
            memset(&m, 0, sizeof(m));
            m.src_offset = pl.rampOff + (uint64_t)done * rq.channels * 4;
            m.dst_offset = pl.outOff + (uint64_t)done * rq.channels * pl.outSub;
            m.n_frames = std::min(pl.blockFrames, pl.outFrames - done);
            m.ramp_start = m.ramp_end = OHGPU_RAMP_MAX;
            m.attenuation = OHGPU_UNITY_ATTENUATION;
            m.channels = (uint8_t)rq.channels;
            m.src_bits = 32; m.src_endian = OHGPU_ENDIAN_BIG;
            m.dst_bits = (uint8_t)rq.bitDepth; m.dst_endian = OHGPU_ENDIAN_BIG;
            m.flags = rq.bitDepth == 32 ? OHGPU_FLAG_ZERO_LSB32 : 0;                          // "discard least significant byte", :311-320
            pack.push_back(m);
        }
        pl.blocks = pack.size() - pl.firstBlock;
    }
    // passes 2-4 on the context's stream, in order: a11, N1, a12
    ohgpu_ctx* ctx = iFactory.Gpu();
    std::vector<TByte> packedOut(outBytes);
    ohgpu_batch *ub = nullptr, *fb = nullptr, *pb = nullptr;
    void *dIn = nullptr, *dPlanar = nullptr, *dRamp = nullptr, *dOut = nullptr;
    int err = ohgpu_fmt_batch_create(ctx, unpack.data(), n, packedIn.size(), planarBytes, &ub);
    if (err == OHGPU_OK) err = ohgpu_flywheel_batch_create(ctx, fly.data(), n, planarBytes, rampBytes, &fb);
    if (err == OHGPU_OK && !pack.empty()) err = ohgpu_pcm_batch_create(ctx, pack.data(), pack.size(), rampBytes, outBytes, &pb);
    if (iArena != nullptr) {                             // the manager's persistent buffers
        const size_t need[4] = { packedIn.size(), planarBytes, std::max<size_t>(rampBytes, 16), std::max<size_t>(outBytes, 16) };
        void** into[4] = { &dIn, &dPlanar, &dRamp, &dOut };
        for (TUint k = 0; k < 4 && err == OHGPU_OK; k++) {
            *into[k] = iArena->Get(ctx, k, need[k]);
            if (*into[k] == nullptr) err = OHGPU_ERR_NOMEM;
        }
    }
    else {
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, packedIn.size(), &dIn);
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, planarBytes, &dPlanar);
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, std::max<size_t>(rampBytes, 16), &dRamp);
        if (err == OHGPU_OK) err = ohgpu_malloc(ctx, std::max<size_t>(outBytes, 16), &dOut);
    }
    if (err == OHGPU_OK) err = ohgpu_memcpy_h2d(ctx, dIn, packedIn.data(), packedIn.size(), nullptr);
    if (err == OHGPU_OK) err = ohgpu_fmt_batch_run(ctx, ub, dIn, dPlanar, nullptr);
    if (err == OHGPU_OK) { err = ohgpu_flywheel_batch_run(ctx, fb, dPlanar, dRamp, nullptr); gFlywheelLaunches++; }
    if (err == OHGPU_OK && pb) err = ohgpu_pcm_batch_run(ctx, pb, dRamp, dOut, nullptr);
    if (err == OHGPU_OK && outBytes) err = ohgpu_memcpy_d2h(ctx, packedOut.data(), dOut, outBytes, nullptr);
    if (err == OHGPU_OK) err = ohgpu_stream_sync(ctx, nullptr);
    if (iArena == nullptr) {
        for (void* d : {dIn, dPlanar, dRamp, dOut}) if (d) ohgpu_free(ctx, d);
    }
    for (ohgpu_batch* b : {ub, fb, pb}) if (b) ohgpu_batch_destroy(ctx, b);
    if (err != OHGPU_OK) {
        // The device let the rescue down (out of memory, a failed launch).  Nothing is thrown across the other lanes' period:
        // the starving lanes get no extrapolated audio, so their next message is the halt that would have followed it and they
        // ramp up from silence when audio returns -- an audible cut on those streams instead of a dead tick for all of them.
        gRescueFailures++;
        iRequests.clear();
        return;
    }
    // every block becomes a message that continues the stream's ramp downwards (a ramp already at its minimum: muted)
    for (size_t i = 0; i < n; i++) {
        const RescueRequest& rq = iRequests[i];
        const Place& pl = place[i];
        TUint current = rq.rampValue;
        TUint remaining = Jiffies::PerSample(rq.sampleRate) * pl.outFrames;                    // :241
        for (size_t k = 0; k < pl.blocks; k++) {
            const ohgpu_msg_desc& m = pack[pl.firstBlock + k];
            MsgAudioPcm* audio = iFactory.CreateMsgAudioPcm(Brn(packedOut.data() + m.dst_offset, m.n_frames * rq.channels * pl.outSub), rq.channels,
                                                            rq.sampleRate, rq.bitDepth, AudioDataEndian::Big, MsgAudioPcm::kTrackOffsetInvalid);
            if (current == Ramp::kMin) {
                audio->SetMuted();
            }
            else {
                MsgAudio* split = nullptr;
                current = audio->SetRamp(current, remaining, Ramp::EDown, split);
                ASSERT(split == nullptr);
            }
            rq.out->push_back(audio);
        }
    }
    iRequests.clear();
}

// ------------------------------------------------------------------------------------------------ one lane
struct StarvationManager::Lane {
    LaneConfig cfg;
    // ---- the inbox: filled by the feeder thread, emptied by the ticking thread; everything below `m` is guarded by it
    mutable std::mutex m;
    std::condition_variable room, arrival;
    std::deque<Msg*> inbox;
    TUint jiffies = 0, decodedStreams = 0, drains = 0, halts = 0, maxJiffies = 0;
    TBool quitSeen = false, closing = false;
    TUint gateJiffies = 0;                               // WaitForOccupancy: the next Pull waits once for this much audio
    LaneState state = LaneState::Halted;
    TUint rampValue = Ramp::kMin, rampRemaining = 0, flushTarget = MsgFlush::kIdInvalid;
    // ---- touched by the ticking thread only
    std::thread feeder;
    std::atomic<TBool> startDrain{false}, draining{false};
    std::deque<MsgAudio*> recent;                        // clones of what went out last, at least a training window of it
    TUint recentJiffies = 0;
    std::deque<Msg*> rescue;                             // extrapolated audio not yet handed out
    TBool starving = false, buffering = false, finished = false;   // finished: the quit has gone out, nothing follows it
    std::string mode;
    TUint streamId = IStreamHandler::kStreamIdInvalid, sampleRate = 0, bitDepth = 0, channels = 0;
    AudioFormat format = AudioFormat::Undefined;
    IStreamHandler* handler = nullptr;

    TBool Full() const { return jiffies >= maxJiffies || decodedStreams == cfg.maxStreamCount; }     // (m held)
    void Count(Msg* aMsg, MsgKind aKind, int aSign)                                                   // (m held)
    {
        if (aKind == MsgKind::AudioPcm || aKind == MsgKind::Silence) jiffies += aSign * static_cast<MsgAudio*>(aMsg)->Jiffies();
        else if (aKind == MsgKind::DecodedStream) decodedStreams += aSign;
        else if (aKind == MsgKind::Drain) drains += aSign;
        else if (aKind == MsgKind::Halt) halts += aSign;
    }
    void PushFront(Msg* aMsg)                            // something the consumer puts back: counted, no feeder-side effects
    {
        std::lock_guard<std::mutex> lock(m);
        Count(aMsg, KindOf(aMsg), +1);
        inbox.push_front(aMsg);
    }
    TBool CanStarve() const { return state == LaneState::Running || (state == LaneState::RampingUp && rampValue != Ramp::kMin); }
    void Feed();
};

void StarvationManager::Lane::Feed()
{   // the reference's puller (:469-489): pull, queue, park while the inbox is full, stop after a MsgQuit
    for (;;) {
        Msg* msg = cfg.upstream->Pull();
        const MsgKind kind = KindOf(msg);
        std::unique_lock<std::mutex> lock(m);
        Count(msg, kind, +1);
        if (kind == MsgKind::Delay) {                    // :712-716: the animator's delay sizes the reservoir, never below 140 ms
            maxJiffies = std::max(static_cast<MsgDelay*>(msg)->RemainingJiffies(), 140 * Jiffies::kPerMs);
        }
        else if (kind == MsgKind::Quit) {
            quitSeen = true;
        }
        inbox.push_back(msg);
        arrival.notify_all();
        if (quitSeen) {
            return;
        }
        room.wait(lock, [this] { return !Full() || closing; });
        if (closing) {
            return;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the manager
StarvationManager::StarvationManager(MsgFactory& aFactory)
    : iFactory(aFactory)
    , iRescueLaunches(0)
    , iDriverKnown(false)
{
}

void StarvationManager::ClaimDriverThread()
{
    std::lock_guard<std::mutex> lock(iDriverLock);
    if (!iDriverKnown) {
        iDriver = std::this_thread::get_id();
        iDriverKnown = true;
    }
    ASSERT(iDriver == std::this_thread::get_id());       // one driver thread per manager: the rescues share their device buffers
}

TUint64 StarvationManager::DeviceAllocations() const
{
    uint64_t n = 0;
    ohgpu_ctx* ctx = iFactory.Gpu();
    if (ctx != nullptr) {
        const int err = ohgpu_device_allocations(ctx, &n);
        ASSERT(err == OHGPU_OK);
    }
    return n;
}

StarvationManager::~StarvationManager()
{
    for (auto& lp : iLanes) {
        Lane& lane = *lp;
        {
            std::lock_guard<std::mutex> lock(lane.m);
            lane.closing = true;
        }
        lane.room.notify_all();
        if (lane.feeder.joinable()) {
            lane.feeder.join();
        }
        for (Msg* m : lane.inbox) m->RemoveRef();
        for (Msg* m : lane.rescue) m->RemoveRef();
        for (MsgAudio* m : lane.recent) m->RemoveRef();
    }
}

TUint StarvationManager::AddLane(const LaneConfig& aConfig)
{
    ASSERT(aConfig.upstream != nullptr && aConfig.observer != nullptr);
    iLanes.emplace_back(new Lane());
    Lane& lane = *iLanes.back();
    lane.cfg = aConfig;
    lane.maxJiffies = aConfig.sizeJiffies;
    SetBuffering(lane, true);
    lane.feeder = std::thread(&Lane::Feed, &lane);
    return (TUint)iLanes.size() - 1;
}

void StarvationManager::Flush(TUint aLane, TUint aId)
{   // :422-429
    Lane& lane = *iLanes.at(aLane);
    std::lock_guard<std::mutex> lock(lane.m);
    lane.flushTarget = aId;
    lane.rampValue = Ramp::kMax;
    lane.rampRemaining = kRampDownJiffies;
    lane.state = LaneState::RampingDown;
}

void StarvationManager::DrainAllAudio(TUint aLane)
{
    iLanes.at(aLane)->startDrain.store(true);
}

void StarvationManager::WaitForOccupancy(TUint aLane, TUint aJiffies)
{   // :906-918
    Lane& lane = *iLanes.at(aLane);
    std::lock_guard<std::mutex> lock(lane.m);
    if (lane.drains == 0 && lane.halts == 0) {
        lane.gateJiffies = aJiffies;
    }
}

LaneState StarvationManager::State(TUint aLane) const
{
    const Lane& lane = *iLanes.at(aLane);
    std::lock_guard<std::mutex> lock(lane.m);
    return lane.state;
}

TBool StarvationManager::IsEmpty(TUint aLane) const
{
    const Lane& lane = *iLanes.at(aLane);
    std::lock_guard<std::mutex> lock(lane.m);
    return lane.inbox.empty();
}

TUint StarvationManager::SizeInJiffies(TUint aLane) const
{
    const Lane& lane = *iLanes.at(aLane);
    std::lock_guard<std::mutex> lock(lane.m);
    return lane.jiffies;
}

TBool StarvationManager::Draining(TUint aLane) const
{
    return iLanes.at(aLane)->draining.load();
}

TBool StarvationManager::Finished(TUint aLane) const
{
    return iLanes.at(aLane)->finished;
}

TBool StarvationManager::DrainRequested(TUint aLane) const
{
    return iLanes.at(aLane)->startDrain.load();
}

void StarvationManager::SetBuffering(Lane& aLane, TBool aBuffering)
{   // :589-604, the observer thread replaced by a direct call
    if (aLane.buffering != aBuffering) {
        aLane.buffering = aBuffering;
        aLane.cfg.observer->NotifyStarvationRamperBuffering(aBuffering);
    }
}

void StarvationManager::NewStream(Lane& aLane)
{
    {
        std::lock_guard<std::mutex> lock(aLane.m);
        aLane.state = LaneState::Starting;
    }
    for (MsgAudio* m : aLane.recent) m->RemoveRef();
    aLane.recent.clear();
    aLane.recentJiffies = 0;
    aLane.streamId = IStreamHandler::kStreamIdInvalid;
}

void StarvationManager::RememberAudio(Lane& aLane, MsgAudio* aMsg)
{   // :529-559: the stream is audible again; keep a clone so that the last millisecond is at hand when it runs dry
    if (aLane.starving) {
        aLane.starving = false;
        if (aLane.handler != nullptr) {
            aLane.handler->NotifyStarving(Brn((const TByte*)aLane.mode.data(), (TUint)aLane.mode.size()), aLane.streamId, false);
        }
    }
    if (aLane.format == AudioFormat::Dsd) {
        return;
    }
    MsgAudio* copy = aMsg->Clone();
    aLane.recent.push_back(copy);
    aLane.recentJiffies += copy->Jiffies();
    if (aLane.recentJiffies > kTrainingJiffies && aLane.recent.size() > 1) {
        MsgAudio* oldest = aLane.recent.front();         // droppable only if a full window stays without it
        if (aLane.recentJiffies - oldest->Jiffies() >= kTrainingJiffies) {
            aLane.recent.pop_front();
            aLane.recentJiffies -= oldest->Jiffies();
            oldest->RemoveRef();
        }
    }
}

void StarvationManager::QueueRescue(Lane& aLane, RescueBatch& aBatch)
{   // :491-537: exactly one training window -- the excess cut off the front, a shortfall made up with silence in front
    while (aLane.recentJiffies > kTrainingJiffies) {
        const TUint excess = aLane.recentJiffies - kTrainingJiffies;
        MsgAudio* oldest = aLane.recent.front();
        aLane.recent.pop_front();
        if (oldest->Jiffies() > excess) {
            aLane.recent.push_front(oldest->Split(excess));   // the part that stays
        }
        aLane.recentJiffies -= oldest->Jiffies();
        oldest->RemoveRef();
    }
    while (aLane.recentJiffies < kTrainingJiffies) {
        TUint size = std::min(kTrainingJiffies - aLane.recentJiffies, (TUint)kMaxAudioOutJiffies);
        MsgSilence* pad = iFactory.CreateMsgSilence(size, aLane.sampleRate, aLane.bitDepth, aLane.channels);   // (size comes back rounded to whole samples)
        aLane.recent.push_front(pad);
        aLane.recentJiffies += pad->Jiffies();
    }
    RescueRequest rq;
    rq.audio.swap(aLane.recent);
    rq.jiffies = aLane.recentJiffies;
    rq.sampleRate = aLane.sampleRate; rq.bitDepth = aLane.bitDepth; rq.channels = aLane.channels;
    rq.out = &aLane.rescue;
    aLane.recentJiffies = 0;
    {
        std::lock_guard<std::mutex> lock(aLane.m);
        rq.rampValue = aLane.rampValue;
        aLane.state = LaneState::FlywheelRamping;
    }
    aBatch.Add(std::move(rq));
    aLane.starving = true;
    if (aLane.handler != nullptr) {
        aLane.handler->NotifyStarving(Brn((const TByte*)aLane.mode.data(), (TUint)aLane.mode.size()), aLane.streamId, true);
    }
}

void StarvationManager::RescueNow(Lane& aLane)
{
    RescueBatch one(iFactory, &iArena);
    QueueRescue(aLane, one);
    one.Run();
    iRescueLaunches++;
}

TBool StarvationManager::Prepare(Lane& aLane, RescueBatch& aBatch, TBool aMayBlock)
{   // :622-646.  Returns false when the lane sits this period out (its occupancy gate is still shut and the caller may not wait).
    if (aLane.finished) {
        return true;
    }
    TBool dry;
    {
        std::unique_lock<std::mutex> lock(aLane.m);
        if (aLane.gateJiffies > 0 && aLane.drains == 0 && aLane.halts == 0) {
            auto open = [&aLane] { return aLane.jiffies >= aLane.gateJiffies || aLane.drains > 0 || aLane.halts > 0; };
            if (aMayBlock) {
                aLane.arrival.wait(lock, open);
            }
            else if (!open()) {
                return false;                                // one gated stream does not hold the other lanes' period up
            }
            aLane.gateJiffies = 0;
        }
        dry = aLane.inbox.empty();
    }
    if (dry || aLane.startDrain.load()) {
        SetBuffering(aLane, true);
        if (aLane.startDrain.exchange(false)) {
            aLane.draining.store(true);
        }
        TBool rescue;
        {
            std::lock_guard<std::mutex> lock(aLane.m);
            rescue = aLane.CanStarve() && !aLane.quitSeen;
        }
        if (rescue) {
            QueueRescue(aLane, aBatch);
        }
    }
    return true;
}

Msg* StarvationManager::Next(Lane& aLane, TBool aMayBlock)
{   // :648-673.  aMayBlock == false (Tick): a lane with nothing to hand over -- halted, starting or flushing with an empty
    // inbox; a lane that could still play would have been rescued by Prepare -- returns nullptr instead of waiting for its feeder.
    if (aLane.finished) {
        return nullptr;
    }
    for (;;) {
        if (!aLane.rescue.empty()) {
            Msg* audio = aLane.rescue.front();
            aLane.rescue.pop_front();
            return audio;
        }
        TBool wasFlushing;
        Msg* msg;
        {
            std::unique_lock<std::mutex> lock(aLane.m);
            if (aLane.state == LaneState::FlywheelRamping) { // the extrapolated audio has gone out: halt, then ramp up from silence
                aLane.state = LaneState::RampingUp;
                aLane.rampValue = Ramp::kMin;
                aLane.rampRemaining = aLane.cfg.rampUpJiffies;
                return iFactory.CreateMsgHalt();
            }
            wasFlushing = aLane.state == LaneState::Flushing;
            if (!aMayBlock && aLane.inbox.empty()) {
                return nullptr;                              // no message this period
            }
            aLane.arrival.wait(lock, [&aLane] { return !aLane.inbox.empty(); });
            msg = aLane.inbox.front();
            aLane.inbox.pop_front();
            aLane.Count(msg, KindOf(msg), -1);
            if (!aLane.Full()) {
                aLane.room.notify_all();
            }
        }
        msg = Handle(aLane, msg);
        if (msg != nullptr && wasFlushing) {
            TBool still;
            {
                std::lock_guard<std::mutex> lock(aLane.m);
                still = aLane.state == LaneState::Flushing;
            }
            if (still) {                                 // between the ramp down and the awaited flush everything is discarded
                msg->RemoveRef();
                msg = nullptr;
            }
        }
        if (msg != nullptr) {
            return msg;
        }
        if (!aMayBlock) {
            // The message was one Handle() consumes (Track, MetaText, Wait, discarded flush content) and it may have been the inbox's
            // last: Prepare saw a non-empty inbox and did not rescue, the reference would block in DoDequeue here, and a lane that IS
            // playing must not sit a period out unramped -- rescue it now, the flywheel audio goes out this period.
            TBool rescue;
            {
                std::lock_guard<std::mutex> lock(aLane.m);
                rescue = aLane.inbox.empty() && aLane.CanStarve() && !aLane.quitSeen;
            }
            if (rescue) {
                SetBuffering(aLane, true);
                RescueNow(aLane);
            }
        }
    }
}

Msg* StarvationManager::Handle(Lane& aLane, Msg* aMsg)
{
    const MsgKind kind = KindOf(aMsg);
    switch (kind) {
    case MsgKind::Mode:
        NewStream(aLane);
        aLane.mode = static_cast<MsgMode*>(aMsg)->Mode();
        return aMsg;
    case MsgKind::Track:                                 // not wanted downstream (:718-724)
        NewStream(aLane);
        aMsg->RemoveRef();
        return nullptr;
    case MsgKind::MetaText:
    case MsgKind::Wait:
        aMsg->RemoveRef();
        return nullptr;
    case MsgKind::Drain: {                               // :726-737: audio still playing is ramped out before the drain passes
        aLane.draining.store(false);
        TBool audible;
        {
            std::lock_guard<std::mutex> lock(aLane.m);
            audible = aLane.CanStarve();
        }
        if (audible) {
            aLane.PushFront(aMsg);
            SetBuffering(aLane, true);
            RescueNow(aLane);
            return nullptr;
        }
        return aMsg;
    }
    case MsgKind::Halt: {                                // :743-751
        std::lock_guard<std::mutex> lock(aLane.m);
        aLane.state = LaneState::Halted;
        return aMsg;
    }
    case MsgKind::Flush: {                               // :753-770
        const TUint id = static_cast<MsgFlush*>(aMsg)->Id();
        aMsg->RemoveRef();
        LaneState st;
        TBool awaited;
        {
            std::lock_guard<std::mutex> lock(aLane.m);
            st = aLane.state;
            awaited = aLane.flushTarget != MsgFlush::kIdInvalid && id == aLane.flushTarget;
        }
        if (awaited && st == LaneState::RampingDown) {   // the flush arrived before the ramp down ended: extrapolate the rest
            RescueNow(aLane);
        }
        else if (awaited && st == LaneState::Flushing) {
            std::lock_guard<std::mutex> lock(aLane.m);
            aLane.state = LaneState::Halted;
            aLane.flushTarget = MsgFlush::kIdInvalid;
            return iFactory.CreateMsgHalt();
        }
        return nullptr;
    }
    case MsgKind::DecodedStream: {                       // :778-790
        NewStream(aLane);
        const DecodedStreamInfo& info = static_cast<MsgDecodedStream*>(aMsg)->StreamInfo();
        aLane.streamId = info.StreamId();
        aLane.handler = info.StreamHandler();
        aLane.sampleRate = info.SampleRate();
        aLane.bitDepth = info.BitDepth();
        aLane.channels = info.NumChannels();
        aLane.format = info.Format();
        std::lock_guard<std::mutex> lock(aLane.m);
        aLane.rampValue = Ramp::kMax;
        return aMsg;
    }
    case MsgKind::AudioPcm: {                            // :792-834
        if (aLane.draining.load()) {
            aMsg->RemoveRef();
            return nullptr;
        }
        MsgAudioPcm* audio = static_cast<MsgAudioPcm*>(aMsg);
        if (audio->Jiffies() > kMaxAudioOutJiffies) {    // the driver gets at most 5 ms at a time
            aLane.PushFront(audio->Split(kMaxAudioOutJiffies));
        }
        std::unique_lock<std::mutex> lock(aLane.m);
        if (aLane.state == LaneState::Starting || aLane.state == LaneState::Halted) {
            aLane.state = LaneState::Running;
        }
        const TBool ramping = (aLane.state == LaneState::RampingUp || aLane.state == LaneState::RampingDown) && aLane.rampRemaining > 0;
        if (ramping) {
            const Ramp::EDirection dir = aLane.state == LaneState::RampingUp ? Ramp::EUp : Ramp::EDown;
            lock.unlock();
            if (audio->Jiffies() > aLane.rampRemaining) {
                aLane.PushFront(audio->Split(aLane.rampRemaining));
            }
            MsgAudio* tail = nullptr;
            const TUint value = audio->SetRamp(aLane.rampValue, aLane.rampRemaining, dir, tail);
            if (tail != nullptr) {
                aLane.PushFront(tail);
            }
            lock.lock();
            aLane.rampValue = value;
            if (aLane.rampRemaining == 0) {
                aLane.state = dir == Ramp::EUp ? LaneState::Running : LaneState::Flushing;
            }
        }
        lock.unlock();
        RememberAudio(aLane, audio);
        SetBuffering(aLane, false);
        return audio;
    }
    case MsgKind::Silence: {                             // :836-852
        if (aLane.draining.load()) {
            aMsg->RemoveRef();
            return nullptr;
        }
        MsgSilence* silence = static_cast<MsgSilence*>(aMsg);
        {
            std::lock_guard<std::mutex> lock(aLane.m);
            if (aLane.state == LaneState::Halted) {
                aLane.state = LaneState::Starting;
            }
        }
        if (silence->Jiffies() > kMaxAudioOutJiffies) {
            aLane.PushFront(silence->Split(kMaxAudioOutJiffies));
        }
        RememberAudio(aLane, silence);
        return silence;
    }
    case MsgKind::Quit:
        aLane.finished = true;
        return aMsg;
    default:
        return aMsg;                                     // delay, DSD, ...: straight through
    }
}

void StarvationManager::Tick(std::vector<Msg*>& aOut)
{
    // Nothing in a tick waits for a feeder: the reference gives every pipeline a StarvationRamper and a driver thread of its
    // own, so an idle pipeline blocks nobody else; here the lanes share the tick, and one that has nothing to say this period
    // (halted or not yet started with an empty inbox, or held at its occupancy gate) just says nothing -- nullptr.
    ClaimDriverThread();
    RescueBatch batch(iFactory, &iArena);
    std::vector<TBool> takesPart(iLanes.size());
    for (size_t i = 0; i < iLanes.size(); i++) {
        takesPart[i] = Prepare(*iLanes[i], batch, false);
    }
    if (batch.Count() > 0) {
        batch.Run();                                     // every lane that ran dry in this tick, together
        iRescueLaunches++;
    }
    aOut.clear();
    for (size_t i = 0; i < iLanes.size(); i++) {
        aOut.push_back(takesPart[i] ? Next(*iLanes[i], false) : nullptr);
    }
}

Msg* StarvationManager::Pull(TUint aLane)
{
    ClaimDriverThread();
    Lane& lane = *iLanes.at(aLane);
    RescueBatch batch(iFactory, &iArena);
    Prepare(lane, batch, true);                          // (one lane, its own caller: blocks as the reference's Pull does)
    if (batch.Count() > 0) {
        batch.Run();
        iRescueLaunches++;
    }
    return Next(lane, true);
}

} // namespace Media
} // namespace OpenHome
