// StarvationManager.h -- what happens to MANY streams when their audio runs out, designed for the device.
//
// The reference gives every pipeline one StarvationRamper (OpenHome/Media/Pipeline/StarvationRamper.cpp): a reservoir fed
// by a puller thread; when the driver finds it empty the last millisecond of audio is extrapolated by FlywheelRamper
// (:491-537 -> FlywheelRamper.cpp:45-66), handed out as 20 ms of down-ramped audio and followed by a MsgHalt; the next audio
// ramps up over aRampUpSize (:791-832); Flush(id) ramps down and discards to the flush, DrainAllAudio() forces the ramp and
// discards to the next MsgDrain (:622-673).  One stream at a time that is a few thousand integer operations on its own thread.
//
// On a GPU the unit of work is the DRIVER TICK: every period each stream ("lane") owes the driver one message, and every lane
// that ran dry in that tick is rescued in the SAME device passes -- one read of the lanes' last milliseconds (row a7), one
// unpack to planes (a11), one flywheel launch (N1), one pack (a12).  So the manager is a table of lanes advanced per tick:
//     Tick():  1. look at every lane: gate on occupancy, notice an empty inbox or a pending drain, collect who must be rescued
//              2. ONE RescueBatch for all of them
//              3. every lane hands over its next message (rescue audio first, then the halt, then whatever the inbox holds)
// A lane's behaviour towards ITS driver is the reference's, message for message (the reference's own suite, restated in
// tests/cpp/test_host.cpp, runs against the one-lane facade StarvationRamper).  No PCM byte is touched on the host.
// Left out: DSD (MsgAudioDsd passes through unramped), the observer thread (observers are called synchronously, as the
// reference's tests do with ElementObserverSync), thread priorities.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "Msg.h"

namespace OpenHome {
namespace Media {

class IStarvationRamperObserver {                        // StarvationRamper.h:22-27
public:
    virtual ~IStarvationRamperObserver() {}
    virtual void NotifyStarvationRamperBuffering(TBool aBuffering) = 0;
};

enum class LaneState { Starting, Running, Halted, RampingUp, FlywheelRamping, RampingDown, Flushing };

/** One stream's last millisecond, as the rescue wants it: the messages that hold it (references owned), oldest first. */
struct RescueRequest {
    std::deque<MsgAudio*> audio;
    TUint jiffies = 0, sampleRate = 0, bitDepth = 0, channels = 0;
    TUint rampValue = 0;                                 // where the stream's ramp stood
    std::deque<Msg*>* out = nullptr;                     // receives the 1 ms messages of extrapolated, down-ramped audio
};

/** The four device buffers a rescue works in (newest frames, planes, extrapolated blocks, packed output), kept from rescue to
 *  rescue and grown on demand: a rescue happens in the very period in which its lanes starve, and four allocations and frees
 *  in it were four too many (the advisor's finding).  One per StarvationManager; a RescueBatch without one allocates per run. */
class RescueArena {
public:
    RescueArena() {}
    ~RescueArena();
    RescueArena(const RescueArena&) = delete;
    RescueArena& operator=(const RescueArena&) = delete;
    /** Buffer `aWhich` (0..3) of at least aBytes on aCtx's device, or nullptr when the device refuses. */
    void* Get(ohgpu_ctx* aCtx, TUint aWhich, size_t aBytes);
    TUint64 Allocations() const { return iAllocations; }             // device allocations so far (a steady state makes none)
private:
    ohgpu_ctx* iCtx = nullptr;
    void* iBuf[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t iCap[4] = { 0, 0, 0, 0 };
    TUint64 iAllocations = 0;
};

/** Extrapolates every request's audio in one chain of device passes (a7 read, a11, N1, a12) and queues the messages. */
class RescueBatch {
public:
    static const TUint kTrainingJiffies = Jiffies::kPerMs * 1;       // StarvationRamper.cpp:374-376
    static const TUint kRampDownJiffies = Jiffies::kPerMs * 20;
public:
    explicit RescueBatch(MsgFactory& aFactory, RescueArena* aArena = nullptr) : iFactory(aFactory), iArena(aArena) {}
    void Add(RescueRequest&& aRequest) { iRequests.push_back(std::move(aRequest)); }
    TUint Count() const { return (TUint)iRequests.size(); }
    void Run();
    /** Extrapolation launches (ohgpu_flywheel_batch_run calls) by every batch of this process so far. */
    static TUint64 FlywheelLaunches();
    /** Rescues the device failed (their lanes went straight to the halt: RescueBatch::Run never throws for a device error). */
    static TUint64 Failures();
private:
    MsgFactory& iFactory;
    RescueArena* iArena;                                  // whose buffers to work in (nullptr: allocated and freed by Run)
    std::vector<RescueRequest> iRequests;
};

class StarvationManager {
public:
    static const TUint kTrainingJiffies = RescueBatch::kTrainingJiffies;
    static const TUint kRampDownJiffies = RescueBatch::kRampDownJiffies;
    static const TUint kMaxAudioOutJiffies = Jiffies::kPerMs * 5;
    struct LaneConfig {
        IPipelineElementUpstream* upstream = nullptr;    // pulled by the lane's feeder thread until the inbox holds sizeJiffies
        IStarvationRamperObserver* observer = nullptr;
        TUint sizeJiffies = 0, rampUpJiffies = 0, maxStreamCount = 0;
    };
public:
    explicit StarvationManager(MsgFactory& aFactory);
    /** A lane's feeder ends once it has passed a MsgQuit on, as the reference's puller does: send one before destroying. */
    ~StarvationManager();
    TUint AddLane(const LaneConfig& aConfig);
    TUint LaneCount() const { return (TUint)iLanes.size(); }
    /** THREADS: Tick(), Pull() and everything they call belong to ONE driver thread per StarvationManager (the reference has one
     *  driver thread per pipeline; a manager is many pipelines' worth of lanes behind one driver).  The rescues of all lanes share
     *  one set of device buffers (RescueArena) and one context: two threads pulling different lanes would hand each other's
     *  buffers out.  The first call names the driver thread; a call from another one ASSERTs.  Feeders, Flush(), DrainAllAudio(),
     *  WaitForOccupancy() and the inspection calls may come from anywhere. */
    /** One driver period for every lane: aOut[i] is lane i's next message, or nullptr when the lane has none this period --
     *  its inbox is empty and it is not playing (halted, starting, flushing: a lane that IS playing is rescued instead), it is
     *  held at its occupancy gate, or its MsgQuit has already gone out (Finished()).  Never waits for a feeder. */
    void Tick(std::vector<Msg*>& aOut);
    /** One driver period for one lane (what IPipelineElementUpstream::Pull is to the reference). */
    Msg* Pull(TUint aLane);
    void Flush(TUint aLane, TUint aId);                  // ramps down quickly, then discards everything up to the flush with this id
    void DrainAllAudio(TUint aLane);                     // discards buffered audio, forcing a rescue, until the next MsgDrain
    void WaitForOccupancy(TUint aLane, TUint aJiffies);  // the lane's next Pull blocks once until this level is reached
public: // inspection (the reference's suite reads these as a friend)
    LaneState State(TUint aLane) const;
    TBool IsEmpty(TUint aLane) const;
    TUint SizeInJiffies(TUint aLane) const;
    TBool Draining(TUint aLane) const;
    TBool Finished(TUint aLane) const;                   // the lane's MsgQuit has gone out
    TBool DrainRequested(TUint aLane) const;             // DrainAllAudio() called, not yet seen by a tick
    TUint64 RescueLaunches() const { return iRescueLaunches.load(); }   // device rescues so far (one per tick that needed any)
    TUint64 RescueAllocations() const { return iArena.Allocations(); } // device allocations of the rescue buffers so far
    TUint64 DeviceAllocations() const;                   // ... and of everything the context allocated for batches (descriptors, plans): ohgpu_device_allocations
private:
    struct Lane;
    TBool Prepare(Lane& aLane, RescueBatch& aBatch, TBool aMayBlock);   // tick step 1; false: the lane sits this period out
    Msg* Next(Lane& aLane, TBool aMayBlock);             // tick step 3
    Msg* Handle(Lane& aLane, Msg* aMsg);                 // one message leaving the inbox; nullptr = consumed
    void QueueRescue(Lane& aLane, RescueBatch& aBatch);  // the lane's last millisecond -> a request
    void RescueNow(Lane& aLane);                         // a rescue decided while dequeuing (drain, flush): a batch of one
    void RememberAudio(Lane& aLane, MsgAudio* aMsg);
    void SetBuffering(Lane& aLane, TBool aBuffering);
    void NewStream(Lane& aLane);
private:
    MsgFactory& iFactory;
    std::vector<std::unique_ptr<Lane>> iLanes;
    RescueArena iArena;                                   // the rescues' device buffers (one tick, one thread)
    std::atomic<TUint64> iRescueLaunches;
    void ClaimDriverThread();                            // the first caller of Tick / Pull is the driver; anyone else ASSERTs
    std::mutex iDriverLock;
    std::thread::id iDriver;
    TBool iDriverKnown;
};

} // namespace Media
} // namespace OpenHome
