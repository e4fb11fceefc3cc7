// StarvationRamper.h -- the batch builder's second half: the element that decides when a stream's
// audio is ramped.  Host-side mirrors of
//   MsgReservoir       OpenHome/Media/Pipeline/Msg.h:1326-1460, Msg.cpp:3101-3560   a queue that counts what it holds
//   StarvationRamper   OpenHome/Media/Pipeline/StarvationRamper.{h,cpp}:100-209, 372-920
// Same structure as the reference: a puller thread fills the reservoir from upstream until it holds aSizeJiffies; the
// driver side calls Pull().  When Pull() finds the reservoir empty while running, the most recent millisecond of audio
// goes through FlywheelInput -> RampGenerator (rows a11, N1, a12 -- on the device, see RampGenerator.h) and the
// extrapolated, down-ramped audio is handed out, then a MsgHalt, then the next audio ramps up over aRampUpSize.
// Flush(id) ramps down over 20 ms and discards up to that flush; DrainAllAudio() forces a starvation ramp and discards
// audio until the next MsgDrain.  No PCM byte is touched here: ramps are Ramp fields on the messages, applied when the
// audio is read (row a7).
// Left out: DSD (MsgAudioDsd passes through unramped), the observer thread (the observer is called synchronously, what
// the reference's tests do with ElementObserverSync), thread priorities.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

#include "Msg.h"
#include "RampGenerator.h"

namespace OpenHome {
namespace Media {

class Semaphore {                                        // ohNet's counting semaphore: Wait / Signal / Clear
public:
    explicit Semaphore(TUint aCount = 0) : iCount(aCount) {}
    void Wait();
    void Signal();
    TBool Clear();                                       // true if a signal was pending
private:
    std::mutex iLock;
    std::condition_variable iCv;
    TUint iCount;
};

class IStarvationRamperObserver {                        // StarvationRamper.h:22-27
public:
    virtual ~IStarvationRamperObserver() {}
    virtual void NotifyStarvationRamperBuffering(TBool aBuffering) = 0;
};

class MsgReservoir {
protected:
    MsgReservoir();
    virtual ~MsgReservoir();
    void DoEnqueue(Msg* aMsg);
    Msg* DoDequeue(TBool aAllowNull = false);            // blocks until a message is available
    void EnqueueAtHead(Msg* aMsg);
    TUint Jiffies() const { return iJiffies.load(); }
    TUint TrackCount() const { return iTrackCount.load(); }
    TUint DecodedStreamCount() const { return iDecodedStreamCount.load(); }
    TUint DecodedAudioCount() const { return iDecodedAudioCount.load(); }
public:
    TBool IsEmpty() const;
    TUint NumMsgs() const;                               // test use only
private: // hooks, called as a message enters / leaves (Msg.cpp:3209-3243)
    virtual void ProcessMsgIn(MsgTrack*) {}
    virtual void ProcessMsgIn(MsgDrain*) {}
    virtual void ProcessMsgIn(MsgDelay*) {}
    virtual void ProcessMsgIn(MsgHalt*) {}
    virtual void ProcessMsgIn(MsgDecodedStream*) {}
    virtual void ProcessMsgIn(MsgQuit*) {}
    virtual Msg* ProcessMsgOut(MsgMode* aMsg);
    virtual Msg* ProcessMsgOut(MsgTrack* aMsg);
    virtual Msg* ProcessMsgOut(MsgDrain* aMsg);
    virtual Msg* ProcessMsgOut(MsgDelay* aMsg);
    virtual Msg* ProcessMsgOut(MsgMetaText* aMsg);
    virtual Msg* ProcessMsgOut(MsgHalt* aMsg);
    virtual Msg* ProcessMsgOut(MsgFlush* aMsg);
    virtual Msg* ProcessMsgOut(MsgWait* aMsg);
    virtual Msg* ProcessMsgOut(MsgDecodedStream* aMsg);
    virtual Msg* ProcessMsgOut(MsgAudioPcm* aMsg);
    virtual Msg* ProcessMsgOut(MsgAudioDsd* aMsg);
    virtual Msg* ProcessMsgOut(MsgSilence* aMsg);
private:
    class ProcessorIn;
    class ProcessorOut;
    friend class ProcessorIn;
    friend class ProcessorOut;
private:
    mutable std::mutex iLock;
    std::deque<Msg*> iQueue;
    Semaphore iSem;
    std::atomic<TUint> iJiffies, iTrackCount, iDecodedStreamCount, iDecodedAudioCount;
};

class StarvationRamper : public MsgReservoir, public IPipelineElementUpstream {
public:
    static const TUint kTrainingJiffies = Jiffies::kPerMs * 1;       // StarvationRamper.cpp:374-376
    static const TUint kRampDownJiffies = Jiffies::kPerMs * 20;
    static const TUint kMaxAudioOutJiffies = Jiffies::kPerMs * 5;
    enum class State { Starting, Running, Halted, RampingUp, FlywheelRamping, RampingDown, Flushing };
public:
    StarvationRamper(MsgFactory& aMsgFactory, IPipelineElementUpstream& aUpstream, IStarvationRamperObserver& aObserver,
                     TUint aSizeJiffies, TUint aRampUpSize, TUint aMaxStreamCount);
    /** The puller thread ends once it has passed a MsgQuit on, as in the reference: send one before destroying. */
    ~StarvationRamper();
    void Flush(TUint aId);          // ramps down quickly then discards everything up to a flush with the given id
    void DrainAllAudio();           // from IPipelineDrainer: discard buffered audio, forcing a starvation ramp, until the next MsgDrain
    TUint SizeInJiffies() const { return Jiffies(); }
    void WaitForOccupancy(TUint aJiffies);               // from IStarvationRamper: Pull blocks once until this level is reached
public: // from IPipelineElementUpstream
    Msg* Pull() override;
public: // what the reference's suite reads as a friend
    State CurrentState() const { return iState; }
    TBool Draining() const { return iDraining.load(); }
    TBool StartDrainPending() const { return iStartDrain.load(); }
private:
    TBool IsFull() const { return Jiffies() >= iMaxJiffies || DecodedStreamCount() == iMaxStreamCount; }
    void PullerThread();
    void StartFlywheelRamp();
    void NewStream();
    void ProcessAudioOut(MsgAudio* aMsg);
    void SetBuffering(TBool aBuffering);
private: // from MsgReservoir
    void ProcessMsgIn(MsgTrack* aMsg) override;
    void ProcessMsgIn(MsgDrain* aMsg) override;
    void ProcessMsgIn(MsgDelay* aMsg) override;
    void ProcessMsgIn(MsgHalt* aMsg) override;
    void ProcessMsgIn(MsgDecodedStream* aMsg) override;
    void ProcessMsgIn(MsgQuit* aMsg) override;
    Msg* ProcessMsgOut(MsgMode* aMsg) override;
    Msg* ProcessMsgOut(MsgTrack* aMsg) override;
    Msg* ProcessMsgOut(MsgDrain* aMsg) override;
    Msg* ProcessMsgOut(MsgMetaText* aMsg) override;
    Msg* ProcessMsgOut(MsgHalt* aMsg) override;
    Msg* ProcessMsgOut(MsgFlush* aMsg) override;
    Msg* ProcessMsgOut(MsgWait* aMsg) override;
    Msg* ProcessMsgOut(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsgOut(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsgOut(MsgSilence* aMsg) override;
private:
    MsgFactory& iMsgFactory;
    IPipelineElementUpstream& iUpstream;
    IStarvationRamperObserver& iObserver;
    std::atomic<TUint> iMaxJiffies;
    const TUint iRampUpJiffies, iMaxStreamCount;
    std::mutex iLock;
    Semaphore iSem;
    FlywheelInput iFlywheelInput;
    RampGenerator* iRampGenerator;
    std::thread iPullerThread;
    std::deque<MsgAudio*> iRecentAudio;
    TUint iRecentAudioJiffies;
    IStreamHandler* iStreamHandler;
    State iState;
    TBool iStarving;
    std::atomic<TBool> iExit, iStartDrain, iDraining;
    std::string iMode;
    TUint iStreamId, iSampleRate, iBitDepth, iNumChannels;
    AudioFormat iFormat;
    TUint iCurrentRampValue, iRemainingRampSize, iTargetFlushId, iLastPulledAudioRampValue;
    std::atomic<TUint> iTrackStreamCount, iDrainCount, iHaltCount, iStartOccupancyJiffies;
    Semaphore iSemStartOccupancy;
    std::atomic<TBool> iEventBuffering;
};

} // namespace Media
} // namespace OpenHome
