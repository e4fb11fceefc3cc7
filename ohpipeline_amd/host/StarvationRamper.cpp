// StarvationRamper.cpp -- see StarvationRamper.h.  File:line comments are relative to the reference tree
// (OpenHome/Media/Pipeline/StarvationRamper.cpp unless another file is named).
#include "StarvationRamper.h"

#include <algorithm>

using namespace OpenHome;
using namespace OpenHome::Media;

// ---------------------------------------------------------------------------------------------- Semaphore
void Semaphore::Wait()
{
    std::unique_lock<std::mutex> lock(iLock);
    iCv.wait(lock, [this] { return iCount > 0; });
    iCount--;
}

void Semaphore::Signal()
{
    {
        std::lock_guard<std::mutex> lock(iLock);
        iCount++;
    }
    iCv.notify_one();
}

TBool Semaphore::Clear()
{
    std::lock_guard<std::mutex> lock(iLock);
    const TBool pending = iCount > 0;
    iCount = 0;
    return pending;
}

// ---------------------------------------------------------------------------------------------- MsgReservoir
// Msg.cpp:3248-3578: counts go up as a message enters (DoEnqueue and EnqueueAtHead), down as it leaves; the ProcessMsgIn
// hooks run for DoEnqueue only, the ProcessMsgOut hooks for DoDequeue.
class MsgReservoir::ProcessorIn : public IMsgProcessor {
public:
    ProcessorIn(MsgReservoir& aQueue, TBool aHooks) : iQueue(aQueue), iHooks(aHooks) {}
private:
    Msg* ProcessMsg(MsgMode* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgTrack* aMsg) override { iQueue.iTrackCount++; if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
    Msg* ProcessMsg(MsgDrain* aMsg) override { if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
    Msg* ProcessMsg(MsgDelay* aMsg) override { if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgStreamSegment* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgAudioEncoded* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgMetaText* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgHalt* aMsg) override { if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
    Msg* ProcessMsg(MsgFlush* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgWait* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override { iQueue.iDecodedStreamCount++; if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override { iQueue.iDecodedAudioCount++; iQueue.iJiffies += aMsg->Jiffies(); return aMsg; }
    Msg* ProcessMsg(MsgAudioDsd* aMsg) override { iQueue.iDecodedAudioCount++; return aMsg; }
    Msg* ProcessMsg(MsgSilence* aMsg) override { iQueue.iDecodedAudioCount++; iQueue.iJiffies += aMsg->Jiffies(); return aMsg; }
    Msg* ProcessMsg(MsgPlayable*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgQuit* aMsg) override { if (iHooks) iQueue.ProcessMsgIn(aMsg); return aMsg; }
private:
    MsgReservoir& iQueue;
    TBool iHooks;
};

class MsgReservoir::ProcessorOut : public IMsgProcessor {
public:
    explicit ProcessorOut(MsgReservoir& aQueue) : iQueue(aQueue) {}
private:
    Msg* ProcessMsg(MsgMode* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgTrack* aMsg) override { iQueue.iTrackCount--; return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgDrain* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgDelay* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgEncodedStream* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgStreamSegment* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgAudioEncoded* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgMetaText* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgStreamInterrupted* aMsg) override { return aMsg; }
    Msg* ProcessMsg(MsgHalt* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgFlush* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgWait* aMsg) override { return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override { iQueue.iDecodedStreamCount--; return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override { iQueue.iDecodedAudioCount--; iQueue.iJiffies -= aMsg->Jiffies(); return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgAudioDsd* aMsg) override { iQueue.iDecodedAudioCount--; return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgSilence* aMsg) override { iQueue.iDecodedAudioCount--; iQueue.iJiffies -= aMsg->Jiffies(); return iQueue.ProcessMsgOut(aMsg); }
    Msg* ProcessMsg(MsgPlayable*) override { ASSERTS(); return nullptr; }
    Msg* ProcessMsg(MsgQuit* aMsg) override { return aMsg; }
private:
    MsgReservoir& iQueue;
};

MsgReservoir::MsgReservoir()
    : iJiffies(0)
    , iTrackCount(0)
    , iDecodedStreamCount(0)
    , iDecodedAudioCount(0)
{
}

MsgReservoir::~MsgReservoir()
{
    for (auto* m : iQueue) {
        m->RemoveRef();
    }
}

void MsgReservoir::DoEnqueue(Msg* aMsg)
{
    ASSERT(aMsg != nullptr);
    ProcessorIn procIn(*this, true);
    Msg* msg = aMsg->Process(procIn);
    {
        std::lock_guard<std::mutex> lock(iLock);
        iQueue.push_back(msg);
    }
    iSem.Signal();
}

Msg* MsgReservoir::DoDequeue(TBool aAllowNull)
{
    Msg* msg;
    do {
        iSem.Wait();                                     // MsgQueue::Dequeue blocks until there is a message, Msg.cpp:3065-3070
        {
            std::lock_guard<std::mutex> lock(iLock);
            msg = iQueue.front();
            iQueue.pop_front();
        }
        ProcessorOut procOut(*this);
        msg = msg->Process(procOut);
    } while (!aAllowNull && msg == nullptr);
    return msg;
}

void MsgReservoir::EnqueueAtHead(Msg* aMsg)
{
    ProcessorIn proc(*this, false);
    Msg* msg = aMsg->Process(proc);
    {
        std::lock_guard<std::mutex> lock(iLock);
        iQueue.push_front(msg);
    }
    iSem.Signal();
}

TBool MsgReservoir::IsEmpty() const
{
    std::lock_guard<std::mutex> lock(iLock);
    return iQueue.empty();
}

TUint MsgReservoir::NumMsgs() const
{
    std::lock_guard<std::mutex> lock(iLock);
    return (TUint)iQueue.size();
}

Msg* MsgReservoir::ProcessMsgOut(MsgMode* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgTrack* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgDrain* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgDelay* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgMetaText* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgHalt* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgFlush* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgWait* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgDecodedStream* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgAudioPcm* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgAudioDsd* aMsg) { return aMsg; }
Msg* MsgReservoir::ProcessMsgOut(MsgSilence* aMsg) { return aMsg; }

// ---------------------------------------------------------------------------------------------- StarvationRamper
StarvationRamper::StarvationRamper(MsgFactory& aMsgFactory, IPipelineElementUpstream& aUpstream,
                                   IStarvationRamperObserver& aObserver, TUint aSizeJiffies, TUint aRampUpSize,
                                   TUint aMaxStreamCount)
    : iMsgFactory(aMsgFactory)
    , iUpstream(aUpstream)
    , iObserver(aObserver)
    , iMaxJiffies(aSizeJiffies)
    , iRampUpJiffies(aRampUpSize)
    , iMaxStreamCount(aMaxStreamCount)
    , iSem(0)
    , iFlywheelInput(aMsgFactory, kTrainingJiffies)
    , iRecentAudioJiffies(0)
    , iStreamHandler(nullptr)
    , iState(State::Halted)
    , iStarving(false)
    , iExit(false)
    , iStartDrain(false)
    , iDraining(false)
    , iStreamId(IStreamHandler::kStreamIdInvalid)
    , iSampleRate(0)
    , iBitDepth(0)
    , iNumChannels(0)
    , iFormat(AudioFormat::Undefined)
    , iCurrentRampValue(Ramp::kMin)
    , iRemainingRampSize(0)
    , iTargetFlushId(MsgFlush::kIdInvalid)
    , iLastPulledAudioRampValue(Ramp::kMax)
    , iTrackStreamCount(0)
    , iDrainCount(0)
    , iHaltCount(0)
    , iStartOccupancyJiffies(0)
    , iSemStartOccupancy(0)
    , iEventBuffering(false)
{
    SetBuffering(true);
    iRampGenerator = new RampGenerator(aMsgFactory, kTrainingJiffies, kRampDownJiffies);
    iPullerThread = std::thread(&StarvationRamper::PullerThread, this);
}

StarvationRamper::~StarvationRamper()
{
    if (iPullerThread.joinable()) {
        iSem.Signal();                                   // a puller parked on a full reservoir must see iExit
        iPullerThread.join();
    }
    delete iRampGenerator;
    for (auto* m : iRecentAudio) {
        m->RemoveRef();
    }
}

void StarvationRamper::Flush(TUint aId)
{
    std::lock_guard<std::mutex> lock(iLock);
    iTargetFlushId = aId;
    iCurrentRampValue = Ramp::kMax;
    iRemainingRampSize = kRampDownJiffies;
    iState = State::RampingDown;
}

void StarvationRamper::DrainAllAudio()
{
    iStartDrain.store(true);
}

void StarvationRamper::PullerThread()
{                                                        // :452-472
    do {
        Msg* msg = iUpstream.Pull();
        TBool isFull, triggerStart;
        {
            std::lock_guard<std::mutex> lock(iLock);
            DoEnqueue(msg);
            isFull = IsFull();
            if (isFull) {
                (void)iSem.Clear();
            }
            const TUint startOccupancy = iStartOccupancyJiffies.load();
            triggerStart = startOccupancy > 0 && Jiffies() >= startOccupancy;
        }
        if (triggerStart) {
            iSemStartOccupancy.Signal();
        }
        if (isFull && !iExit.load()) {
            iSem.Wait();
        }
    } while (!iExit.load());
}

void StarvationRamper::StartFlywheelRamp()
{                                                        // :474-518
    if (iRecentAudioJiffies > kTrainingJiffies) {
        TInt excess = (TInt)(iRecentAudioJiffies - kTrainingJiffies);
        while (excess > 0) {
            MsgAudio* audio = iRecentAudio.front();
            iRecentAudio.pop_front();
            if (audio->Jiffies() > (TUint)excess) {
                MsgAudio* remaining = audio->Split((TUint)excess);
                iRecentAudio.push_front(remaining);
            }
            const TUint msgJiffies = audio->Jiffies();
            excess -= (TInt)msgJiffies;
            iRecentAudioJiffies -= msgJiffies;
            audio->RemoveRef();
        }
    }
    else {
        TInt remaining = (TInt)(kTrainingJiffies - iRecentAudioJiffies);
        while (remaining > 0) {
            TUint size = std::min((TUint)remaining, (TUint)kMaxAudioOutJiffies);
            MsgSilence* silence = iMsgFactory.CreateMsgSilence(size, iSampleRate, iBitDepth, iNumChannels);
            iRecentAudio.push_front(silence);
            size = silence->Jiffies();                   // original size may have been rounded to a sample boundary
            remaining -= (TInt)size;
            iRecentAudioJiffies += size;
        }
    }

    const Brx& recentSamples = iFlywheelInput.Prepare(iRecentAudio, iRecentAudioJiffies, iSampleRate, iBitDepth, iNumChannels);
    iRecentAudioJiffies = 0;
    ASSERT(iRecentAudio.empty());

    const TUint rampStart = iCurrentRampValue;
    iRampGenerator->Start(recentSamples, iSampleRate, iNumChannels, iBitDepth, rampStart);
    iState = State::FlywheelRamping;

    iStarving = true;
    if (iStreamHandler != nullptr) {
        iStreamHandler->NotifyStarving(Brn((const TByte*)iMode.data(), (TUint)iMode.size()), iStreamId, true);
    }
}

void StarvationRamper::NewStream()
{
    iState = State::Starting;
    for (auto* m : iRecentAudio) {
        m->RemoveRef();
    }
    iRecentAudio.clear();
    iRecentAudioJiffies = 0;
    iStreamId = IStreamHandler::kStreamIdInvalid;
    iLastPulledAudioRampValue = Ramp::kMax;
}

void StarvationRamper::ProcessAudioOut(MsgAudio* aMsg)
{                                                        // :529-559
    if (iStarving) {
        iStarving = false;
        if (iStreamHandler != nullptr) {
            iStreamHandler->NotifyStarving(Brn((const TByte*)iMode.data(), (TUint)iMode.size()), iStreamId, false);
        }
    }
    if (iFormat == AudioFormat::Dsd) {
        return;
    }
    iLastPulledAudioRampValue = aMsg->Ramp().End();

    MsgAudio* clone = aMsg->Clone();
    iRecentAudio.push_back(clone);
    iRecentAudioJiffies += clone->Jiffies();
    if (iRecentAudioJiffies > kTrainingJiffies && iRecentAudio.size() > 1) {
        MsgAudio* audio = iRecentAudio.front();
        iRecentAudio.pop_front();
        iRecentAudioJiffies -= audio->Jiffies();
        if (iRecentAudioJiffies >= kTrainingJiffies) {
            audio->RemoveRef();
        }
        else {
            iRecentAudio.push_front(audio);
            iRecentAudioJiffies += audio->Jiffies();
        }
    }
}

void StarvationRamper::SetBuffering(TBool aBuffering)
{                                                        // :589-604, the observer thread replaced by a direct call
    const TBool prev = iEventBuffering.exchange(aBuffering);
    if (prev != aBuffering) {
        iObserver.NotifyStarvationRamperBuffering(aBuffering);
    }
}

Msg* StarvationRamper::Pull()
{                                                        // :606-659
    {
        const TUint startOccupancy = iStartOccupancyJiffies.load();
        if (startOccupancy > 0 && iDrainCount.load() == 0 && iHaltCount.load() == 0) {
            if (Jiffies() < startOccupancy) {
                iSemStartOccupancy.Wait();
            }
            iStartOccupancyJiffies.store(0);
        }
    }

    if (IsEmpty() || iStartDrain.load()) {
        SetBuffering(true);
        if (iStartDrain.load()) {
            iStartDrain.store(false);
            iDraining.store(true);
        }
        if ((iState == State::Running || (iState == State::RampingUp && iCurrentRampValue != Ramp::kMin)) && !iExit.load()) {
            StartFlywheelRamp();
        }
    }

    Msg* msg = nullptr;
    do {
        if (iRampGenerator->TryGetAudio(msg)) {
            return msg;
        }
        else if (iState == State::FlywheelRamping) {
            iState = State::RampingUp;
            iCurrentRampValue = Ramp::kMin;
            iRemainingRampSize = iRampUpJiffies;
            return iMsgFactory.CreateMsgHalt();
        }

        const TBool wasFlushing = iState == State::Flushing;
        msg = DoDequeue(true);
        {
            std::lock_guard<std::mutex> lock(iLock);
            if (!IsFull()) {
                iSem.Signal();
            }
        }
        if (wasFlushing && iState == State::Flushing && msg != nullptr) {
            msg->RemoveRef();
            msg = nullptr;
        }
    } while (msg == nullptr);
    return msg;
}

void StarvationRamper::ProcessMsgIn(MsgTrack*) { iTrackStreamCount++; }
void StarvationRamper::ProcessMsgIn(MsgDrain*) { iDrainCount++; iSemStartOccupancy.Signal(); }
void StarvationRamper::ProcessMsgIn(MsgDelay* aMsg) { iMaxJiffies.store(std::max(aMsg->RemainingJiffies(), 140 * Jiffies::kPerMs)); }
void StarvationRamper::ProcessMsgIn(MsgHalt*) { iHaltCount++; iSemStartOccupancy.Signal(); }
void StarvationRamper::ProcessMsgIn(MsgDecodedStream*) { iTrackStreamCount++; }
void StarvationRamper::ProcessMsgIn(MsgQuit*) { iExit.store(true); }

Msg* StarvationRamper::ProcessMsgOut(MsgMode* aMsg)
{
    NewStream();
    iMode = aMsg->Mode();
    return aMsg;
}

Msg* StarvationRamper::ProcessMsgOut(MsgTrack* aMsg)
{
    NewStream();
    iTrackStreamCount--;
    aMsg->RemoveRef();
    return nullptr;
}

Msg* StarvationRamper::ProcessMsgOut(MsgDrain* aMsg)
{
    iDrainCount--;
    iDraining.store(false);
    if (iState == State::Running || (iState == State::RampingUp && iCurrentRampValue != Ramp::kMin)) {
        EnqueueAtHead(aMsg);
        SetBuffering(true);
        StartFlywheelRamp();
        return nullptr;
    }
    return aMsg;
}

Msg* StarvationRamper::ProcessMsgOut(MsgMetaText* aMsg)
{
    aMsg->RemoveRef();
    return nullptr;
}

Msg* StarvationRamper::ProcessMsgOut(MsgHalt* aMsg)
{
    // set Halted state on both entry and exit of this msg (:743-751)
    iState = State::Halted;
    iHaltCount--;
    return aMsg;
}

Msg* StarvationRamper::ProcessMsgOut(MsgFlush* aMsg)
{
    const TUint id = aMsg->Id();
    aMsg->RemoveRef();
    if (iTargetFlushId != MsgFlush::kIdInvalid && id == iTargetFlushId) {
        if (iState == State::RampingDown) {
            StartFlywheelRamp();
        }
        else if (iState == State::Flushing) {
            iState = State::Halted;
            iTargetFlushId = MsgFlush::kIdInvalid;
            return iMsgFactory.CreateMsgHalt();
        }
    }
    return nullptr;
}

Msg* StarvationRamper::ProcessMsgOut(MsgWait* aMsg)
{
    aMsg->RemoveRef();
    return nullptr;
}

Msg* StarvationRamper::ProcessMsgOut(MsgDecodedStream* aMsg)
{
    NewStream();
    iTrackStreamCount--;

    const DecodedStreamInfo& streamInfo = aMsg->StreamInfo();
    iStreamId = streamInfo.StreamId();
    iStreamHandler = streamInfo.StreamHandler();
    iSampleRate = streamInfo.SampleRate();
    iBitDepth = streamInfo.BitDepth();
    iNumChannels = streamInfo.NumChannels();
    iFormat = streamInfo.Format();
    iCurrentRampValue = Ramp::kMax;
    return aMsg;
}

Msg* StarvationRamper::ProcessMsgOut(MsgAudioPcm* aMsg)
{                                                        // :792-834
    if (iDraining.load()) {
        aMsg->RemoveRef();
        return nullptr;
    }
    if (iState == State::Starting || iState == State::Halted) {
        iState = State::Running;
    }

    if (aMsg->Jiffies() > kMaxAudioOutJiffies) {
        MsgAudio* split = aMsg->Split(kMaxAudioOutJiffies);
        EnqueueAtHead(split);
    }

    if ((iState == State::RampingUp || iState == State::RampingDown) && iRemainingRampSize > 0) {
        if (aMsg->Jiffies() > iRemainingRampSize) {
            MsgAudio* remaining = aMsg->Split(iRemainingRampSize);
            EnqueueAtHead(remaining);
        }
        MsgAudio* split = nullptr;
        const Ramp::EDirection direction = iState == State::RampingUp ? Ramp::EUp : Ramp::EDown;
        iCurrentRampValue = aMsg->SetRamp(iCurrentRampValue, iRemainingRampSize, direction, split);
        if (split != nullptr) {
            EnqueueAtHead(split);
        }
        if (iRemainingRampSize == 0) {
            iState = (iState == State::RampingUp) ? State::Running : State::Flushing;
        }
    }

    ProcessAudioOut(aMsg);
    SetBuffering(false);
    return aMsg;
}

Msg* StarvationRamper::ProcessMsgOut(MsgSilence* aMsg)
{
    if (iDraining.load()) {
        aMsg->RemoveRef();
        return nullptr;
    }
    if (iState == State::Halted) {
        iState = State::Starting;
    }
    if (aMsg->Jiffies() > kMaxAudioOutJiffies) {
        MsgAudio* split = aMsg->Split(kMaxAudioOutJiffies);
        EnqueueAtHead(split);
    }
    ProcessAudioOut(aMsg);
    return aMsg;
}

void StarvationRamper::WaitForOccupancy(TUint aJiffies)
{
    if (iDrainCount.load() > 0 || iHaltCount.load() > 0) {
        return;
    }
    (void)iSemStartOccupancy.Clear();
    iStartOccupancyJiffies.store(aJiffies);
}
