// StarvationRamper.h -- the reference's element surface (OpenHome/Media/Pipeline/StarvationRamper.h:100-209) over ONE lane of a
// StarvationManager: same constructor arguments, Pull(), Flush(id), DrainAllAudio(), WaitForOccupancy().  Everything it
// does is the manager's doing (StarvationManager.{h,cpp}); a host that runs many pipelines gives the lanes to one manager and
// ticks them together, so that the streams that run dry in the same period share their device passes.
#pragma once

#include "StarvationManager.h"

namespace OpenHome {
namespace Media {

class StarvationRamper : public IPipelineElementUpstream {
public:
    static const TUint kTrainingJiffies = StarvationManager::kTrainingJiffies;
    static const TUint kRampDownJiffies = StarvationManager::kRampDownJiffies;
    static const TUint kMaxAudioOutJiffies = StarvationManager::kMaxAudioOutJiffies;
    typedef LaneState State;
public:
    StarvationRamper(MsgFactory& aMsgFactory, IPipelineElementUpstream& aUpstream, IStarvationRamperObserver& aObserver,
                     TUint aSizeJiffies, TUint aRampUpSize, TUint aMaxStreamCount)
        : iManager(aMsgFactory)
    {
        StarvationManager::LaneConfig cfg;
        cfg.upstream = &aUpstream;
        cfg.observer = &aObserver;
        cfg.sizeJiffies = aSizeJiffies;
        cfg.rampUpJiffies = aRampUpSize;
        cfg.maxStreamCount = aMaxStreamCount;
        iLane = iManager.AddLane(cfg);
    }
    void Flush(TUint aId) { iManager.Flush(iLane, aId); }
    void DrainAllAudio() { iManager.DrainAllAudio(iLane); }
    void WaitForOccupancy(TUint aJiffies) { iManager.WaitForOccupancy(iLane, aJiffies); }
    TUint SizeInJiffies() const { return iManager.SizeInJiffies(iLane); }
public: // from IPipelineElementUpstream
    Msg* Pull() override { return iManager.Pull(iLane); }
public: // what the reference's suite reads as a friend
    State CurrentState() const { return iManager.State(iLane); }
    TBool IsEmpty() const { return iManager.IsEmpty(iLane); }
    TBool Draining() const { return iManager.Draining(iLane); }
    TBool DrainRequested() const { return iManager.DrainRequested(iLane); }
    TUint64 RescueLaunches() const { return iManager.RescueLaunches(); }
private:
    StarvationManager iManager;
    TUint iLane;
};

} // namespace Media
} // namespace OpenHome
