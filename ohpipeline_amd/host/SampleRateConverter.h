// SampleRateConverter.h -- the pipeline element BASELINE.json's north star names.  The reference has NO
// sample-rate converter: it rejects unsupported rates (Codec/CodecController.cpp:724-726) and pulls
// the DAC clock instead (ClockPuller.h:17-34).  This element is therefore new; it follows the reference's element
// conventions (PipelineElement + IPipelineElementUpstream, Pull() = upstream.Pull()->Process(*this),
// Ramper.cpp:46-58) and the reference's lazy data plane: it never touches a PCM byte.  It keeps each stream's input
// history and hands downstream MsgAudioPcm messages at the output rate whose audio is *virtual* -- "output frames
// [m0, m0+n) of this stream" -- so that downstream elements can split them and set ramps on them as usual, and the
// fused resample -> ramp -> pack kernel produces the bytes when the driver finally reads the playable.
// Specification of the filter: DESIGN.md section 4.
#pragma once

#include <memory>
#include <mutex>
#include <vector>

#include "Msg.h"

struct ohgpu_src_msg_desc;

namespace OpenHome {
namespace Media {

/** Input history of one rate-converted stream (shared by the output messages that refer to it); the filter is the factory's,
 *  shared by every stream of the same conversion (MsgFactory::SharedFilter).
 *  The history is a ring of aHistoryMs of input (default two seconds: far beyond what a pipeline buffers between this element and
 *  its driver -- Pipeline.h:97-109's reservoirs hold well under one).  A reader asks for the WINDOW of input a run of output frames
 *  is made of and copies just that (PlayableBatch::Run packs a period's windows back to back: a kilobyte or two per 5 ms message, not
 *  the history).  Append is called by the element's puller thread, the window calls by the driver's thread: both take iLock. */
class SampleRateConverterStream {
public:
    SampleRateConverterStream(MsgFactory& aFactory, TUint aRateIn, TUint aRateOut, TUint aChannels, TUint aBitDepth,
                              AudioDataEndian aEndian, TUint aTapsPerPhase, double aBeta, double aPassHz, TUint aHistoryMs = 2000);
    /** ... over a filter the caller holds (what the factory's cache hands out; a test's own). */
    SampleRateConverterStream(const SrcFilter& aFilter, TUint aRateIn, TUint aChannels, TUint aBitDepth, AudioDataEndian aEndian,
                              TUint aHistoryMs = 2000);
    /** Appends input frames; returns how many output frames exist now (ceil(in*L/M)). */
    TUint64 Append(const TByte* aData, TUint aBytes);
    TUint64 InputFrames() const;
    /** The input frames [aFirst, aFirst + aFrames) that output frames [aOut0, aOut0 + aCount) are made of (DESIGN.md section 4:
     *  n0(m) = floor(m * M / L); output m reads n0(m) - T + 1 .. n0(m); frames before the stream's start do not exist). */
    void Window(TUint64 aOut0, TUint aCount, TUint64& aFirst, TUint& aFrames) const;
    /** Copies input frames [aFirst, aFirst + aFrames) to aDst; ASSERTs that the ring still holds them. */
    void CopyFrames(TUint64 aFirst, TUint aFrames, TByte* aDst) const;
    const SrcFilter& Filter() const { return iFilter; }
    TUint FrameBytes() const { return iFrameBytes; }
    TUint SourceBitDepth() const { return iBitDepth; }
    AudioDataEndian SourceEndian() const { return iEndian; }
    TUint L() const { return iFilter.L; }
    TUint M() const { return iFilter.M; }
private:
    const SrcFilter& iFilter;
    const TUint iChannels, iBitDepth, iFrameBytes;
    const AudioDataEndian iEndian;
    mutable std::mutex iLock;
    std::vector<TByte> iRing;        // frame f lives at (f % iCapacity) * iFrameBytes
    TUint64 iCapacity;               // frames
    TUint64 iFrames;                 // appended so far: the ring holds [max(0, iFrames - iCapacity), iFrames)
};

class SampleRateConverter : public PipelineElement, public IPipelineElementUpstream {
    static const TUint kSupportedMsgTypes;
public:
    SampleRateConverter(MsgFactory& aFactory, IPipelineElementUpstream& aUpstreamElement, TUint aOutputRate,
                        TUint aTapsPerPhase = 32, double aBeta = 9.0, double aPassHz = 20000.0);
public: // from IPipelineElementUpstream
    Msg* Pull() override;
private: // IMsgProcessor
    Msg* ProcessMsg(MsgDecodedStream* aMsg) override;
    Msg* ProcessMsg(MsgAudioPcm* aMsg) override;
    Msg* ProcessMsg(MsgSilence* aMsg) override;
    Msg* ProcessMsg(MsgHalt* aMsg) override;
private:
    MsgFactory& iFactory;
    IPipelineElementUpstream& iUpstreamElement;
    const TUint iOutputRate, iTapsPerPhase;
    const double iBeta, iPassHz;
    DecodedStreamInfo iInfo;
    std::shared_ptr<SampleRateConverterStream> iStream;   // null while the stream already runs at the output rate
    TUint64 iOutFrames;                                   // output frames handed downstream so far
    TUint64 iTrackOffset;                                 // jiffies, at the output rate
};

} // namespace Media
} // namespace OpenHome
