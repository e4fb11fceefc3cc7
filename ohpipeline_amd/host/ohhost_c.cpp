// ohhost_c.cpp -- C entry points over the host adapter, for scripted callers (tests, bench.py).
// Exceptions never cross this boundary: AssertionFailed -> -1, SampleRateInvalid -> -2.
#include <cstdint>
#include <vector>

#include "Ramp.h"

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
